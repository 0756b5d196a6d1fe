#!/bin/bash
python tools/window_coherence.py
python tools/window_coherence.py --masked
python tools/window_coherence.py --config 2c
python tools/window_coherence.py --config indoor
bash tools/refresh_profiles.sh r03a > gpurun_out/r03a_refresh.log 2>&1
tail -3 gpurun_out/r03a_refresh.log
