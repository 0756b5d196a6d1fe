#!/bin/bash
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/t_all3.txt 2>&1
tail -4 gpurun_out/r03/t_all3.txt
bash tools/refresh_profiles.sh r03b > gpurun_out/r03b_refresh.log 2>&1
tail -3 gpurun_out/r03b_refresh.log | cut -c1-400
