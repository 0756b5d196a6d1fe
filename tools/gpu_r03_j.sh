#!/bin/bash
FQ_WINDOWS=1 python tools/fq_exp.py 208 8 20 4 25
FQ_WINDOWS=1 CASMTR_FQ_WAVES_PER_XCD=512 python tools/fq_exp.py 208 8 20 4 25
FQ_WINDOWS=1 CASMTR_FQ_WAVES_PER_XCD=384 python tools/fq_exp.py 208 8 20 4 25
FQ_WINDOWS=1 CASMTR_FQ_WAVES_PER_XCD=256 python tools/fq_exp.py 208 8 20 4 25
FQ_WINDOWS=1 python tools/fq_exp.py 416 8 10 2 25
python tools/cascade_only.py 2>&1 | tail -2
