python -m pytest tests/test_gpu_ds_split.py tests/test_tie_semantics.py "tests/test_gpu_ops.py::test_dual_softmax" tests/test_gpu_pipeline.py -q 2>&1 | tail -3
python tools/ds_time.py 2>&1 | grep "gemm=split"
