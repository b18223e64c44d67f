timeout -k 10 900 python -m pytest tests/test_gpu_anchors.py tests/test_capi_cpu.py tests/test_c_example.py -x -q > gpurun_out/r03_t6.log 2>&1; echo rc=$?; tail -30 gpurun_out/r03_t6.log
