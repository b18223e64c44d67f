timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fm.py -x -q > gpurun_out/r03_t9.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03_t9.log
