timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fm.py tests/test_gpu_fuzz.py tests/test_gpu_fine.py tests/test_gpu_anchors.py -x -q > gpurun_out/r03_t11.log 2>&1; echo rc=$?; tail -2 gpurun_out/r03_t11.log
for c in cfg2 cfg4 cfg5; do python3 bench.py --config $c --no-cpu-baseline > gpurun_out/r03_j_$c.json 2> gpurun_out/r03_j_$c.err; echo rc=$?; done
python3 - <<'PY'
import json
for c in ("cfg2","cfg4","cfg5"):
    d=json.loads(open("gpurun_out/r03_j_%s.json"%c).read().strip().splitlines()[-1])
    print(c, d["ms_per_step"], d["value"], d["graph_replay"]["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
PY
