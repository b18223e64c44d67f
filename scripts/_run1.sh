timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest3.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03_gputest3.log
for c in cfg2 cfg3 cfg4 cfg5; do python3 bench.py --config $c --no-cpu-baseline > gpurun_out/r03_k_$c.json 2> gpurun_out/r03_k_$c.err; echo rc=$?; done
python3 - <<'PY'
import json
for c in ("cfg2","cfg3","cfg4","cfg5"):
    d=json.loads(open("gpurun_out/r03_k_%s.json"%c).read().strip().splitlines()[-1])
    print(c, d["ms_per_step"], d["value"], d["graph_replay"]["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
PY
