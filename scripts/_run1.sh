for c in cfg2 cfg3; do python3 bench.py --config $c --no-cpu-baseline --no-graph-leg > gpurun_out/r03_h_$c.json 2> gpurun_out/r03_h_$c.err; echo rc=$?; done
python3 - <<'PY'
import json
for c in ("cfg2","cfg3"):
    d=json.loads(open("gpurun_out/r03_h_%s.json"%c).read().strip().splitlines()[-1])
    print(c, d["ms_per_step"], d["value"], d["roofline"]["kernels_ms_per_step"])
PY
