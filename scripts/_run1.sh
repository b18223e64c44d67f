for c in cfg2 cfg4; do
python3 bench.py --config $c --no-cpu-baseline --no-graph-leg --steps 10 --warmup 3 > gpurun_out/r03_f_$c.json 2> gpurun_out/r03_f_$c.err; echo $c rc=$?
done
python3 - <<'PY'
import json
for f in ("r03_f_cfg2","r03_f_cfg4"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["kernels_ms_per_step"], d.get("parity_window0"))
    except Exception as e: print(f, "ERR", e)
PY
