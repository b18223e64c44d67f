set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest9.log 2>&1; echo rc=$?; tail -4 gpurun_out/r03_gputest9.log
for c in cfg5 cfg5 cfg2; do
python3 bench.py --config $c --no-cpu-baseline > gpurun_out/r03_o_$c.json 2> gpurun_out/r03_o_$c.err; echo rc=$?
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r03_o_$c.json").read().strip().splitlines()[-1])
print("$c", d["ms_per_step"], d["value"], d["graph_replay"]["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
PY
done
