set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest5.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03_gputest5.log
for c in cfg2 cfg3 cfg4; do
  python3 bench.py --config $c --no-cpu-baseline > gpurun_out/r03_o_$c.json 2> gpurun_out/r03_o_$c.err; echo rc=$?
done
python3 - <<'PY'
import json
for c in ("cfg2","cfg3","cfg4"):
    d=json.loads(open("gpurun_out/r03_o_%s.json"%c).read().strip().splitlines()[-1])
    print(c, d["ms_per_step"], d["value"], d["graph_replay"]["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
PY
bash scripts/collect_pmc.sh r03x > gpurun_out/r03x_pmc.log 2>&1; echo rc=$?
python3 -c "
import json
d=json.load(open('gpurun_out/r03x_pmc_traffic.json'))['kernels']
for k,v in d.items():
    if v['traffic_bytes_per_launch']>1e6: print(k, v)
"
