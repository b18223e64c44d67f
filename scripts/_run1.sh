bash scripts/collect_profiles.sh r03x quick > gpurun_out/r03x_collect.log 2>&1; echo rc=$?; tail -15 gpurun_out/r03x_collect.log
python3 bench.py --steps 20 --warmup 5 --cpu-budget 6 > gpurun_out/r03x_bench.json 2> gpurun_out/r03x_bench.err; echo rc=$?; tail -3 gpurun_out/r03x_bench.err
for sim in fm random; do python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph-leg --sim $sim > gpurun_out/r03x_sim_$sim.json 2> gpurun_out/r03x_sim_$sim.err; echo rc=$?; done
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph-leg --force-dist > gpurun_out/r03x_forcedist.json 2> gpurun_out/r03x_forcedist.err; echo rc=$?; tail -3 gpurun_out/r03x_forcedist.err
python3 - <<'PY'
import json
for f in ("r03x_bench","r03x_sim_fm","r03x_sim_random","r03x_forcedist"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["kernels_ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
