set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest4.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03_gputest4.log
for Z in 512 256 1024 1536 0; do
TDOA_ZPAD=$Z python3 bench.py --config cfg3 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r03_z$Z.json 2> gpurun_out/r03_z$Z.err; echo rc=$?
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r03_z$Z.json").read().strip().splitlines()[-1])
print("ZPAD", $Z, d["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
PY
done
