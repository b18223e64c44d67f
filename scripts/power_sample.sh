#!/bin/bash
# Socket power and shader clock while a bench configuration runs (rocm-smi polled every 0.2 s).
#   usage (GPU box): bash scripts/power_sample.sh <label> [ENV=VAL ...] -- <bench args>
label=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
( env "${envs[@]}" python3 bench.py --no-cpu-baseline --no-graph-leg "$@" > gpurun_out/pw_$label.json 2> gpurun_out/pw_$label.err ) &
pid=$!
: > gpurun_out/pw_$label.smi
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --csv 2>/dev/null | tr '\n' ' ' >> gpurun_out/pw_$label.smi; echo >> gpurun_out/pw_$label.smi
  sleep 0.2
done
wait $pid
python3 - "$label" <<'PY'
import re, statistics, json, sys
l = sys.argv[1]
sclk, pw = [], []
for line in open("gpurun_out/pw_%s.smi" % l):
    m = re.search(r"card0,\((\d+)Mhz\),\d+,\((\d+)Mhz\),\d+,\((\d+)Mhz\),\d+,\((\d+)Mhz\),\w+,([\d.]+)", line)
    if m:
        sclk.append(int(m.group(3))); pw.append(float(m.group(5)))
busy = [(s, p) for s, p in zip(sclk, pw) if p > 1000]
d = json.loads(open("gpurun_out/pw_%s.json" % l).read().strip().splitlines()[-1])
print(json.dumps({"run": l, "samples_under_load": len(busy),
                  "median_sclk_MHz_under_load": statistics.median(s for s, _ in busy) if busy else None,
                  "median_socket_power_W_under_load": statistics.median(p for _, p in busy) if busy else None,
                  "ms_per_step": d["ms_per_step"], "kernels_ms_per_step": d["roofline"]["kernels_ms_per_step"]}))
PY
