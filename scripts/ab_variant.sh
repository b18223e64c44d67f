#!/bin/bash
# A/B of library variants (measurement builds, tdoa_amd.build.build_variant + TDOA_LIB_VARIANT) against the shipped library, per
# kernel scope:  bash scripts/ab_variant.sh "<variant names, or a blank for none>" "<configs>"
mkdir -p gpurun_out/r05s
VARS=${1:-" "}; CFGS=${2:-"cfg4 cfg5"}
for cfg in $CFGS; do
  for v in base $VARS; do
    if [ $v = base ]; then unset TDOA_LIB_VARIANT; else export TDOA_LIB_VARIANT=$v; fi
    timeout -k 10 300 python3 bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-h2d --no-clocks --no-graph-leg > gpurun_out/r05s/${cfg}_$v.json 2> gpurun_out/r05s/${cfg}_$v.err || { echo "$cfg $v FAILED"; tail -3 gpurun_out/r05s/${cfg}_$v.err; continue; }
    python3 - $cfg $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05s/%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
k = d["roofline"]["kernels_ms_per_step"]
print("%s %-8s pair step %8.4f ms   step %8.4f ms" % (sys.argv[1], sys.argv[2], k["k_inv_row_pair"], d["ms_per_step"]))
PY
  done
  unset TDOA_LIB_VARIANT
done
