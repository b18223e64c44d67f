#!/bin/bash
# A/B of the LDS-staged column walk (dec_staged.hpp) against the per-pair walk, and of its ring geometry, on one box:
#   bash scripts/sweep_staged.sh <tag> <cfg> [steps]
# prints one line per variant: pair-step ms (the scope k_inv_row_pair) and the step
# STG_VARIANTS="rows bufs loaders [walks per workgroup];..." : geometries of the staged walk to run (0 = the library's choice)
TAG=${1:-r05}; CFG=${2:-cfg4}; STEPS=${3:-5}
mkdir -p gpurun_out/$TAG
run() {   # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-h2d --no-clocks --no-graph-leg \
      > gpurun_out/$TAG/stg_${CFG}_$name.json 2> gpurun_out/$TAG/stg_${CFG}_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/$TAG/stg_${CFG}_$name.err; return 1; }
  python3 - "$name" gpurun_out/$TAG/stg_${CFG}_$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["roofline"]["kernels_ms_per_step"]
print("%-28s pair step %8.4f ms   step %8.4f ms   (col %.3f row %.3f inv %.3f)" % (sys.argv[1], k["k_inv_row_pair"], d["ms_per_step"], k["k_fwd_col"], k["k_fwd_row"], k["k_inv_col_peak"]))
PY
}
run walk_per_pair TDOA_NO_DEC_STAGED=1 || exit 1
# STG_ENV: extra environment of the staged runs (cfg2: STG_ENV=TDOA_DEC_COLS_ALWAYS=1 -- the library's own choice there is the tile form)
run staged_default TDOA_NO_DEC_STAGED=0 $STG_ENV || exit 1
VARIANTS=${STG_VARIANTS:-"2 4 0;4 2 0;4 4 0;8 2 0;4 4 1"}
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  set -- $v
  run staged_r$1_b$2_l$3_c${4:-0} TDOA_DEC_STAGED_ROWS=$1 TDOA_DEC_STAGED_BUFS=$2 TDOA_DEC_STAGED_LOADERS=$3 TDOA_DEC_STAGED_CW=${4:-0} $STG_ENV || exit 1
done
