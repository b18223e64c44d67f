#!/bin/bash
# same-box A/B of an environment switch of the library (TDOA_NO_...=1) against the default path
# usage: scripts/ab_env.sh TAG "VAR=1" "bench flags"   -> gpurun_out/r04/abe_<TAG>_<default|switch>_<n>.json
cd "$(dirname "$0")/.." || exit 1
tag=$1
mkdir -p gpurun_out/r04
for n in 1 2 3; do
  for mode in default switch; do
    f=gpurun_out/r04/abe_${tag}_${mode}_$n
    if [ $mode = switch ]; then
      env "$2" python3 bench.py --no-cpu-baseline --no-graph-leg --no-h2d $3 > $f.json 2> $f.err || exit 1
    else
      python3 bench.py --no-cpu-baseline --no-graph-leg --no-h2d $3 > $f.json 2> $f.err || exit 1
    fi
    python3 -c "import json; d=json.load(open('$f.json')); print('$tag $mode $n', d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
  done
done
