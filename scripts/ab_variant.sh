#!/bin/bash
# same-box A/B of a compile-time variant (tdoa_amd/build.py --variant NAME -D...) against the default build
# usage: scripts/ab_variant.sh NAME "cfg2 cfg4" [steps]   -> gpurun_out/r04/abv_<NAME>_<cfg>_<default|variant>_<n>.json
cd "$(dirname "$0")/.." || exit 1
name=$1
mkdir -p gpurun_out/r04
for cfg in ${2:-cfg2 cfg4}; do
  for n in 1 2; do
    for mode in default variant; do
      if [ $mode = variant ]; then export TDOA_LIB_VARIANT=$name; else unset TDOA_LIB_VARIANT; fi
      f=gpurun_out/r04/abv_${name}_${cfg}_${mode}_$n
      python3 bench.py --no-cpu-baseline --no-graph-leg --config $cfg ${3:+--steps $3} > $f.json 2> $f.err || exit 1
      python3 -c "import json; d=json.load(open('$f.json')); print('$cfg $mode $n', d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
    done
  done
done
