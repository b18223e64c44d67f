#!/bin/bash
# A/B of the single-look K1 against the statistics pre-pass on one box: alternating runs of bench.py per configuration.
# usage: scripts/ab_k1_once.sh "cfg2 cfg4" [steps]     -> gpurun_out/r04/ab_<cfg>_<once|pre>_<n>.json
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04
for cfg in ${1:-cfg2 cfg4}; do
  for n in 1 2; do
    for mode in once pre; do
      if [ $mode = pre ]; then export TDOA_NO_K1_ONCE=1; else unset TDOA_NO_K1_ONCE; fi
      python3 bench.py --no-cpu-baseline --config $cfg ${2:+--steps $2} > gpurun_out/r04/ab_${cfg}_${mode}_$n.json 2> gpurun_out/r04/ab_${cfg}_${mode}_$n.err || exit 1
      python3 -c "import json; d=json.load(open('gpurun_out/r04/ab_${cfg}_${mode}_$n.json')); print('$cfg $mode $n', d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
    done
  done
done
