#!/bin/bash
# HBM-side traffic of every kernel of one bench step of ONE configuration, from the L2 fabric counters.
# Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), counters only
# with --kernel-trace, as MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots) prescribes.
# Run on the GPU box from the repo root:  bash scripts/collect_pmc.sh <tag> [cfg2|cfg3|cfg4|cfg5]
#   -> gpurun_out/<tag>_pmc_traffic_<cfg>.json (bench.py quotes profiles/*pmc_traffic_<cfg>.json of the same kernel sources)
set -e
TAG=${1:-r04}
CFG=${2:-cfg2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${TAG}_${CFG}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_${CFG}_$c -- \
      python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-graph-leg --no-clocks --no-h2d $BENCH_ARGS > gpurun_out/pmc_${TAG}_${CFG}_$c.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_${CFG}_FETCH_SIZE gpurun_out/pmc_${TAG}_${CFG}_WRITE_SIZE gpurun_out/${TAG}_pmc_traffic_${CFG}.json $CFG
