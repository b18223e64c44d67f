#!/bin/bash
# HBM-side traffic of every kernel of one bench step, from the L2 fabric counters.
# Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), counters only
# with --kernel-trace, as MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots) prescribes.
# Run on the GPU box from the repo root:  bash scripts/collect_pmc.sh <tag>
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${TAG}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$c -- \
      python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/pmc_${TAG}_$c.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/${TAG}_pmc_traffic.json
