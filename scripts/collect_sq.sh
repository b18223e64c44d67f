#!/bin/bash
# SQ (shader sequencer) counters of one bench step, one rocprofv3 pass per counter group.
# usage (GPU box, repo root):  [BENCH_ARGS="--max-lag 512"] bash scripts/collect_sq.sh <tag> "<CTR CTR ...>" ["<CTR ...>" ...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "$@"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/sq_${TAG}_$i -- \
      python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph-leg --no-clocks --no-h2d $BENCH_ARGS > gpurun_out/sq_${TAG}_$i.log 2>&1
  i=$((i+1))
done
python3 scripts/sq_summary.py gpurun_out/sq_${TAG}_ $i
