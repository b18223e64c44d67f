#!/bin/bash
# Round profile set, run on the GPU box from the repo root:  bash scripts/collect_profiles.sh r02
#   1. rocprofv3 --kernel-trace --stats of the default bench (cfg2) and of --config cfg3 / cfg4 / cfg5 and cfg2 --max-lag 512
#   2. FETCH_SIZE / WRITE_SIZE passes of the default bench (collect_pmc.sh) -> <tag>_pmc_traffic.json
# Only summaries are kept (copy gpurun_out/<tag>_* into profiles/).
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run_stats() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$name -- \
      python3 bench.py --no-cpu-baseline --no-graph-leg "$@" > gpurun_out/${TAG}_${name}_bench.json 2> gpurun_out/prof_${TAG}_$name.log
  local f=$(ls gpurun_out/prof_${TAG}_$name/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/${TAG}_${name}_kernel_stats.csv
  echo "== $name"; head -12 gpurun_out/${TAG}_${name}_kernel_stats.csv
}
run_stats cfg2 --steps 20 --warmup 5
run_stats cfg2_maxlag512 --steps 20 --warmup 5 --max-lag 512
run_stats cfg4 --config cfg4 --steps 5 --warmup 2
run_stats cfg3 --config cfg3 --steps 2 --warmup 1
run_stats cfg5 --config cfg5 --steps 2 --warmup 1
bash scripts/collect_pmc.sh $TAG
