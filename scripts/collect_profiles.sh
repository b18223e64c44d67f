#!/bin/bash
# Round profile set, run on the GPU box from the repo root:  bash scripts/collect_profiles.sh r03 [quick]
#   1. FETCH_SIZE / WRITE_SIZE passes of the default bench (collect_pmc.sh) -> <tag>_pmc_traffic.json
#   2. SQ counters of the default bench -> <tag>_sq_cfg2.csv
#   3. rocprofv3 --kernel-trace --stats of the default bench (cfg2) and (unless "quick") of --config cfg3 / cfg4 / cfg5 and
#      cfg2 --max-lag 512
# The counter tables are copied into profiles/ BEFORE the bench lines are taken, so that the bench's roofline can quote them.
# Only summaries are kept (copy gpurun_out/<tag>_* into profiles/).
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash scripts/collect_pmc.sh $TAG
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
bash scripts/collect_sq.sh $TAG "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > gpurun_out/${TAG}_sq_cfg2.csv
cp gpurun_out/${TAG}_sq_cfg2.csv profiles/${TAG}_sq_cfg2.csv
run_stats() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$name -- \
      python3 bench.py --no-cpu-baseline --no-graph-leg "$@" > gpurun_out/${TAG}_${name}_bench.json 2> gpurun_out/prof_${TAG}_$name.log
  local f=$(ls gpurun_out/prof_${TAG}_$name/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/${TAG}_${name}_kernel_stats.csv
  echo "== $name"; head -12 gpurun_out/${TAG}_${name}_kernel_stats.csv
}
run_stats cfg2 --steps 20 --warmup 5
if [ "$2" != "quick" ]; then
  run_stats cfg2_maxlag512 --steps 20 --warmup 5 --max-lag 512
  run_stats cfg4 --config cfg4 --steps 5 --warmup 2
  run_stats cfg3 --config cfg3 --steps 2 --warmup 1
  run_stats cfg5 --config cfg5 --steps 2 --warmup 1
fi
