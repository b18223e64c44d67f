#!/bin/bash
# Everything the round's DESIGN.md / bench roofline quote, in one GPU call (about 25 minutes):
#   bash scripts/collect_round.sh r05 [quick]
# (run_stats passes --no-clocks: the profiled process must not replay 1.5 s of extra steps next to rocm-smi children that
#  inherit the profiler's preload; the clocks come from the unprofiled default bench line further down)
# Order matters: the counter tables of every configuration are copied into profiles/ BEFORE the bench lines are taken, so
# that bench.py's roofline can quote them (same kernel sources: source_sha16).
set -e
TAG=${1:-r05}
QUICK=$2
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
say() { echo "[collect $(date +%H:%M:%S)] $*"; }
say hbm microbench
python3 scripts/hbm_microbench.py > gpurun_out/${TAG}_hbm_microbench.txt 2>&1
cp gpurun_out/${TAG}_hbm_microbench.txt profiles/
CFGS="cfg2 cfg4 cfg5 cfg3"
[ "$QUICK" = quick ] && CFGS="cfg2 cfg4"
for cfg in $CFGS; do
  say pmc $cfg
  bash scripts/collect_pmc.sh $TAG $cfg > gpurun_out/${TAG}_pmc_$cfg.log 2>&1
  cp gpurun_out/${TAG}_pmc_traffic_$cfg.json profiles/
  say sq $cfg
  BENCH_ARGS="--config $cfg" bash scripts/collect_sq.sh ${TAG}_$cfg "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > gpurun_out/${TAG}_sq_$cfg.csv
  cp gpurun_out/${TAG}_sq_$cfg.csv profiles/
done
run_stats() {   # name, bench args...
  local name=$1; shift
  say stats $name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$name -- \
      python3 bench.py --no-cpu-baseline --no-graph-leg --no-h2d --no-clocks "$@" > gpurun_out/${TAG}_${name}_bench.json 2> gpurun_out/prof_${TAG}_$name.log
  local f=$(ls gpurun_out/prof_${TAG}_$name/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/${TAG}_${name}_kernel_stats.csv
  cp gpurun_out/${TAG}_${name}_bench.json gpurun_out/${TAG}_${name}_kernel_stats.csv profiles/
}
run_stats cfg2 --steps 20 --warmup 5
run_stats cfg4 --config cfg4 --steps 5 --warmup 2
if [ "$QUICK" != quick ]; then
  run_stats cfg2_maxlag512 --steps 20 --warmup 5 --max-lag 512
  run_stats cfg3 --config cfg3 --steps 2 --warmup 1
  run_stats cfg5 --config cfg5 --steps 2 --warmup 1
fi
say default bench
python3 bench.py > gpurun_out/${TAG}_default_bench.json 2> gpurun_out/${TAG}_default_bench.err
cp gpurun_out/${TAG}_default_bench.json profiles/
say A/B single-look K1 against the pre-pass
TDOA_NO_K1_ONCE=1 python3 bench.py --no-cpu-baseline --no-h2d > gpurun_out/${TAG}_cfg2_prepass_bench.json 2> gpurun_out/${TAG}_prepass.err
cp gpurun_out/${TAG}_cfg2_prepass_bench.json profiles/
say A/B ten-second plan: N = 5 x 2^22 against 2^25
TDOA_POW2_ONLY=1 python3 bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-h2d > gpurun_out/${TAG}_cfg3_pow2_bench.json 2> gpurun_out/${TAG}_cfg3_pow2.err
cp gpurun_out/${TAG}_cfg3_pow2_bench.json profiles/
say A/B staged column walk against the per-pair walk
for cfg in cfg4 cfg5 cfg3; do
  TDOA_NO_DEC_STAGED=1 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-clocks > gpurun_out/${TAG}_${cfg}_perpair_walk_bench.json 2> gpurun_out/${TAG}_${cfg}_perpair.err
  cp gpurun_out/${TAG}_${cfg}_perpair_walk_bench.json profiles/
done
say A/B cfg2: tile form of the pair step, two-kernel small plan on cfg5
TDOA_NO_DEC_COLS=1 python3 bench.py --no-cpu-baseline --no-h2d --no-clocks > gpurun_out/${TAG}_cfg2_tile_form_bench.json 2> gpurun_out/${TAG}_cfg2_tile.err
TDOA_NO_SMALL_FUSED=1 python3 bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-clocks > gpurun_out/${TAG}_cfg5_two_kernel_small_plan_bench.json 2> gpurun_out/${TAG}_cfg5_two_kernel.err
cp gpurun_out/${TAG}_cfg2_tile_form_bench.json gpurun_out/${TAG}_cfg5_two_kernel_small_plan_bench.json profiles/
if [ -x scripts/microbench/lds_dma_probe ]; then
  say LDS-DMA probe
  timeout -k 10 300 scripts/microbench/lds_dma_probe > gpurun_out/${TAG}_lds_dma_probe.txt 2>&1 || true
  cp gpurun_out/${TAG}_lds_dma_probe.txt profiles/
fi
if [ -f tdoa-geolocation_amd/libtdoa_mi355x_stgt.so ]; then      # tdoa_amd.build.build_variant("stgt", ["TDOA_STG_TIMING"])
  say wave-cycle counters of the staged walk
  : > gpurun_out/${TAG}_staged_walk_wave_cycles.txt
  for cfg in cfg2 cfg4 cfg5 cfg3; do
    TDOA_STG_PROF=1 TDOA_LIB_VARIANT=stgt python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-clocks --no-graph-leg > /dev/null 2> gpurun_out/${TAG}_stgt_$cfg.err || true
    python3 - $cfg "$(grep stg_prof gpurun_out/${TAG}_stgt_$cfg.err)" >> gpurun_out/${TAG}_staged_walk_wave_cycles.txt <<'PY'
import sys
v = [int(x) for x in sys.argv[2].split()[1:]]
walks = "%s: walks %.0fk ticks each, at the barrier %.1f %%" % (sys.argv[1], v[0] / v[5] / 1e3, 100 * v[1] / v[0])
if v[6]:
    print("%s; %d loaders: %.0fk ticks each, inside issue() %.1f %%, waiting for their data %.1f %%, at the barrier %.1f %%  (raw: %s)"
          % (walks, v[6], v[2] / v[6] / 1e3, 100 * v[7] / v[2], 100 * v[3] / v[2], 100 * v[4] / v[2], " ".join(map(str, v))))
else:
    print("%s; no loader wave (the walks bring the rows themselves)  (raw: %s)" % (walks, " ".join(map(str, v))))
PY
  done
  cp gpurun_out/${TAG}_staged_walk_wave_cycles.txt profiles/
fi
say multi-rank rehearsals
python3 bench.py --force-dist --no-cpu-baseline > gpurun_out/${TAG}_forcedist_rccl_world1_bench.json 2> gpurun_out/${TAG}_forcedist.err
# (round 5: bench.py --gpus N starts its own ranks -- the contract verb, no launcher in front)
TDOA_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_2rank_gloo.json 2> gpurun_out/${TAG}_gloo2.err
cp gpurun_out/${TAG}_forcedist_rccl_world1_bench.json gpurun_out/${TAG}_bench_2rank_gloo.json profiles/
if [ "$QUICK" != quick ]; then
  say K1 on other byte distributions
  bash scripts/collect_k1_bytes.sh $TAG > gpurun_out/${TAG}_k1bytes.log 2>&1 || true
  cp gpurun_out/${TAG}_k1_bytes.json profiles/ || true
fi
say done
tail -1 gpurun_out/${TAG}_default_bench.json | head -c 600
