#!/bin/bash
# the default bench line next to kernel variants picked through the environment (one JSON summary line each ->
# gpurun_out/variants.jsonl).   usage: VARIANTS="TDOA_NO_FUSED_K1=1 TDOA_NO_DECIMATE=1" bash scripts/bench_variants.sh
out=gpurun_out/variants.jsonl
: > $out
run() {
  env "$@" python bench.py --no-cpu-baseline --steps 30 $BENCH_ARGS 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'variant': '$*', 'ms': d['ms_per_step'], 'graph_ms': (d.get('graph_replay') or {}).get('ms_per_step'), 'kernels': d['roofline']['kernels_ms_per_step']}))" >> $out
}
run TDOA_DEFAULT=1 || exit 1
for v in ${VARIANTS:-TDOA_NO_FUSED_K1=1 TDOA_NO_DECIMATE=1}; do run $v || exit 1; done
cat $out
