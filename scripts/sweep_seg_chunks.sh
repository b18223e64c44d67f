#!/bin/bash
# time the segment form (cfg2, +-512 lags) for several chunk counts; one bench line per value -> gpurun_out/seg_chunks.jsonl
out=gpurun_out/seg_chunks.jsonl
: > $out
for c in ${CHUNKS:-8 10 12 14 16 19 24 31}; do
  TDOA_SEG_CHUNKS=$c python bench.py --max-lag ${MAXLAG:-512} --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'chunks': $c, 'ms': d['ms_per_step'], 'graph_ms': (d.get('graph_replay') or {}).get('ms_per_step'), 'parity': d.get('parity_window0')}))" >> $out || exit 1
done
cat $out
