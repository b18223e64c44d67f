#!/bin/bash
# K1 on realistic capture bytes: bench lines + LDS / VALU counters of the kernels that evaluate the
# discriminator (the fused column kernel; the edge sums of the single-look path), for the config's simulator bytes (+-1..3 LSB), half-scale FM carriers and uniform random bytes.
# usage (GPU box, repo root): bash scripts/collect_k1_bytes.sh <tag>
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for sim in config fm random; do
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph-leg --no-h2d --no-clocks --sim $sim > gpurun_out/${TAG}_k1bytes_${sim}_bench.json 2> gpurun_out/${TAG}_k1bytes_${sim}.err
  BENCH_ARGS="--sim $sim" bash scripts/collect_sq.sh ${TAG}k1$sim "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" > gpurun_out/${TAG}_k1bytes_${sim}_sq.csv
done
python3 - "$TAG" <<'PY'
import csv, json, sys
tag = sys.argv[1]
rows = []
for sim in ("config", "fm", "random"):
    d = json.loads(open("gpurun_out/%s_k1bytes_%s_bench.json" % (tag, sim)).read().strip().splitlines()[-1])
    ms = d["roofline"]["kernels_ms_per_step"]
    lines = open("gpurun_out/%s_k1bytes_%s_sq.csv" % (tag, sim)).read().strip().splitlines()
    sq = {r["kernel"].split("<")[0]: r for r in csv.DictReader(lines[1:])}
    rec = {"capture_bytes": sim, "ms_per_step": d["ms_per_step"], "k_fm_demod_ms": ms["k_fm_demod"], "k_fwd_col_ms": ms["k_fwd_col"]}
    for k in ("k_once_edges", "k_fm_demod", "k_fwd_col256_k1"):      # (k_fm_demod: only with TDOA_NO_K1_ONCE=1)
        if k not in sq:
            continue
        r = sq[k]
        rec[k] = {"lds_conflict_frac": round(float(r["SQ_LDS_BANK_CONFLICT"]) / max(float(r["SQ_LDS_IDX_ACTIVE"]), 1.0), 3),
                  "valu_issue_frac": round(float(r["SQ_ACTIVE_INST_VALU"]) * 4 / 1024 / (float(r["SQ_BUSY_CYCLES"]) / 32), 3)}
    rows.append(rec)
open("gpurun_out/%s_k1_bytes.json" % tag, "w").write("\n".join(json.dumps(r) for r in rows) + "\n")
print("\n".join(json.dumps(r) for r in rows))
PY
