#!/usr/bin/env python3
"""Folds the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units) into per-kernel HBM-side
traffic per launch.  gfx950 correction: FETCH_SIZE counts 128-byte requests as 64 bytes for wide
coalesced loads.  Calibration on this pipeline's own kernels (known byte counts, no reuse possible):
  k_fwd_row4096 (8 B/lane float2 rows)  : FETCH_SIZE*1024 / bytes read = 0.500  -> x2
  k_inv_col_pruned (16 B/lane)          : 0.500 -> x2      k_fm_demod (16 B/lane): 0.504 -> x2
  k_fwd_col256_c16 (4 B/lane, 128-B runs): 0.94 -> x1 (left uncorrected)
  k_pair_decimate16 (8 B/lane float2, contiguous 4 KB runs of a tile): as k_fwd_row4096 -> x2 (checked: see DESIGN.md 6)
  WRITE_SIZE*1024 / bytes written = 1.000 for every kernel -> x1
usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> [config name]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (hash of the kernel sources the counters were taken on)

FETCH_CORRECTION = {"k_fm_demod": 2.0, "k_fwd_row4096": 2.0, "k_fwd_row4096_unpack": 2.0, "k_inv_row_pair4096": 2.0, "k_inv_col_pruned": 2.0,
                    "k_fwd_col256_c16": 1.0, "k_pair_decimate16": 2.0, "k_inv_rows_plain_r8": 2.0, "k_small_rows_col_peak": 2.0,      # (the same loads as k_inv_rows_plain_r8)
                    "k_inv_col_pruned_any": 2.0,
                    # round 3: k_fwd_col256_k1 reads the capture bytes (2 bytes per sample: 1.19 GB per cfg2 step) as 4 B/lane.
                    # Its first form (512 threads, 128-byte runs per half-wave, like k_fwd_col256_c16's codes) counted x1
                    # (FETCH_SIZE*1024 / bytes = 1.00); the 1024-thread form reads 256-byte runs per wave and counts like
                    # the wide streaming loads: 0.635 GB reported for the same 1.19 GB (ratio 0.534 = 1/2 + the boundary
                    # samples and the tables) -> x2
                    "k_fwd_col256_k1": 2.0, "k_fwd_col512_k1": 2.0,      # (the N2 = 512 column kernel: the same construction)
                    # round 4: the running sums of the window edges are streamed like k_fm_demod (16 B per lane)
                    "k_once_edges": 2.0,
                    # the column walk of the decimated pair step reads 8 B per lane along the rows, 512-byte runs per wave,
                    # like k_fwd_row4096
                    "k_pair_decimate_cols": 2.0,
                    # round 5: the finish sweep of the two-sweep column pass reads 16 bytes per lane with non-temporal loads
                    # (fft_radix16.hpp k_fwd_col_finish) and works in place -- it reads exactly what it writes, and its raw
                    # FETCH_SIZE came out at half of its WRITE_SIZE (68.85 GB against 137.8 GB per cfg3 step, round 4) -> x2
                    "k_fwd_col_finish": 2.0,
                    # round 5: the staged column walk brings its rows in by LDS-DMA, 16 bytes per lane (global_load_lds_dwordx4,
                    # 512-byte row pieces); calibrated on cfg3, where a window's three spectra can only come from memory once per
                    # workgroup group: raw FETCH_SIZE / bytes = 0.50 -> x2 (collect_round's cfg3 file)
                    "k_pair_decimate_staged": 2.0}


def load(d):
    out = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("tdoa::", "").replace("void ", "")
            out[name].append(float(r["Counter_Value"]))   # template instances keep their <...> suffix
    return out


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    # a step may take several launches of a kernel (launch groups of windows): per-step sums = all launches of the pass /
    # the steps of the pass (k_decode_peaks runs once per step; every step does the same work)
    steps_f = max(1, len(fetch.get("k_decode_peaks", [0.0])))
    steps_w = max(1, len(write.get("k_decode_peaks", [0.0])))
    res = {}
    for k in sorted(set(fetch) & set(write)):
        if not k.startswith("k_") or k.startswith("k_synth"):
            continue
        base = k.split("<")[0]                      # template instances of one kernel are summed
        f, w = sum(fetch[k]) / steps_f, sum(write[k]) / steps_w
        corr = FETCH_CORRECTION.get(base, 1.0)
        rec = res.setdefault(base, {"FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0, "fetch_correction": corr,
                                    "traffic_bytes_per_step": 0.0, "launches_per_step": 0.0})
        rec["FETCH_SIZE_KiB"] += f
        rec["WRITE_SIZE_KiB"] += w
        rec["traffic_bytes_per_step"] += f * 1024 * corr + w * 1024
        rec["launches_per_step"] = max(rec["launches_per_step"], len(fetch[k]) / steps_f)      # (instances of one scope launch together)
    for rec in res.values():
        rec["traffic_bytes_per_launch"] = rec["traffic_bytes_per_step"] / max(rec["launches_per_step"], 1.0)
    cfg = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --config %s; per-step sums over "
                         "the launch groups of a step" % cfg,
               "config": cfg, "steps_in_pass": steps_f,
               "source_sha16": source_hash(), "kernels": res}, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
