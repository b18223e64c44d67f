#!/usr/bin/env python3
"""Regenerates tests/golden/*.  The vectors come from the CPU oracle (oracle/tdoa_oracle.c):
the reference is Go, cannot run here and holds no numeric fixtures of its own, so these are
regression data for the oracle and expected values for the GPU path -- not reference output.
Run from the repo root:  python tests/golden/make_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as o  # noqa: E402

BLOCK = 4000          # samples per block -> 24 000-byte captures
WLEN = 2000
MAX_LAG = 150


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {"block": BLOCK, "window_len": WLEN, "max_lag": MAX_LAG, "stations": o.COLLECTORS, "files": {}}
    caps = []
    for i, name in enumerate(o.COLLECTORS):
        raw = o.simulate_station(name, BLOCK, o.SEED_BASE + i, tx_power=50000.0)
        fn = "sim-%s-1754900000.dat" % name                      # simulator.go:164-165 naming
        raw.tofile(os.path.join(HERE, fn))
        out["files"][name] = {"file": fn, "sha256": sha(raw)}
        caps.append(raw)
    # a pair with a true sample delay (the simulators only model carrier phase)
    fm_a = o.simulate_delayed_fm(3 * BLOCK, 0, 2024, 1)
    fm_b = o.simulate_delayed_fm(3 * BLOCK, 23, 2024, 2)
    fm_a.tofile(os.path.join(HERE, "fm-a.dat"))
    fm_b.tofile(os.path.join(HERE, "fm-b.dat"))
    out["files"]["fm-a"] = {"file": "fm-a.dat", "sha256": sha(fm_a)}
    out["files"]["fm-b"] = {"file": "fm-b.dat", "sha256": sha(fm_b), "delay": 23}

    # mode A: the reference's call pattern on these files (processor.go:756-850)
    data = [o.iq_u8_to_c64(c) for c in caps]
    refs = [o.extract_reference(d) for d in data]
    tgts = [o.extract_target(d) for d in data]
    mode_a = {"preprocess_sha256": {}, "ref": [], "tgt": []}
    for name, r, t in zip(o.COLLECTORS, refs, tgts):
        pr, weak_r = o.preprocess(r)
        pt, weak_t = o.preprocess(t)
        mode_a["preprocess_sha256"][name] = {"ref": sha(pr), "tgt": sha(pt), "weak": [bool(weak_r), bool(weak_t)]}
    for kind, sigs in (("ref", refs), ("tgt", tgts)):
        for i in range(3):
            for j in range(i + 1, 3):
                d, c = o.cross_correlate(sigs[i], sigs[j])
                mode_a[kind].append({"pair": [i, j], "delay": d, "corr": float(c).hex()})
    # lag search with unequal lengths through the full chain
    a, b = o.iq_u8_to_c64(fm_a)[:3000], o.iq_u8_to_c64(fm_b)[:6000]
    d, c = o.cross_correlate(a, b)
    mode_a["fm_unequal"] = {"n1": 3000, "n2": 6000, "delay": d, "corr": float(c).hex()}
    out["mode_a"] = mode_a

    # mode B: K1 statistics and peaks per (window, pair)
    mode_b = {"stats": {}, "peaks": [], "fm_pair": {}}
    wpb = BLOCK // WLEN
    for wid in range(3 * wpb):
        off = (wid // wpb) * BLOCK + (wid % wpb) * WLEN
        pre = []
        for name, c in zip(o.COLLECTORS, caps):
            p, st = o.b_preprocess(c[2 * off:2 * (off + WLEN)])
            pre.append(p)
            mode_b["stats"]["%s/%d" % (name, wid)] = {"s1": st.s1, "s2_lo": st.s2_lo, "s2_hi": st.s2_hi,
                                                      "mean": float(st.mean).hex(), "scale": float(st.scale).hex(),
                                                      "sha256": sha(p)}
        row = []
        for (i, j) in [(0, 1), (0, 2), (1, 2)]:
            lag, corr = o.b_xcorr_peak(pre[i], pre[j], MAX_LAG)
            row.append({"lag": lag, "corr": float(corr).hex()})
        mode_b["peaks"].append(row)
    pa, _ = o.b_preprocess(fm_a[:2 * 6000])
    pb, _ = o.b_preprocess(fm_b[:2 * 6000])
    lag, corr = o.b_xcorr_peak(pa, pb, MAX_LAG)
    mode_b["fm_pair"] = {"n": 6000, "lag": lag, "corr": float(corr).hex()}
    fine = o.b_refine_peak(pa, pb, lag, 20.0)     # gate 20 < lag 23: implausible on purpose
    mode_b["fm_pair_fine"] = {"gate": 20.0, "frac": float(fine["frac"]).hex(), "delay": float(fine["delay"]).hex(),
                              "y": [float(v).hex() for v in fine["y"]], "plausible": bool(fine["plausible"])}
    out["mode_b"] = mode_b

    # simple_corr / fast analyzer
    sig = o.iq_u8_to_c64(fm_a)[:5000]
    d, c = o.simple_correlate(sig[100:3100], sig)
    out["simple_corr"] = {"delay": d, "corr": float(c).hex()}
    rc, ra, ta = o.fast_analyze_capture(caps[0])
    out["fast_analyzer"] = {"ref_snr": float(ra.snr_estimate).hex(), "tgt_snr": float(ta.snr_estimate).hex(),
                            "ref_power": float(ra.power_level).hex(), "tgt_power": float(ta.power_level).hex()}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.json"))


if __name__ == "__main__":
    main()
