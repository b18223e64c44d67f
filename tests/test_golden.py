"""Committed golden vectors (tests/golden, produced by the oracle; see make_golden.py):
CPU: the oracle still reproduces them.  GPU: the C-ABI path reproduces them."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gold():
    g = json.load(open(os.path.join(HERE, "golden.json")))
    g["caps"] = [np.fromfile(os.path.join(HERE, g["files"][n]["file"]), dtype=np.uint8) for n in g["stations"]]
    g["fm_a"] = np.fromfile(os.path.join(HERE, "fm-a.dat"), dtype=np.uint8)
    g["fm_b"] = np.fromfile(os.path.join(HERE, "fm-b.dat"), dtype=np.uint8)
    for n, c in zip(g["stations"], g["caps"]):
        assert sha(c) == g["files"][n]["sha256"]
        assert c.size == 6 * g["block"]                      # .dat = 3 blocks x 2 bytes/sample
    return g


def _signals(g, api):
    data = [api.iq_u8_to_c64(c) if hasattr(api, "iq_u8_to_c64") else api.load_iq_u8(c) for c in g["caps"]]
    from oracle import pyoracle as o
    return [o.extract_reference(d) for d in data], [o.extract_target(d) for d in data]


def test_oracle_reproduces_golden(gold, oracle):
    g = gold
    for i, n in enumerate(g["stations"]):                    # simulators are deterministic
        raw = oracle.simulate_station(n, g["block"], oracle.SEED_BASE + i, tx_power=50000.0)
        assert sha(raw) == g["files"][n]["sha256"]
    refs, tgts = _signals(g, oracle)
    for kind, sigs in (("ref", refs), ("tgt", tgts)):
        for rec in g["mode_a"][kind]:
            i, j = rec["pair"]
            d, c = oracle.cross_correlate(sigs[i], sigs[j])
            assert d == rec["delay"] and c == float.fromhex(rec["corr"])
    for n, r, t in zip(g["stations"], refs, tgts):
        assert sha(oracle.preprocess(r)[0]) == g["mode_a"]["preprocess_sha256"][n]["ref"]
        assert sha(oracle.preprocess(t)[0]) == g["mode_a"]["preprocess_sha256"][n]["tgt"]
    pa, _ = oracle.b_preprocess(g["fm_a"][:12000])
    pb, _ = oracle.b_preprocess(g["fm_b"][:12000])
    lag, corr = oracle.b_xcorr_peak(pa, pb, g["max_lag"])
    assert lag == g["mode_b"]["fm_pair"]["lag"] == g["files"]["fm-b"]["delay"]
    assert corr == float.fromhex(g["mode_b"]["fm_pair"]["corr"])
    rec = g["mode_b"]["fm_pair_fine"]
    fine = oracle.b_refine_peak(pa, pb, lag, rec["gate"])
    assert fine["frac"] == float.fromhex(rec["frac"]) and fine["delay"] == float.fromhex(rec["delay"])
    assert [float(v) for v in fine["y"]] == [float.fromhex(v) for v in rec["y"]]
    assert fine["plausible"] == rec["plausible"]


@pytest.mark.gpu
def test_gpu_mode_a_reproduces_golden(gold, oracle):
    import tdoa_amd
    g = gold
    with tdoa_amd.Context() as c:
        refs, tgts = _signals(g, c)
        for n, r, t in zip(g["stations"], refs, tgts):
            pr, wr = c.preprocess(r)
            pt, wt = c.preprocess(t)
            assert sha(pr) == g["mode_a"]["preprocess_sha256"][n]["ref"]
            assert sha(pt) == g["mode_a"]["preprocess_sha256"][n]["tgt"]
            assert [wr, wt] == g["mode_a"]["preprocess_sha256"][n]["weak"]
        for kind, sigs in (("ref", refs), ("tgt", tgts)):
            for rec in g["mode_a"][kind]:
                i, j = rec["pair"]
                d, corr = c.cross_correlate(sigs[i], sigs[j])
                want = float.fromhex(rec["corr"])
                assert d == rec["delay"] and abs(corr - want) <= 1e-9 * max(abs(want), 1e-3)
        fu = g["mode_a"]["fm_unequal"]
        a, b = c.load_iq_u8(g["fm_a"])[:fu["n1"]], c.load_iq_u8(g["fm_b"])[:fu["n2"]]
        d, corr = c.cross_correlate(a, b)
        assert d == fu["delay"] and abs(corr - float.fromhex(fu["corr"])) <= 1e-9 * abs(corr)
        sig = c.load_iq_u8(g["fm_a"])[:5000]
        d, corr = c.simple_correlate(sig[100:3100], sig)
        assert d == g["simple_corr"]["delay"] and corr == float.fromhex(g["simple_corr"]["corr"])


@pytest.mark.gpu
def test_gpu_mode_b_reproduces_golden(gold):
    import tdoa_amd
    g = gold
    with tdoa_amd.Context(window_len=g["window_len"], max_lag=g["max_lag"]) as c:
        peaks = c.process_u8(g["caps"])
        assert peaks.shape == (len(g["mode_b"]["peaks"]), 3)
        for wid, row in enumerate(g["mode_b"]["peaks"]):
            for p, rec in enumerate(row):
                want = float.fromhex(rec["corr"])
                assert peaks[wid, p]["lag"] == rec["lag"]
                assert abs(peaks[wid, p]["corr"] - want) <= 1e-5 * abs(want)
        wpb = g["block"] // g["window_len"]
        for key, rec in g["mode_b"]["stats"].items():
            name, wid = key.split("/")
            wid = int(wid)
            cap = g["caps"][g["stations"].index(name)]
            off = (wid // wpb) * g["block"] + (wid % wpb) * g["window_len"]
            out, st = c.fm_preprocess(cap[2 * off:2 * (off + g["window_len"])])
            assert (st.s1, st.s2_lo, st.s2_hi) == (rec["s1"], rec["s2_lo"], rec["s2_hi"])
            assert float(st.mean) == float.fromhex(rec["mean"]) and float(st.scale) == float.fromhex(rec["scale"])
            assert sha(out) == rec["sha256"]
        lag, corr = c.fm_xcorr(g["fm_a"][:12000], g["fm_b"][:12000], g["max_lag"])
        want = float.fromhex(g["mode_b"]["fm_pair"]["corr"])
        assert lag == g["mode_b"]["fm_pair"]["lag"] and abs(corr - want) <= 1e-5 * abs(want)
        rec = g["mode_b"]["fm_pair_fine"]
        _, fine = c.fm_xcorr_fine(g["fm_a"][:12000], g["fm_b"][:12000], g["max_lag"], rec["gate"])
        assert abs(fine["frac"] - float.fromhex(rec["frac"])) < 1e-4
        assert max(abs(a - float.fromhex(b)) for a, b in zip(fine["y"], rec["y"])) <= 1e-5 * abs(want)
        assert fine["plausible"] == rec["plausible"]
