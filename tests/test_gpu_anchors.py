"""GPU: the timed mode-B path against its two independent anchors at BASELINE config-2 size
(L = 2 000 000, max_lag 20000, N = 2^21) -- VERDICT r01 item 1.

(i)  oracle/float_pipeline.py: float64 atan2 discriminator (the prebuilt binary's definition),
     f64 FFT correlation; no table, no integer code, nothing shared with the kernels or with ob_*.
(ii) processor.go:646-736 timeDomainCorrelation (o_time_domain_all_lags) on the same normalised signals, unequal
     lengths so that the lag search is not trivial; the FFT path restricted to [0, maxLag_eff) must pick the same index.
Measured deviations are printed (quoted in DESIGN.md section 5)."""
import numpy as np
import pytest

from oracle import float_pipeline as fp

pytestmark = pytest.mark.gpu

L, ML = 2_000_000, 20000


def _inputs(oracle):
    out = [("delayed_fm", oracle.simulate_delayed_fm(L, 0, 4242, 1), oracle.simulate_delayed_fm(L, 37, 4242, 2))]
    sim = [oracle.simulate_station(nm, L, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    weak = [oracle.simulate_weak_station(nm, L, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    for tag, caps in (("simulator.go", sim), ("weak_signal_simulator.go", weak)):
        for kind, off in (("ref", 0), ("tgt", L)):
            for (i, j) in ((0, 1), (0, 2), (1, 2)):
                out.append(("%s %s %d-%d" % (tag, kind, i, j), caps[i][2 * off:2 * (off + L)], caps[j][2 * off:2 * (off + L)]))
    return out


def test_gpu_vs_float_definition_full_size(oracle, capsys):
    """lag identical everywhere and corr within north_star's 1e-5 of the float64 definition on EVERY case, the simulators'
    +-1..3 LSB captures included (where the 16-bit phase code of rounds 1-2 measured up to 4.6e-5: see
    tests/test_mode_b_anchors.py); constant windows give (0, 0.0)"""
    import tdoa_amd
    rows = []
    with tdoa_amd.Context(max_lag=ML, window_len=L) as c:
        for name, a, b in _inputs(oracle):
            lag, corr = c.fm_xcorr(a, b, ML)
            # these anchors judge the TIMED path: single-look K1 (every capture byte read once).  A change that silently
            # routed them back through the statistics pre-pass would leave the numbers green and the claim empty.
            assert c.last_k1(0)[1], (name, "the anchor ran through the pre-pass, not the single-look path")
            flag, fcorr, _ = fp.xcorr_peak_u8(a, b, ML)
            assert lag == flag, (name, lag, flag)
            if fcorr == 0.0:
                assert corr == 0.0
                rows.append((name, lag, corr, 0.0))
                continue
            dev = abs(corr - fcorr) / abs(fcorr)
            rows.append((name, lag, corr, dev))
            assert dev < 1e-5, (name, dev)
    with capsys.disabled():
        print("\n  GPU mode B vs float64 atan2 pipeline, L = %d, max_lag %d" % (L, ML))
        for r in rows:
            print("    %-38s lag %6d  corr %13.6f  |dcorr|/|corr| %.2e" % r)


@pytest.mark.parametrize("blocks,ns,delay", [(1990, 2_000_000, 4321), (1999, 2_000_000, 0), (500, 600_000, 19999)])
def test_gpu_fft_path_vs_go_time_domain_correlation(oracle, blocks, ns, delay, capsys):
    """template of exactly B*1000 samples: the mode-B sum and processor.go:686-720 coincide term by term (the extra
    sample handed to the Go form is dropped by its block truncation, :691), so the FFT path on lags [0, maxLag_eff)
    must reproduce timeDomainCorrelation's values and its first-max index."""
    import tdoa_amd
    nt = blocks * 1000
    a = oracle.simulate_delayed_fm(nt, 0, 777, 1)
    b = oracle.simulate_delayed_fm(ns, delay, 777, 2)
    eff = max(1, min(ML, ns - (nt + 1)))                                  # processor.go:668-675
    with tdoa_amd.Context(max_lag=ML, window_len=L) as c:
        lags = c.fm_xcorr_lags(a, b, ML)[ML - 1:ML - 1 + eff]              # lags 0 .. eff-1
        vt, _ = c.fm_preprocess(a)                                         # the SAME preprocessed inputs (bit-equal to ob_*)
        vs, _ = c.fm_preprocess(b)
    wt, _ = oracle.b_preprocess(a)
    ws, _ = oracle.b_preprocess(b)
    assert np.array_equal(vt.view(np.uint32), wt.view(np.uint32)) and np.array_equal(vs.view(np.uint32), ws.view(np.uint32))
    t64 = np.concatenate([vt, [0.0]]).astype(np.complex64)
    go = oracle.time_domain_all_lags(t64, vs.astype(np.complex64), ML)
    assert go.size == eff
    err = np.abs(go - lags).max() / np.abs(go).max()
    assert err < 1e-5
    gd, gc = oracle.time_domain_correlation(t64, vs.astype(np.complex64), ML)
    assert gd == int(np.argmax(np.abs(lags))) == delay
    assert abs(gc - lags[gd]) <= 1e-5 * abs(gc)
    with capsys.disabled():
        print("\n  FFT path vs timeDomainCorrelation: %d lags, template %d, max |diff| %.2e of the peak, index %d" % (eff, nt, err, gd))


@pytest.mark.parametrize("nt,ns,delay", [(1_990_000, 2_000_000, 4321), (1_999_000, 2_000_000, 0), (500_000, 600_000, 19999),
                                         (600_000, 500_000, 777), (70_001, 90_000, 12345)])
def test_k5_go_lag_set_through_the_abi(oracle, nt, ns, delay, capsys):
    """tdoa_params.lag_mode = TDOA_LAGS_GO: template = the shorter input, its first B*1000 samples, lags [0, maxLag_eff),
    first strict maximum (processor.go:650-678, 691, 719-725).  The index the C ABI returns must EQUAL the one
    timeDomainCorrelation (oracle restatement, f64 time domain) picks on the same preprocessed inputs, the value within
    1e-5, and the lag array must be its lag array -- whichever input comes first."""
    import tdoa_amd
    # the LONGER input is the signal and carries the delay; (600 000, 500 000): the second input is the template
    a = oracle.simulate_delayed_fm(nt, delay if nt > ns else 0, 777, 1)
    b = oracle.simulate_delayed_fm(ns, delay if nt <= ns else 0, 777, 2)
    with tdoa_amd.Context(max_lag=ML, window_len=L, lag_mode=tdoa_amd.capi.LAGS_GO) as c:
        lag, corr = c.fm_xcorr(a, b, ML)
        lags = c.fm_xcorr_lags(a, b, ML)
        va, _ = c.fm_preprocess(a)
        vb, _ = c.fm_preprocess(b)
    gd, gc = oracle.time_domain_correlation(va.astype(np.complex64), vb.astype(np.complex64), ML)
    go = oracle.time_domain_all_lags(va.astype(np.complex64), vb.astype(np.complex64), ML)
    eff = max(1, min(ML, abs(ns - nt)))
    assert go.size == eff
    assert lag == gd == delay and abs(corr - gc) <= 1e-5 * abs(gc)
    got = lags[ML - 1:ML - 1 + eff]
    assert np.abs(got - go).max() <= 1e-5 * np.abs(go).max()
    assert not lags[:ML - 1].any() and not lags[ML - 1 + eff:].any()            # no other lag is searched or reported
    with capsys.disabled():
        print("\n  TDOA_LAGS_GO: template %d signal %d -> index %d (timeDomainCorrelation: %d), %d lags" % (min(nt, ns), max(nt, ns), lag, gd, eff))


def test_k5_go_lag_set_edge_cases(oracle):
    """equal lengths -> lag 0 only (the reference's own call pattern, processor.go:668-675); a template of at most one
    block -> (0, 0.0) (:708-717); tdoa_process windows all have one length: lag 0 everywhere, value = timeDomainCorrelation's"""
    import tdoa_amd
    n = 300_000
    a = oracle.simulate_delayed_fm(n, 0, 31, 1)
    b = oracle.simulate_delayed_fm(n, 41, 31, 2)
    with tdoa_amd.Context(max_lag=2000, window_len=n, lag_mode=tdoa_amd.capi.LAGS_GO) as c:
        lag, corr = c.fm_xcorr(a, b, 2000)
        va, _ = c.fm_preprocess(a)
        vb, _ = c.fm_preprocess(b)
        assert c.fm_xcorr(a[:2 * 1000], b, 2000) == (0, 0.0) and c.fm_xcorr(a, b[:2 * 700], 2000) == (0, 0.0)
        with pytest.raises(tdoa_amd.TdoaError):
            c.fm_xcorr_fine(a, b, 2000, 120.0)
    gd, gc = oracle.time_domain_correlation(va.astype(np.complex64), vb.astype(np.complex64), 2000)
    assert lag == gd == 0 and abs(corr - gc) <= 1e-5 * abs(gc)
    blk, wl = 140_000, 70_000
    caps = [np.concatenate([oracle.simulate_delayed_fm(blk, 100 + d, 310 + k, 10 * s + k) for k in range(3)])
            for s, d in enumerate((0, 41, -17))]
    with tdoa_amd.Context(max_lag=20000, window_len=wl, lag_mode=tdoa_amd.capi.LAGS_GO) as c:
        peaks = c.process_u8(caps)
        assert c.graph_info()["roots"] == 1
        again = c.process()                                                       # replayed graph
    assert peaks.shape == (6, 3) and not peaks["lag"].any() and np.array_equal(again, peaks)
    for wid in (0, 3, 5):
        off = (wid // 2) * blk + (wid % 2) * wl
        pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0].astype(np.complex64) for cp in caps]
        for p, (i, j) in enumerate(((0, 1), (0, 2), (1, 2))):
            gd, gc = oracle.time_domain_correlation(pre[i], pre[j], 20000)
            # lag 0 is not where these windows correlate (their delays are 41, -17, -58 samples): the value there is noise
            # of order 1 against a full-scale sqrt(69 000) = 263, so the bound is 1e-6 of full scale
            assert gd == 0 and abs(peaks[wid, p]["corr"] - gc) <= 1e-6 * np.sqrt(69000.0), (wid, p, peaks[wid, p], gc)


def test_k5_go_lag_set_on_the_timed_batch_plan(oracle, capsys):
    """tdoa_process with lag_mode = TDOA_LAGS_GO at the TIMED geometry: windows of 2 000 000 samples, max_lag 20000 -- the
    4096 x 256 plan, K1 fused into the column pass, full inverse restricted to lag 0 -- on three simulator.go stations:
    every window has one length, so timeDomainCorrelation evaluates lag 0 only (processor.go:668-678, 772-780) over the
    first 1999 blocks of 1000 samples (:691); every (window, pair) must return (0, corr) with corr = the oracle's
    timeDomainCorrelation on the SAME preprocessed windows to 1e-5 (of full scale where the value at lag 0 is noise)."""
    import tdoa_amd
    blk, wl = 4_000_000, 2_000_000
    caps = [oracle.simulate_station(nm, blk, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    with tdoa_amd.Context(max_lag=ML, window_len=wl, lag_mode=tdoa_amd.capi.LAGS_GO) as c:
        peaks = c.process_u8(caps)
        assert tuple(c.plan_info())[1:] == (4096, 256) and c.graph_info()["roots"] == 1
        again = c.process()                                                       # the replayed step graph
        assert not c.last_k1(0)[1]                                                # (statistics over the whole windows: pre-pass)
    assert peaks.shape == (6, 3) and not peaks["lag"].any() and np.array_equal(again, peaks)
    worst = 0.0
    for wid in (0, 2, 3, 5):                                                      # reference, target and second reference block
        off = (wid // 2) * blk + (wid % 2) * wl
        pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0].astype(np.complex64) for cp in caps]
        for p, (i, j) in enumerate(((0, 1), (0, 2), (1, 2))):
            gd, gc = oracle.time_domain_correlation(pre[i], pre[j], ML)
            assert gd == 0
            # the value at lag 0 of these captures is noise of order 1 against a full scale of sqrt(1 999 000) = 1414
            err = abs(peaks[wid, p]["corr"] - gc)
            worst = max(worst, err)
            assert err <= 1e-5 * abs(gc) + 1e-6 * np.sqrt(1_999_000.0), (wid, p, peaks[wid, p], gc)
    with capsys.disabled():
        print("\n  TDOA_LAGS_GO on the 4096 x 256 batch plan: 12 (window, pair) units at lag 0, worst |dcorr| %.2e (full scale 1414)" % worst)


def test_k1_exact_reversals_bit_exact(oracle):
    """exactly reversed samples (+pi), collinear reversals of different magnitude, and non-collinear samples whose
    angle codes are exactly opposite (sign of Im p decides), in the vector fast path and in the window head / tail"""
    import tdoa_amd
    rng = np.random.default_rng(5)
    iq = rng.integers(0, 256, size=2 * 70000, dtype=np.uint8)
    special = [129, 129, 126, 126, 128, 126, 127, 129, 128, 128, 126, 126, 129, 129, 127, 127, 254, 253, 0, 1, 234, 229, 0, 6]
    for off in (0, 2, 4096, 2 * 512 * 40 + 10, 2 * 70000 - len(special)):
        iq[off:off + len(special)] = special
    with tdoa_amd.Context(max_lag=100, window_len=70000) as c:
        for lo, n in ((0, 70000), (2, 69999), (4096, 30001)):
            got, st = c.fm_preprocess(iq[lo:lo + 2 * n])
            want, ost = oracle.b_preprocess(iq[lo:lo + 2 * n])
            assert (st.s1, st.s2_lo, st.s2_hi) == (ost.s1, ost.s2_lo, ost.s2_hi)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        sim = oracle.simulate_station("n3pay", 50000, 3)[:100000]          # +-1..3 LSB: reversals everywhere
        got, st = c.fm_preprocess(sim)
        want, ost = oracle.b_preprocess(sim)
        assert (st.s1, st.s2_lo, st.s2_hi) == (ost.s1, ost.s2_lo, ost.s2_hi) and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_optional_k1_smoothing_on_the_gpu(oracle):
    """tdoa_params.k1_smooth = 10 (the prebuilt binary's applyLowPassFilter(10) on the discriminator output): smoothed,
    normalised samples and statistics bit-equal to ob_smooth_codes; end to end against the float64 chain in the binary's
    order; default contexts are untouched"""
    import tdoa_amd
    n, ml = 300_000, 2000
    a = oracle.simulate_delayed_fm(n, 0, 4242, 1)
    b = oracle.simulate_delayed_fm(n - 1001, 37, 4242, 2)                 # ragged: the tail chunk of k_k1_smooth
    with tdoa_amd.Context(max_lag=ml, window_len=n, k1_smooth=10) as c:
        for x in (a, b, a[:2 * 4099], a[:2 * 7]):
            got, st = c.fm_preprocess(x)
            want, ost = oracle.b_preprocess_smooth(x, 10)
            assert (st.s1, st.s2_lo, st.s2_hi) == (ost.s1, ost.s2_lo, ost.s2_hi)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        lag, corr = c.fm_xcorr(a, b, ml)
        lags = c.fm_xcorr_lags(a, b, ml)
        c.debug_flags(no_short_lag=True)
        lag_g, corr_g = c.fm_xcorr(a, b, ml)
    ta, _ = oracle.b_preprocess_smooth(a, 10)
    tb, _ = oracle.b_preprocess_smooth(b, 10)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, ml)
    assert lag == lag_g == olag == 37
    assert abs(corr - ocorr) <= 1e-5 * abs(ocorr) and abs(corr_g - ocorr) <= 1e-5 * abs(ocorr)
    assert np.abs(lags - want).max() <= 1e-5 * np.abs(want).max()
    flag, fcorr, _ = fp.xcorr_peak_u8(a, b, ml, smooth=10)
    assert lag == flag and abs(corr - fcorr) <= 1e-5 * abs(fcorr)
    with tdoa_amd.Context(max_lag=ml, window_len=n) as c:                  # k1_smooth = 0: the unsmoothed pipeline
        lag0, corr0 = c.fm_xcorr(a, b, ml)
    assert lag0 == 37 and abs(corr0) < abs(corr)                           # the message is low-pass: smoothing removes noise


def test_optional_k1_power_gate_on_the_gpu(oracle):
    """tdoa_params.k1_gate = 1 (the prebuilt binary's preprocessSignal gate): windows of mean power <= 0.01 carry envelope
    codes, bit-equal to ob_preprocess_gate_u8; stronger windows are untouched; end to end through tdoa_process against the
    oracle and the float64 envelope chain; with k1_smooth the envelope windows are not smoothed"""
    import tdoa_amd
    n, ml = 300_000, 2000
    mid_a, mid_b = fp.am_capture(n, 0, 0.07, 5, 1), fp.am_capture(n - 1001, 91, 0.07, 5, 2)
    strong = oracle.simulate_delayed_fm(n, 0, 4242, 1)
    with tdoa_amd.Context(max_lag=ml, window_len=n, k1_gate=1) as c:
        for x, want_cls in ((mid_a, 1), (mid_b, 1), (strong, 0), (mid_a[:2 * 4099], 1), (mid_a[:2 * 7], 1)):
            got, st = c.fm_preprocess(x)
            want, ost, cls = oracle.b_preprocess_gate(x)
            assert cls == want_cls
            assert (st.s1, st.s2_lo, st.s2_hi) == (ost.s1, ost.s2_lo, ost.s2_hi)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        lag, corr = c.fm_xcorr(mid_a, mid_b, ml)
        c.debug_flags(no_short_lag=True)
        lag_g, corr_g = c.fm_xcorr(mid_a, mid_b, ml)
    ta, _, _ = oracle.b_preprocess_gate(mid_a)
    tb, _, _ = oracle.b_preprocess_gate(mid_b)
    olag, ocorr, _ = oracle.b_xcorr_peak_fft(ta, tb, ml)
    assert lag == lag_g == olag == 91
    assert abs(corr - ocorr) <= 1e-5 * abs(ocorr) and abs(corr_g - ocorr) <= 1e-5 * abs(ocorr)
    flag, fcorr, _ = fp.xcorr_peak_u8(mid_a, mid_b, ml, gate=True)
    assert lag == flag and abs(corr - fcorr) <= 1e-5 * abs(fcorr)
    # batched: three stations whose windows fall on both sides of the gate (blocks 1 and 3 moderate, block 2 strong)
    blk, wl = 140_000, 70_000
    caps = []
    for s, d in enumerate((0, 41, -17)):
        caps.append(np.concatenate([fp.am_capture(blk, 100 + d, 0.06, 7, s), oracle.simulate_delayed_fm(blk, 100 + d, 77, s),
                                    fp.am_capture(blk, 100 + d, 0.05, 8, s)]))
    for smooth in (0, 10):
        with tdoa_amd.Context(max_lag=300, window_len=wl, k1_gate=1, k1_smooth=smooth) as c:
            peaks = c.process_u8(caps)
        assert peaks.shape == (6, 3)
        for wid in range(6):
            off = (wid // 2) * blk + (wid % 2) * wl
            pre = [oracle.b_preprocess_gate(cp[2 * off:2 * (off + wl)], window=smooth) for cp in caps]
            assert all(p[2] == (0 if wid in (2, 3) else 1) for p in pre)
            for p, (i, j) in enumerate(((0, 1), (0, 2), (1, 2))):
                olag, ocorr = oracle.b_xcorr_peak(pre[i][0], pre[j][0], 300)
                assert peaks[wid, p]["lag"] == olag == (0, 41, -17)[j] - (0, 41, -17)[i], (smooth, wid, p)
                assert abs(peaks[wid, p]["corr"] - ocorr) <= 1e-5 * abs(ocorr)
    with tdoa_amd.Context(max_lag=ml, window_len=n) as c:                  # k1_gate = 0: the discriminator for every window
        got, _ = c.fm_preprocess(mid_a)
        assert np.array_equal(got.view(np.uint32), oracle.b_preprocess(mid_a)[0].view(np.uint32))
