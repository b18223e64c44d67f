"""CPU: the two independent anchors of the mode-B oracle.

(i)  oracle/float_pipeline.py -- the discriminator as read off the prebuilt reference binary
     (atan2(Im p, Re p), float64, no table, no integer code) carried end to end
     through mean / scale / f64 FFT correlation / peak pick -- against the ob_* oracle the device
     kernels are bit-exact with.
(ii) timeDomainCorrelation as processor.go:646-736 executes it (o_time_domain_all_lags: 1000-sample
     blocks, f64 block sums, sqrt(numBlocks*1000) gain, lags [0, maxLag_eff), first strict max) against
     the mode-B correlation restricted to the same lags, on the SAME normalised real signals.

The GPU versions at BASELINE sizes are in tests/test_gpu_anchors.py."""
import numpy as np
import pytest

from oracle import float_pipeline as fp

L, ML = 200_000, 3000


def _three_inputs(oracle, n):
    """(name, template bytes, signal bytes): a true sample delay, simulator.go and weak_signal_simulator.go captures"""
    out = [("delayed_fm", oracle.simulate_delayed_fm(n, 0, 4242, 1), oracle.simulate_delayed_fm(n, 37, 4242, 2))]
    sim = [oracle.simulate_station(nm, n, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    weak = [oracle.simulate_weak_station(nm, n, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    for tag, caps in (("simulator.go", sim), ("weak_signal_simulator.go", weak)):
        for kind, off in (("ref", 0), ("tgt", n)):
            for (i, j) in ((0, 1), (0, 2), (1, 2)):
                out.append(("%s %s %d-%d" % (tag, kind, i, j), caps[i][2 * off:2 * (off + n)], caps[j][2 * off:2 * (off + n)]))
    return out


def test_code_pipeline_vs_float_definition(oracle, capsys):
    """peak lag identical on every input and corr within north_star's 1e-5 of the float64 definition EVERYWHERE --
    also on the simulators' +-1..3 LSB captures, where few distinct I/Q values make a code's rounding error a fixed
    function of the sample pair that does not average out (the 16-bit code of rounds 1-2 cost up to 5e-5 there;
    the 24-bit code, pi/2^23 per step, measures below 1e-6).  Measured values are printed."""
    rows = []
    for name, a, b in _three_inputs(oracle, L):
        ta, _ = oracle.b_preprocess(a)
        tb, _ = oracle.b_preprocess(b)
        olag, ocorr, oc = oracle.b_xcorr_peak_fft(ta, tb, ML)
        flag, fcorr, fc = fp.xcorr_peak_u8(a, b, ML)
        assert olag == flag, name
        if fcorr == 0.0:                       # weak-simulator reference blocks: constant bytes, zero phase
            assert ocorr == 0.0 and not oc.any()
            rows.append((name, olag, fcorr, 0.0, 0.0))
            continue
        dev = abs(ocorr - fcorr) / abs(fcorr)
        dev_all = np.abs(oc - fc).max() / abs(fcorr)
        rows.append((name, olag, fcorr, dev, dev_all))
        assert dev < (1e-6 if name == "delayed_fm" else 1e-5), (name, dev)
    with capsys.disabled():
        print("\n  24-bit-code oracle vs float64 atan2 pipeline, L = %d, max_lag %d" % (L, ML))
        for r in rows:
            print("    %-38s lag %6d  corr %12.6f  |dcorr|/|corr| %.2e  max|dc|/|corr| %.2e" % r)


def test_float_pipeline_conventions():
    assert fp.pick_peak(np.array([0.0, -2.0, 1.0, 2.0, 0.0]), 3) == (1, 2.0)        # tie -> positive lag
    assert fp.pick_peak(np.array([2.0, 0.0, 1.0, 0.0, 2.0]), 3) == (2, 2.0)
    assert fp.pick_peak(np.array([2.0, 0.0, 2.0, 0.0, -3.0]), 3) == (2, -3.0)
    assert fp.pick_peak(np.array([np.nan, 0.0, 0.0, 0.0, 0.0]), 3) == (0, 0.0)      # NaN never wins; all-zero -> (0, 0.0)
    assert fp.pick_peak(np.array([1.0, 1.0, 1.0, 1.0, 1.0]), 3) == (0, 1.0)         # tie -> smaller |lag|
    y = fp.discriminate(np.array([128, 128, 126, 126, 128, 128], np.uint8))        # (1,1) -> (-3,-3) -> (1,1)
    assert y[1] == np.pi and y[2] == np.pi and y[0] == y[1]
    assert fp.discriminate(np.zeros(0, np.uint8)).size == 0 and fp.discriminate(np.array([1, 2], np.uint8))[0] == 0.0
    v = fp.normalise(np.array([1.0, 2.0, 3.0, 4.0]))
    assert abs(v.mean()) < 1e-15 and abs((v * v).mean() - 1.0) < 1e-15
    assert not fp.normalise(np.ones(5)).any()
    t, s = np.array([1.0, 2.0, 3.0]), np.array([0.0, 1.0, 2.0, 3.0, 0.0])
    c = fp.xcorr_lags(t, s, 3)                                                        # template found at lag +1
    assert np.allclose(c * np.sqrt(3.0), [0.0, 3.0, 8.0, 14.0, 8.0], atol=1e-12)         # lags -2 .. +2
    assert fp.pick_peak(c, 3)[0] == 1


@pytest.mark.parametrize("blocks,ns,delay", [(20, 26000, 137), (7, 9500, 0), (33, 40000, 1999)])
def test_mode_b_correlation_vs_go_time_domain(oracle, blocks, ns, delay):
    """SURVEY section 7 'hard parts': mode B restricted to [0, maxLag_eff) must pick the index timeDomainCorrelation
    picks on the same preprocessed inputs.  With a template of exactly B*1000 samples (+1, which the block
    truncation of processor.go:691 drops) the two definitions coincide term by term:
    corr[d] = (B*1000)^(-1/2) sum_{i < B*1000} t_i s_{i+d}."""
    nt = blocks * 1000
    a = oracle.simulate_delayed_fm(nt, 0, 777, 1)
    b = oracle.simulate_delayed_fm(ns, delay, 777, 2)
    vt, _ = oracle.b_preprocess(a)
    vs, _ = oracle.b_preprocess(b)
    max_lag = 20000                                                       # processor.go:633
    eff = max(1, min(max_lag, ns - (nt + 1)))                             # processor.go:668-675
    go = oracle.time_domain_all_lags(np.concatenate([vt, [0.0]]).astype(np.complex64), vs.astype(np.complex64), max_lag)
    assert go.size == eff
    mb = oracle.b_xcorr_all_lags(vt, vs, eff)[eff - 1:]                   # lags 0 .. eff-1
    assert np.abs(go - mb).max() <= 1e-6 * np.abs(go).max()
    gd, gc = oracle.time_domain_correlation(np.concatenate([vt, [0.0]]).astype(np.complex64), vs.astype(np.complex64), max_lag)
    assert gd == int(np.argmax(np.abs(mb))) == delay                     # first strict max == argmax (no exact ties here)
    assert abs(gc - mb[gd]) <= 1e-6 * abs(gc)


def test_optional_k1_smoothing_vs_the_binarys_chain(oracle, capsys):
    """tdoa_params.k1_smooth = 10: discriminator -> removeDCBias -> applyLowPassFilter(10) -> normalizeSignal is the prebuilt
    binary's strong-signal chain.  The integer smoothing of the phase codes (ob_smooth_codes, what the
    device runs) against that chain in ITS order in float64: same peak lag, corr within 1e-5 (integer rounding of the
    average + the edge samples' share of the mean)."""
    n, ml = 200_000, 3000
    # the integer filter itself: exact average, rounded half up, edges truncated
    code = oracle.b_discriminate(oracle.simulate_delayed_fm(5000, 0, 9, 1)).astype(np.int64)
    want = np.array([np.floor((2 * code[max(i - 5, 0):i + 6].sum() + len(code[max(i - 5, 0):i + 6])) /
                              (2 * len(code[max(i - 5, 0):i + 6]))) for i in range(code.size)])
    pre, st = oracle.b_preprocess_smooth(oracle.simulate_delayed_fm(5000, 0, 9, 1), 10)
    lp = np.rint(pre.astype(np.float64) / st.scale + st.mean)
    assert np.array_equal(lp, want)
    assert np.abs(fp.lowpass(code.astype(np.float64), 10) - want).max() <= 0.5 + 1e-9
    rows = []
    for name, a, b in _three_inputs(oracle, n)[:7]:
        ta, _ = oracle.b_preprocess_smooth(a, 10)
        tb, _ = oracle.b_preprocess_smooth(b, 10)
        olag, ocorr, _ = oracle.b_xcorr_peak_fft(ta, tb, ml)
        flag, fcorr, _ = fp.xcorr_peak_u8(a, b, ml, smooth=10)
        assert olag == flag, name
        dev = abs(ocorr - fcorr) / abs(fcorr)
        rows.append((name, olag, fcorr, dev))
        assert dev < 1e-5, (name, dev)
    # smoothing is a low-pass: it must not move the true delay, and it raises the peak of a low-pass message
    assert rows[0][1] == 37
    with capsys.disabled():
        print("\n  k1_smooth = 10: integer-smoothed codes vs float64 chain in the binary's order")
        for r in rows:
            print("    %-38s lag %6d  corr %12.6f  |dcorr|/|corr| %.2e" % r)


def test_optional_k1_power_gate_vs_the_binarys_envelope_branch(oracle, capsys):
    """tdoa_params.k1_gate = 1: windows of mean power <= 0.01 take envelope -> removeDCBias -> normalizeSignal, as the
    prebuilt binary's preprocessSignal does.  The integer envelope codes (what the device runs)
    against sqrt(re^2 + im^2) in float64: same class, same peak lag, corr within 1e-5 (the code resolves |x| to
    1/16384 of half an LSB: the 1/90 of rounds 1-2 cost 1.1e-4 at the low end of the class, |x| ~ 4 LSB, where the
    rounding is the same function of the bytes at both stations and does not average out like noise)."""
    # the code itself: round-half-up(16384 sqrt(m)) for every byte pair, monotone in |x|
    grid = [(i, q) for i in range(0, 256, 5) for q in range(0, 256, 7)] + [(0, 0), (255, 255), (127, 128), (128, 127)]
    for i, q in grid:
        m = (2 * i - 255) ** 2 + (2 * q - 255) ** 2
        assert oracle.b_envelope_code(i, q) == int(np.floor(16384.0 * np.sqrt(np.float64(m)) + 0.5))
    assert oracle.b_envelope_code(0, 0) == 5908471 and oracle.b_envelope_code(127, 128) == 23170
    n, ml = 200_000, 3000
    rows = []
    for amp, want_cls in ((0.03, 1), (0.07, 1), (0.09, 1), (0.3, 0)):
        a, b = fp.am_capture(n, 0, amp, 5, 1), fp.am_capture(n, 91, amp, 5, 2)
        ta, _, ca = oracle.b_preprocess_gate(a)
        tb, _, cb = oracle.b_preprocess_gate(b)
        assert ca == cb == want_cls == int(fp.mean_power(a) <= 0.01), (amp, fp.mean_power(a))
        olag, ocorr, _ = oracle.b_xcorr_peak_fft(ta, tb, ml)
        flag, fcorr, _ = fp.xcorr_peak_u8(a, b, ml, gate=True)
        dev = abs(ocorr - fcorr) / abs(fcorr)
        rows.append((amp, fp.mean_power(a), want_cls, olag, fcorr, dev))
        assert olag == flag, amp
        if want_cls:
            assert olag == 91, amp               # the envelope carries the common message
        assert dev < 1e-5, (amp, dev)
    # gate off, or a strong window: exactly the ungated path
    strong = oracle.simulate_delayed_fm(50_000, 0, 9, 1)
    g, _, cls = oracle.b_preprocess_gate(strong)
    assert cls == 0 and np.array_equal(g, oracle.b_preprocess(strong)[0])
    weak = fp.am_capture(50_000, 0, 0.05, 6, 1)
    assert np.array_equal(oracle.b_preprocess_gate(weak, gate=0)[0], oracle.b_preprocess(weak)[0])
    with capsys.disabled():
        print("\n  k1_gate = 1: integer envelope codes vs float64 envelope chain")
        for r in rows:
            print("    amp %.3f  mean power %.5f  class %d  lag %5d  corr %10.6f  |dcorr|/|corr| %.2e" % r)


def test_spectrum_decimation_math_in_float64(oracle, capsys):
    """the decimated inverse (k_pair_decimate16) restated in numpy float64 with the library's
    filter design: G[j] = sum_t h[t] Q[16 j + t], h = sinc(t/16) x Kaiser(126 dB, 191 taps) rounded to f32; the Nc/16-point
    inverse of G divided by w[m] = sum_t h[t] cos(2 pi t m / Nc) / 16 must reproduce the packed lags q[m] of the full
    inverse for |m| <= max_lag/2 + 2; what is left is the stop-band leakage of lags beyond Nc/16 - m"""
    L, N, D, ML = 2_000_000, 1 << 21, 16, 20000
    nc, r = N // 2, N // 2 // D
    mp = ML // 2 + 2
    dw = 2 * np.pi * (r - 2 * mp) / nc
    th = int(np.ceil((140.0 - 8.0) / (2.285 * dw) / 2.0))
    assert th == 106                                      # what 140 dB would take (rounds 2-3: 14 steps of 16 phases)
    th = min(th, 95)                                      # round 4: 12 steps per phase hold |t| <= 95 ...
    att = 8.0 + 2.285 * dw * 2 * th                       # ... and buy this much on the band (tdoa_mi355x.hip decimation_design)
    assert 126.0 < att < 127.0
    beta = 0.1102 * (att - 8.7)
    t = np.arange(-th, th + 1)
    h = (np.sinc(t / D) * np.i0(beta * np.sqrt(1.0 - (t / th) ** 2)) / np.i0(beta)).astype(np.float32).astype(np.float64)
    m = np.arange(-mp, mp + 1)
    w = (h[None, :] * np.cos(2 * np.pi * t[None, :] * m[:, None] / nc)).sum(axis=1) / D
    assert np.abs(w - 1.0).max() < 2e-6                   # the pass band is flat; it is divided out anyway
    rows = []
    sim = [oracle.simulate_station(nm, L, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    cases = [("delayed_fm", oracle.simulate_delayed_fm(L, 0, 4242, 1), oracle.simulate_delayed_fm(L, 37, 4242, 2)),
             ("simulator.go ref 0-1", sim[0][:2 * L], sim[1][:2 * L])]
    for name, a, b in cases:
        ta, tb = oracle.b_preprocess(a)[0].astype(np.float64), oracle.b_preprocess(b)[0].astype(np.float64)
        c = np.fft.irfft(np.conj(np.fft.rfft(ta, N)) * np.fft.rfft(tb, N), N)
        q = c[0::2] + 1j * c[1::2]                        # packed lags; Q = their Nc-point spectrum (what K3 produces)
        Q = np.fft.fft(q)
        idx = np.arange(r) * D
        G = np.zeros(r, dtype=complex)
        for tt, ht in zip(t, h):
            G += ht * Q[(idx + tt) % nc]
        est = (np.fft.ifft(G) * r)[m % r] / w
        ref = (q * nc)[m % nc]
        err = np.abs(est - ref).max() / np.abs(ref).max()
        rows.append((name, 0.0, err))
        assert err < 1.5e-6, (name, err)                  # measured: 6e-9 (FM pair), 7e-7 (noise-level peak)
    with capsys.disabled():
        print("\n  spectrum decimation 16:1, %d taps (float64): max error of the %d packed lags, relative to the peak" % (2 * th + 1, m.size))
        for name, _, err in rows:
            print("    %-24s %.2e" % (name, err))


def _column_walk_float64(Q, n2, n1, tab, steps=12):
    """csrc/dec_stream.hpp (k_pair_decimate_cols + the merge in k_inv_rows_plain_r8) restated line by line in float64: the
    FIR G[j] = sum_t h[t] Q[16 j + t] evaluated as a stencil DOWN the columns of Q[k2][k1] (k = k2 + N2 k1).  A walker
    owns column c downwards (rows 0 .. N2-1) and column N1-1-c upwards (rows N2-1 .. 0), twelve ring accumulators each; what a
    column's bins add to the six outputs next to it in the neighbouring columns goes to X[12][N1] and is merged afterwards.
    tab[p][s]: the tap t = 16 (s - C) + p, zero where |t| > T -- the table ensure_decimation uploads."""
    C, S, ng = steps // 2, steps, n2 // 16
    col = Q.reshape(n1, n2)                                   # col[c][k2] = Q[k2 + N2 c]
    G = np.zeros((ng, n1), dtype=complex)                     # the small plan's layout: output i of column c at [i][c]
    X = np.zeros((12, n1), dtype=complex)
    for c in range(n1 // 2):
        cm = n1 - 1 - c
        # downward walk of column c: row (g, p) -> output g + C - s in at[s], tap (p, s)
        at = np.zeros(S, dtype=complex)

        def top_leaves(i):
            nonlocal at
            if i < 0:
                X[6 + i, c] = at[S - 1]
            elif i < ng:
                G[i, c] = at[S - 1]
            else:
                X[6 + i - ng, c] = at[S - 1]
            at = np.concatenate(([0.0], at[:-1]))

        for k2 in range(n2):
            p, g = k2 & 15, k2 >> 4
            at += tab[p, :S] * col[c, k2]
            if p == 15:
                top_leaves(g - (S - 1 - C))
        for n in range(S - 1):
            top_leaves(ng - (S - 1 - C) + n)
        # upward walk of column cm: row N2 - k2 (k2 = 1 .. N2 - 1), then row 0; slot u holds output gb - (C - 1) + u
        ab = np.zeros(S, dtype=complex)

        def bottom_leaves(i):
            nonlocal ab
            if i >= ng:
                X[6 + i - ng, cm] = ab[S - 1]
            elif i >= 0:
                G[i, cm] = ab[S - 1]
            else:
                X[6 + i, cm] = ab[S - 1]
            ab = np.concatenate(([0.0], ab[:-1]))

        for k2 in range(1, n2):
            p, g = k2 & 15, k2 >> 4
            q = col[cm, n2 - k2]
            if p:
                ab += tab[p, :S] * q                          # phase 16 - p of its own group: the same twelve taps (symmetry)
            else:
                ab += tab[0, :S][::-1] * q                    # phase 0 of group NG - g, that group's last row
                bottom_leaves(ng - g + C)
        ab += tab[0, :S][::-1] * col[cm, 0]
        for n in range(S):
            bottom_leaves(C - n)
    # k_inv_rows_plain_r8, by_column: row i < 6 gets slot row 6 + i of the column to the left, row i >= NG - 6 slot row
    # i - (NG - 6) of the column to the right
    out = G.copy()
    for i in range(6):
        out[i] += np.roll(X[6 + i], 1)
    for i in range(ng - 6, ng):
        out[i] += np.roll(X[i - (ng - 6)], -1)
    return out.T.reshape(-1)                                  # G[(N2/16) c + i]


@pytest.mark.parametrize("n2,n1,th", [(256, 8, 95), (256, 6, 75), (512, 4, 87)])
def test_column_walk_bookkeeping_in_float64(n2, n1, th):
    """the decimated pair step as a column walk (dec_stream.hpp): rings, the tap symmetry the upward walk relies on, the
    neighbour shares and their merge reproduce the plain circular FIR G[j] = sum_t h[t] Q[16 j + t] to float64 rounding"""
    rng = np.random.default_rng(n2 + n1)
    nc = n2 * n1
    Q = rng.standard_normal(nc) + 1j * rng.standard_normal(nc)
    t = np.arange(-th, th + 1)
    h = (np.sinc(t / 16) * np.kaiser(2 * th + 1, 12.0)).astype(np.float32).astype(np.float64)     # any symmetric taps do
    tab = np.zeros((16, 16))
    for tt, ht in zip(t, h):
        pp = tt % 16
        tab[pp, (tt - pp) // 16 + 6] = ht
    want = np.zeros(nc // 16, dtype=complex)
    for tt, ht in zip(t, h):
        want += ht * Q[(16 * np.arange(nc // 16) + tt) % nc]
    got = _column_walk_float64(Q, n2, n1, tab)
    assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()

