"""GPU: single-look K1 (csrc/k1_single_look.hpp) -- every capture byte read once, no statistics pre-pass.

The column kernels transform (code - m0) s0 with a sampled estimate (m0, s0), add up the exact window sums on the way,
and the residual of the mean is removed where the lags come out.  Checked here:
  * the window statistics the path ends up with are BIT-IDENTICAL to the pre-pass's (same integers, same f64 formulas);
  * peak lag identical and corr within 2e-6 of the pre-pass path (TDOA_DEBUG_NO_K1_ONCE) on every kind of input, the whole
    lag array included, in the decimated inverse, the full inverse and the two-sweep plan;
  * against the oracle (ob_*, f64 time domain) inside the usual 1e-5;
  * the cases that must NOT take it (unequal lengths, short search ranges, TDOA_LAGS_GO) still run the pre-pass.
The float64 atan2 anchors (tests/test_gpu_anchors.py, test_gpu_configs.py) run through this path by default."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ML = 20000


def _pairs(oracle, n):
    out = [("delayed_fm", oracle.simulate_delayed_fm(n, 0, 4242, 1), oracle.simulate_delayed_fm(n, 37, 4242, 2))]
    sim = [oracle.simulate_station(nm, n, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    weak = [oracle.simulate_weak_station(nm, n, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    out.append(("simulator.go ref 0-1", sim[0][:2 * n], sim[1][:2 * n]))
    out.append(("simulator.go tgt 0-2", sim[0][2 * n:4 * n], sim[2][2 * n:4 * n]))
    out.append(("weak tgt 1-2", weak[1][2 * n:4 * n], weak[2][2 * n:4 * n]))
    out.append(("weak ref 0-1 (constant bytes)", weak[0][:2 * n], weak[1][:2 * n]))
    return out


def _stats_tuple(st):
    return (st.s1, st.s2_lo, st.s2_hi, np.float32(st.mean).view(np.uint32), np.float32(st.scale).view(np.uint32))


@pytest.mark.parametrize("n", [1_100_000, 1_999_999])
def test_single_look_vs_prepass_and_exact_statistics(oracle, n, capsys):
    import tdoa_amd
    rows = []
    with tdoa_amd.Context(max_lag=ML, window_len=n) as c, tdoa_amd.Context(max_lag=ML, window_len=n) as ref:
        ref.debug_flags(no_k1_once=True)
        for name, a, b in _pairs(oracle, n):
            lag, corr = c.fm_xcorr(a, b, ML)
            st0, once = c.last_k1(0)
            st1, _ = c.last_k1(1)
            assert once, name
            rlag, rcorr = ref.fm_xcorr(a, b, ML)
            assert not ref.last_k1(0)[1]
            # the statistics of both windows: the pre-pass's integers and floats, bit for bit
            assert _stats_tuple(st0) == _stats_tuple(ref.fm_stats(a)), name
            assert _stats_tuple(st1) == _stats_tuple(ref.fm_stats(b)), name
            assert lag == rlag, (name, lag, rlag)
            if rcorr == 0.0:
                assert corr == 0.0
                rows.append((name, lag, corr, 0.0, 0.0))
                continue
            dev = abs(corr - rcorr) / abs(rcorr)
            la = c.fm_xcorr_lags(a, b, ML)
            lr = ref.fm_xcorr_lags(a, b, ML)
            arr = np.abs(la - lr).max() / np.abs(lr).max()
            rows.append((name, lag, corr, dev, arr))
            assert dev < 2e-6 and arr < 2e-6, (name, dev, arr)
    with capsys.disabled():
        print("\n  single-look K1 vs pre-pass path, L = %d" % n)
        for r in rows:
            print("    %-32s lag %6d corr %13.6f  |dcorr|/|corr| %.2e  lag array %.2e of the peak" % r)


def test_single_look_vs_oracle_all_lags(oracle):
    """all 2 max_lag - 1 lags against the f64 time-domain oracle on the same codes"""
    import tdoa_amd
    n, ml = 1_050_000, 6000
    a = oracle.simulate_delayed_fm(n, 0, 99, 1)
    b = oracle.simulate_delayed_fm(n, -1234, 99, 2)
    with tdoa_amd.Context(max_lag=ml, window_len=n) as c:
        lags = c.fm_xcorr_lags(a, b, ml)
        assert c.last_k1(0)[1]
        lag, corr = c.fm_xcorr(a, b, ml)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, olags = oracle.b_xcorr_peak_fft(ta, tb, ml)
    assert lag == olag == -1234 and abs(corr - ocorr) <= 1e-5 * abs(ocorr)
    assert np.abs(lags - olags).max() <= 1e-5 * np.abs(olags).max()


def test_single_look_full_inverse_and_fine(oracle):
    """the pruned column kernels and the refinement carry the same correction as k_small_col_peak"""
    import tdoa_amd
    n = 1_100_000
    a = oracle.simulate_delayed_fm(n, 0, 5, 1)
    b = oracle.simulate_delayed_fm(n, 19990, 5, 2)
    with tdoa_amd.Context(max_lag=ML, window_len=n) as c, tdoa_amd.Context(max_lag=ML, window_len=n) as ref:
        ref.debug_flags(no_k1_once=True)
        c.debug_flags(no_decimate=True)
        ref.debug_flags(no_decimate=True, no_k1_once=True)
        got, want = c.fm_xcorr(a, b, ML), ref.fm_xcorr(a, b, ML)
        assert c.last_k1(0)[1] and not ref.last_k1(0)[1]
        assert got[0] == want[0] == 19990 and abs(got[1] - want[1]) <= 2e-6 * abs(want[1])
        la, lr = c.fm_xcorr_lags(a, b, ML), ref.fm_xcorr_lags(a, b, ML)
        assert np.abs(la - lr).max() <= 2e-6 * np.abs(lr).max()
        for ctx in (c, ref):
            ctx.debug_flags(no_k1_once=ctx is ref)
        fa, fr = c.fm_xcorr_fine(a, b, ML, 25000.0), ref.fm_xcorr_fine(a, b, ML, 25000.0)
        assert c.last_k1(0)[1] and not ref.last_k1(0)[1]
        assert fa[0][0] == fr[0][0] == 19990 and abs(fa[0][1] - fr[0][1]) <= 2e-6 * abs(fr[0][1])
        assert abs(fa[1]["delay"] - fr[1]["delay"]) < 1e-4
        assert np.abs(np.array(fa[1]["y"]) - np.array(fr[1]["y"])).max() <= 2e-6 * abs(want[1])


def test_single_look_batched_process_ragged_alignment(oracle):
    """tdoa_process: windows that start on odd 2-byte boundaries and odd lengths, three stations, replayed graph"""
    import tdoa_amd
    blk, wl = 2_200_002, 1_100_001
    caps = [np.concatenate([oracle.simulate_delayed_fm(blk + (k == 1), 100 + d, 310 + k, 10 * s + k) for k in range(3)])[:2 * (3 * blk + s)]
            for s, d in enumerate((0, 41, -17))]
    with tdoa_amd.Context(max_lag=ML, window_len=wl) as c, tdoa_amd.Context(max_lag=ML, window_len=wl) as ref:
        ref.debug_flags(no_k1_once=True)
        got = c.process_u8(caps)
        assert c.last_k1(0)[1] and c.graph_info()["memsets"] == 0
        again = c.process()
        want = ref.process_u8(caps)
        assert not ref.last_k1(0)[1]
    assert np.array_equal(got, again)
    assert np.array_equal(got["lag"], want["lag"])
    assert np.abs(got["corr"] - want["corr"]).max() <= 2e-6 * np.abs(want["corr"]).max()
    assert (np.abs(got["corr"] - want["corr"]) <= 2e-6 * np.abs(want["corr"]) + 1e-9).all()


def test_single_look_is_not_taken_where_it_does_not_apply(oracle):
    import tdoa_amd
    n = 1_100_000
    a = oracle.simulate_delayed_fm(n, 0, 8, 1)
    b = oracle.simulate_delayed_fm(n - 1000, 300, 8, 2)
    with tdoa_amd.Context(max_lag=ML, window_len=n) as c:
        c.fm_xcorr(a, b, ML)                      # unequal lengths
        assert not c.last_k1(0)[1]
        c.fm_xcorr(a, a, 2000)                    # short search range: short-lag / segment forms
        assert not c.last_k1(0)[1]
        c.fm_xcorr(a, a, ML)
        assert c.last_k1(0)[1]
    with tdoa_amd.Context(max_lag=ML, window_len=n, lag_mode=tdoa_amd.capi.LAGS_GO) as c:
        c.fm_xcorr(a, a, ML)                      # the Go lag set cuts the template: statistics over other windows
        assert not c.last_k1(0)[1]


def test_single_look_worst_case_codes(oracle):
    """a tone at exactly fs/2 (every sample the reverse of the one before): every code is +2^23, the largest the
    accumulators can meet; zero variance -> (0, 0.0) on both paths and exact sums"""
    import tdoa_amd
    n = 1_100_000
    iq = np.empty(2 * n, np.uint8)
    iq[0::4], iq[1::4], iq[2::4], iq[3::4] = 255, 255, 0, 0
    with tdoa_amd.Context(max_lag=ML, window_len=n) as c:
        assert c.fm_xcorr(iq, iq, ML) == (0, 0.0)
        st, once = c.last_k1(0)
        assert once and st.s1 == n * (1 << 23) and (st.s2_hi << 64 | st.s2_lo) == n * (1 << 46)
        assert _stats_tuple(st) == _stats_tuple(c.fm_stats(iq))


def test_single_look_with_a_biased_estimate():
    """the estimate (m0, s0) comes from 256 evenly spaced runs of 16 samples; a modulation whose period IS that spacing
    shows every run the same half of a square wave, so m0 misses the mean by about one sigma -- the largest residual the
    correction can meet.  The result must not care: the removed terms are exact whatever m0 is (only the float32 rounding of
    the transforms sees the pedestal).  Measured: 1.3e-6 of the peak over the lag array, against 5e-7 with an unbiased m0."""
    import tdoa_amd
    n = 1_200_000
    stride = (n - 1 - 16) // 255                                           # k_once_estimate: start = 1 + t (len - 17) / 255
    rng = np.random.default_rng(1)

    def make(delay, seed):
        t = np.arange(n + 200)
        sq = ((t % stride) < 0.5 * stride).astype(np.float64) * 2 - 1
        f = 0.3 + 0.2 * sq + 0.02 * np.cumsum(rng.standard_normal(n + 200)) / np.sqrt(np.arange(1, n + 201))
        phi = np.cumsum(f)[200 - delay:200 - delay + n]
        r = np.random.default_rng(seed)
        iq = np.stack([0.45 * np.cos(phi), 0.45 * np.sin(phi)], 1).reshape(-1) + (r.random(2 * n) * 2 - 1) * 0.02
        return np.clip(np.trunc(iq * 127.5 + 127.5), 0, 255).astype(np.uint8)

    a, b = make(0, 11), make(37, 12)
    with tdoa_amd.Context(max_lag=ML, window_len=n) as c, tdoa_amd.Context(max_lag=ML, window_len=n) as ref:
        ref.debug_flags(no_k1_once=True)
        got, want = c.fm_xcorr(a, b, ML), ref.fm_xcorr(a, b, ML)
        st, once = c.last_k1(0)
        assert once and _stats_tuple(st) == _stats_tuple(ref.fm_stats(a))
        la, lr = c.fm_xcorr_lags(a, b, ML), ref.fm_xcorr_lags(a, b, ML)
    assert got[0] == want[0] == 37 and abs(got[1] - want[1]) <= 2e-6 * abs(want[1])
    assert np.abs(la - lr).max() <= 4e-6 * np.abs(lr).max()


def test_single_look_seeded_sweep(oracle):
    """window lengths, search ranges (decimated inverse, full inverse where the filter does not fit, odd and even lengths)
    and byte distributions drawn from a seed: peak identical and the whole lag array within 3e-6 of the pre-pass form's"""
    import tdoa_amd
    rng = np.random.default_rng(20260404)
    for case in range(8):
        n = int(rng.integers(1_050_000, 1_500_000))
        ml = int(rng.choice([4096, 5000, 12345, 20000, 26000, 32000]))
        delay = int(rng.integers(-ml + 2, ml - 1))
        kind = case % 4
        if kind == 0:
            a, b = oracle.simulate_delayed_fm(n, max(0, -delay), 100 + case, 1), oracle.simulate_delayed_fm(n, max(0, delay), 100 + case, 2)
        elif kind == 1:
            st = [oracle.simulate_station(nm, n, oracle.SEED_BASE + 7 * case + i) for i, nm in enumerate(oracle.COLLECTORS[:2])]
            a, b = st[0][:2 * n], st[1][:2 * n]
        elif kind == 2:
            a, b = (rng.integers(0, 256, size=2 * n, dtype=np.uint8) for _ in range(2))           # uniform random bytes
        else:
            a = oracle.simulate_delayed_fm(n, 0, 200 + case, 1)
            b = np.roll(a.reshape(-1, 2), max(0, delay) % 1000, axis=0).reshape(-1).copy()      # a strong, exactly shifted copy
        with tdoa_amd.Context(max_lag=ml, window_len=n) as c, tdoa_amd.Context(max_lag=ml, window_len=n) as ref:
            ref.debug_flags(no_k1_once=True)
            got, want = c.fm_xcorr(a, b, ml), ref.fm_xcorr(a, b, ml)
            assert c.last_k1(0)[1] and not ref.last_k1(0)[1], (case, n, ml)
            la, lr = c.fm_xcorr_lags(a, b, ml), ref.fm_xcorr_lags(a, b, ml)
        assert got[0] == want[0], (case, n, ml, got, want)
        assert abs(got[1] - want[1]) <= 3e-6 * abs(want[1]), (case, n, ml, got, want)
        assert np.abs(la - lr).max() <= 3e-6 * np.abs(lr).max(), (case, n, ml)


def _offset_carrier(n, omega, seed):
    """a strong, almost unmodulated carrier far off the centre frequency: phase advance `omega` rad per sample plus a small
    random walk, 100 LSB of amplitude, +-1 LSB of noise -- discriminator output = omega + a few hundredths of a radian, i.e.
    |mean| / sigma in the hundreds"""
    rng = np.random.default_rng(seed)
    phi = omega * np.arange(n, dtype=np.float64) + np.cumsum(rng.standard_normal(n) * 0.003)
    x = 100.0 * np.exp(1j * phi) + (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))
    iq = np.empty(2 * n, dtype=np.uint8)
    iq[0::2] = np.clip(np.trunc(x.real + 127.5), 0, 255).astype(np.uint8)
    iq[1::2] = np.clip(np.trunc(x.imag + 127.5), 0, 255).astype(np.uint8)
    return iq


@pytest.mark.parametrize("omega", [2.5, -1.9])
def test_single_look_with_a_large_mean_over_sigma(oracle, omega, capsys):
    """the correction used W = L eps with eps built from the float32-ROUNDED mean; the true window sum differs by
    L s0 (f32(mean) - mean), first order in the rounding, and |mean| >> sigma (a carrier with a large frequency offset and
    little modulation) puts that into noise-level lag values: ~1e-4 of a noise-level peak.  With the exact window sum carried
    in OnceFin the single-look path agrees with the pre-pass path like everywhere else, and both with the f64 oracle."""
    import tdoa_amd
    n = 1_100_000
    # independent, and 0.0123 rad/sample apart (at one frequency the quantisation pattern of the rotating phasor is common to
    # both and correlates): every lag value is noise-level
    a, b = _offset_carrier(n, omega, 71), _offset_carrier(n, omega + 0.0123, 72)
    with tdoa_amd.Context(max_lag=ML, window_len=n) as c, tdoa_amd.Context(max_lag=ML, window_len=n) as ref:
        ref.debug_flags(no_k1_once=True)
        la, lr = c.fm_xcorr_lags(a, b, ML), ref.fm_xcorr_lags(a, b, ML)
        st, once = c.last_k1(0)
        assert once and not ref.last_k1(0)[1]
        got, want = c.fm_xcorr(a, b, ML), ref.fm_xcorr(a, b, ML)
    sigma = 1.0 / float(st.scale)
    assert abs(float(st.mean)) > 100.0 * sigma                                   # the regime the finding is about
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, olags = oracle.b_xcorr_peak_fft(ta, tb, ML)
    peak = np.abs(olags).max()
    assert peak < 12.0                                                           # noise-level (full correlation: 1049)
    dev_pre = np.abs(la - lr).max() / peak
    dev_orc = np.abs(la - olags).max() / peak
    with capsys.disabled():
        print("\n  offset carrier %.1f rad/sample: mean %.0f codes, sigma %.0f (|mean|/sigma %.0f); lag array vs pre-pass %.2e, "
              "vs f64 oracle %.2e of the peak (%.2f)" % (omega, float(st.mean), sigma, abs(float(st.mean)) / sigma, dev_pre, dev_orc, peak))
    assert got[0] == want[0] == olag
    assert dev_pre < 2e-6 and dev_orc < 1e-5
    assert abs(got[1] - ocorr) <= 1e-5 * abs(ocorr)
