"""GPU parity tests for the north-star (mode B) path, through the C ABI:
u8 IQ -> K1 discriminator -> K2 Stockham FFT -> K3 conj-multiply -> K4 inverse -> K5 argmax.
Oracle: oracle/tdoa_oracle.c (ob_* functions, f64 time domain)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5      # north_star: correlation magnitudes within 1e-5 relative


@pytest.fixture(scope="module")
def ctx():
    import tdoa_amd
    c = tdoa_amd.Context(max_lag=500, window_len=10000)
    yield c
    c.close()


def _assert_lags_close(got, want, tol=REL_TOL):
    peak = np.abs(want).max()
    assert peak > 0
    err = np.abs(got - want).max() / peak
    assert err < tol, "max lag error %.3g of peak" % err


def test_k1_discriminator_bit_exact(ctx, oracle):
    raw = oracle.simulate_station("kx0u", 20000, oracle.SEED_BASE)
    for lo, n in ((0, 20000), (40000, 12345), (80002, 777)):   # byte offsets: even sample starts
        iq = raw[lo:lo + 2 * n]
        got, st = ctx.fm_preprocess(iq)
        want, ost = oracle.b_preprocess(iq)
        assert (st.s1, st.s2_lo, st.s2_hi) == (ost.s1, ost.s2_lo, ost.s2_hi)
        assert np.float32(st.mean).tobytes() == np.float32(ost.mean).tobytes()
        assert np.float32(st.scale).tobytes() == np.float32(ost.scale).tobytes()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        only = ctx.fm_stats(iq)                                 # the reduce-only pass of the fused path
        assert (only.s1, only.s2_lo, only.s2_hi) == (ost.s1, ost.s2_lo, ost.s2_hi)
        assert np.float32(only.mean).tobytes() == np.float32(ost.mean).tobytes()
        assert np.float32(only.scale).tobytes() == np.float32(ost.scale).tobytes()


def test_k1_statistics_need_more_than_64_bits(oracle):
    """sum of code^2 over a 10 s window of a tone near the band edge exceeds 2^64: both passes carry it in two words"""
    import tdoa_amd
    n = 4_000_000
    t = np.arange(n)
    iq = np.empty(2 * n, dtype=np.uint8)
    iq[0::2] = np.clip(np.trunc(100.0 * np.cos(2 * np.pi * 0.47 * t) + 127.5), 0, 255).astype(np.uint8)
    iq[1::2] = np.clip(np.trunc(100.0 * np.sin(2 * np.pi * 0.47 * t) + 127.5), 0, 255).astype(np.uint8)
    want, ost = oracle.b_preprocess(iq)
    assert ost.s2_hi > 0
    with tdoa_amd.Context(max_lag=500, window_len=n) as c:
        got, st = c.fm_preprocess(iq)
        only = c.fm_stats(iq)
    for s in (st, only):
        assert (s.s1, s.s2_lo, s.s2_hi) == (ost.s1, ost.s2_lo, ost.s2_hi)
        assert np.float32(s.scale).tobytes() == np.float32(ost.scale).tobytes()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_k1_bit_exact_on_random_bytes(ctx, oracle):
    rng = np.random.default_rng(11)
    iq = rng.integers(0, 256, size=2 * 200000, dtype=np.uint8)
    iq[:8] = [0, 0, 255, 255, 0, 255, 255, 0]                  # extremes
    got, st = ctx.fm_preprocess(iq)
    want, ost = oracle.b_preprocess(iq)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (st.s1, st.s2_lo, st.s2_hi) == (ost.s1, ost.s2_lo, ost.s2_hi)


@pytest.mark.parametrize("n1,n2,max_lag", [
    (40, 40, 8),            # N = 64: smallest plan (N2 = 2)
    (1000, 900, 100),       # N = 2048: N2 = 16
    (3001, 3001, 257),      # odd lengths
    (30000, 30000, 2000),   # N = 32768
    (70000, 66000, 600),    # N = 131072: 4096-point rows
])
def test_xcorr_all_lags_vs_oracle(ctx, oracle, n1, n2, max_lag):
    a = oracle.simulate_delayed_fm(n1, 0, 1001, 7)
    b = oracle.simulate_delayed_fm(n2, 5, 1001, 8)
    got = ctx.fm_xcorr_lags(a, b, max_lag)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    want = oracle.b_xcorr_all_lags(ta, tb, max_lag)
    _assert_lags_close(got, want)
    lag, corr = ctx.fm_xcorr(a, b, max_lag)
    olag, ocorr = oracle.b_pick_peak(want, max_lag)
    assert lag == olag
    assert abs(corr - ocorr) <= REL_TOL * abs(ocorr)


@pytest.mark.parametrize("delay", [0, 1, 37, -37, 113, -114, 499])
def test_peak_finds_true_delay(ctx, oracle, delay):
    n = 20000
    # station b delayed by `delay` samples relative to a (negative: a delayed)
    a = oracle.simulate_delayed_fm(n, max(0, -delay), 4242, 1)
    b = oracle.simulate_delayed_fm(n, max(0, delay), 4242, 2)
    lag, corr = ctx.fm_xcorr(a, b, 500)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr = oracle.b_xcorr_peak(ta, tb, 500)
    assert lag == delay == olag
    assert abs(corr - ocorr) <= REL_TOL * abs(ocorr)


def test_empty_and_tiny_inputs(ctx, oracle):
    a = oracle.simulate_delayed_fm(100, 0, 1, 1)
    assert ctx.fm_xcorr(np.zeros(0, np.uint8), a, 10) == (0, 0.0)     # processor.go:622-625
    flat = np.full(2 * 64, 128, np.uint8)                              # constant capture: zero phase
    lag, corr = ctx.fm_xcorr(flat, flat, 10)
    assert (lag, corr) == (0, 0.0)


def test_process_batched_matches_oracle(oracle):
    import tdoa_amd
    block, wl, ml = 30000, 10000, 300
    caps = [oracle.simulate_station(nm, block, oracle.SEED_BASE + i, tx_power=200000.0)
            for i, nm in enumerate(oracle.COLLECTORS)]
    with tdoa_amd.Context(max_lag=ml, window_len=wl, windows_per_batch=2) as c:
        peaks = c.process_u8(caps)
        assert peaks.shape == (9, 3)
        pairs = [(0, 1), (0, 2), (1, 2)]                               # processor.go:816-817 order
        for wid in range(9):
            off = (wid // 3) * block + (wid % 3) * wl
            pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0] for cp in caps]
            for p, (i, j) in enumerate(pairs):
                olag, ocorr = oracle.b_xcorr_peak(pre[i], pre[j], ml)
                assert peaks[wid, p]["lag"] == olag, (wid, p)
                assert abs(peaks[wid, p]["corr"] - ocorr) <= REL_TOL * abs(ocorr)
        # sharded run (rank r of 2) reproduces exactly the windows it owns
        for r in range(2):
            part = c.process(rank=r, world=2)
            for wid in range(9):
                if wid % 2 == r:
                    assert np.array_equal(part[wid], peaks[wid])
                else:
                    assert not part[wid]["lag"].any() and not part[wid]["corr"].any()


def test_full_size_window_vs_f64_fft(oracle):
    """BASELINE config 2 geometry: L = 2 000 000, max_lag 20000, N = 2^21."""
    import tdoa_amd
    n = 2_000_000
    a = oracle.simulate_delayed_fm(n, 0, 99, 3)
    b = oracle.simulate_delayed_fm(n, 57, 99, 4)
    with tdoa_amd.Context() as c:
        lag, corr = c.fm_xcorr(a, b, 20000)
        lags = c.fm_xcorr_lags(a, b, 20000)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, 20000)
    assert lag == olag == 57
    assert abs(corr - ocorr) <= REL_TOL * abs(ocorr)
    _assert_lags_close(lags, want)


def test_hot_size_and_fallback_kernels_agree(oracle):
    """N = 2^21: radix-16 register kernels + pruned inverse vs the any-size LDS kernels."""
    import tdoa_amd
    n = 2_000_000
    a = oracle.simulate_delayed_fm(n, 3, 123, 5)
    b = oracle.simulate_delayed_fm(n, 0, 123, 6)          # a is the delayed one: lag -3
    with tdoa_amd.Context() as c:
        hot = c.fm_xcorr_lags(a, b, 20000)
        lag_h, corr_h = c.fm_xcorr(a, b, 20000)
        c.force_generic(True)
        gen = c.fm_xcorr_lags(a, b, 20000)
        lag_g, corr_g = c.fm_xcorr(a, b, 20000)
    _assert_lags_close(hot, gen)
    assert lag_h == lag_g == -3
    assert abs(corr_h - corr_g) <= REL_TOL * abs(corr_g)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, 20000)
    assert olag == -3
    _assert_lags_close(hot, want)
    print("max |hot - f64| / peak = %.3g" % (np.abs(hot - want).max() / np.abs(want).max()))


def test_odd_and_unaligned_windows_use_fallback(oracle):
    """window starts that are not 4-byte aligned / odd lengths go through the generic kernels"""
    import tdoa_amd
    blk, wl, ml = 30001, 9999, 200                           # odd block -> windows at odd sample offsets
    caps = [oracle.simulate_delayed_fm(3 * blk, d, 55, 20 + i) for i, d in enumerate((0, 9))]
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        peaks = c.process_u8(caps)
    wpb = blk // wl
    assert peaks.shape == (3 * wpb, 1)
    for wid in range(3 * wpb):
        off = (wid // wpb) * blk + (wid % wpb) * wl
        pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0] for cp in caps]
        olag, ocorr = oracle.b_xcorr_peak(pre[0], pre[1], ml)
        assert peaks[wid, 0]["lag"] == olag == 9
        assert abs(peaks[wid, 0]["corr"] - ocorr) <= REL_TOL * abs(ocorr)


@pytest.mark.parametrize("n,delay,label", [
    (4_000_000, -41, "cfg5 window: 1 s at 4 Msps, N = 2^22 (4096 x 512, radix-2 last column stage)"),
    (8_000_000, 123, "2 s at 4 Msps, N = 2^23 (4096 x 1024, radix-4 last column stage)"),
    (10_000_000, -7, "5 s at 2 Msps, N = 2^24 (4096 x 2048, two-sweep column pass, 8-point finish)"),
    (20_000_000, 88, "cfg3 window: 10 s at 2 Msps, N = 5 x 2^22 (4096 x 2560, two-sweep column pass, 10-point finish)"),
    (21_000_000, -19, "10.5 s at 2 Msps: beyond 5 x 2^22, N = 3 x 2^23 (4096 x 3072, two-sweep column pass, 12-point finish)"),
    (25_200_000, 5, "12.6 s at 2 Msps: beyond 3 x 2^23, N = 2^25 (4096 x 4096, two-sweep column pass, 16-point finish)"),
])
def test_long_windows_vs_f64_fft(oracle, n, delay, label):
    """BASELINE configs 3 and 5 window geometries (and the size between them)."""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n, max(0, -delay), 31, 1)
    b = oracle.simulate_delayed_fm(n, max(0, delay), 31, 2)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, 20000)
    with tdoa_amd.Context() as c:
        lag, corr = c.fm_xcorr(a, b, 20000)
        assert lag == olag == delay, label
        assert abs(corr - ocorr) <= REL_TOL * abs(ocorr), label
        if n <= 10_000_000:
            hot = c.fm_xcorr_lags(a, b, 20000)
            _assert_lags_close(hot, want)
            c.force_generic(True)
            _assert_lags_close(c.fm_xcorr_lags(a, b, 20000), hot)


def test_two_sweep_plan_padded_rows_in_every_form(oracle):
    """N = 2^24 (4096 x 2048): TZ rows of a two-sweep plan are padded after every 256 rows (FftPlan.zpad); every kernel
    family that reads them -- general inverse, short-lag rows (3000 lags), the same from materialised codes -- must agree
    with the f64 oracle and with each other"""
    import tdoa_amd
    n, delay = 9_000_000, 1234
    a = oracle.simulate_delayed_fm(n, 0, 77, 1)
    b = oracle.simulate_delayed_fm(n, delay, 77, 2)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, 3000)
    assert olag == delay
    with tdoa_amd.Context() as c:
        lags = c.fm_xcorr_lags(a, b, 3000)                   # short-lag form (k_inv_row_pair4096<*, 8>)
        assert tuple(c.plan_info())[1:] == (4096, 2048)
        _assert_lags_close(lags, want)
        c.debug_flags(no_short_lag=True)                      # general inverse + pruned column pass
        _assert_lags_close(c.fm_xcorr_lags(a, b, 3000), lags)
        c.debug_flags(no_fused_k1=True)                       # k_fwd_col256_c16<true> writes the padded rows
        _assert_lags_close(c.fm_xcorr_lags(a, b, 3000), lags)


@pytest.mark.parametrize("n1,n2,max_lag,delay", [(2_000_000, 2_000_000, 20000, 57), (1_234_567, 1_999_999, 20000, -19876),
                                                 (4_000_000, 3_999_000, 20000, 1234), (2_200_000, 3_100_000, 26000, -25001)])
def test_decimated_pair_step_column_walk_vs_tiles(oracle, n1, n2, max_lag, delay):
    """the two forms of the decimated pair step on the 4096 x 256 and 4096 x 512 plans: k_pair_decimate_cols (a thread walks
    a spectrum column; the library's choice for batches with more pairs than stations) against k_pair_decimate16 (4096-bin
    tiles in LDS): same filter, same outputs -- lag arrays equal to rounding, and both against the full inverse"""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n1, max(0, -delay), 71, 1)
    b = oracle.simulate_delayed_fm(n2, max(0, delay), 71, 2)
    with tdoa_amd.Context(max_lag=max_lag, window_len=max(n1, n2)) as c:
        tiles, peak_t = c.fm_xcorr_lags(a, b, max_lag), c.fm_xcorr(a, b, max_lag)
        c.debug_flags(dec_cols_always=True)
        cols, peak_c = c.fm_xcorr_lags(a, b, max_lag), c.fm_xcorr(a, b, max_lag)
        c.debug_flags(no_decimate=True)
        full = c.fm_xcorr_lags(a, b, max_lag)
    assert peak_t[0] == peak_c[0] == delay
    scale = np.abs(full).max()
    assert np.abs(cols - tiles).max() <= 5e-7 * scale
    assert np.abs(cols - full).max() <= 2e-6 * scale


@pytest.mark.parametrize("n1,n2,delay,max_lag", [(20_000_000, 20_000_000, 88, 20000), (20_000_000, 19_876_543, -19999, 20000),
                                                 (18_000_001, 20_000_000, 7, 12000)])
def test_decimated_inverse_on_the_ten_second_plan(oracle, n1, n2, delay, max_lag):
    """Ten-second windows.  Default (round 5): N = 5 x 2^22 (4096 x 2560) -- ten 256-point sub-transforms and a 10-point finish
    down the columns, the decimating FIR as a column stencil (k_pair_decimate_cols<2560>, dec_stream.hpp) over the unpacked
    spectra the row pass leaves in place, a 4096 x 160 small plan.  TDOA_DEBUG_POW2_ONLY: the same in N = 2^25 (4096 x 4096,
    k_pair_decimate_cols<4096>).  Both against the full inverse (TDOA_DEBUG_NO_DECIMATE: a power-of-two plan by itself), lag by
    lag, with and without the single-look K1."""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n1, max(0, -delay), 61, 1)
    b = oracle.simulate_delayed_fm(n2, max(0, delay), 61, 2)
    with tdoa_amd.Context(max_lag=max_lag, window_len=max(n1, n2)) as c:
        dec_lags, dec_peak = c.fm_xcorr_lags(a, b, max_lag), c.fm_xcorr(a, b, max_lag)
        assert tuple(c.plan_info()) == (5 << 22, 4096, 2560)
        c.debug_flags(no_k1_once=True)
        pre_lags = c.fm_xcorr_lags(a, b, max_lag)
        assert tuple(c.plan_info()) == (5 << 22, 4096, 2560)
        c.debug_flags(pow2_only=True)
        p2_lags, p2_peak = c.fm_xcorr_lags(a, b, max_lag), c.fm_xcorr(a, b, max_lag)
        assert tuple(c.plan_info()) == (1 << 25, 4096, 4096)
        c.debug_flags(no_decimate=True)
        full_lags, full_peak = c.fm_xcorr_lags(a, b, max_lag), c.fm_xcorr(a, b, max_lag)
        assert tuple(c.plan_info()) == (1 << 25, 4096, 4096)
    assert dec_peak[0] == p2_peak[0] == full_peak[0] == delay
    assert abs(dec_peak[1] - full_peak[1]) <= 2e-6 * abs(full_peak[1])
    assert abs(p2_peak[1] - full_peak[1]) <= 2e-6 * abs(full_peak[1])
    scale = np.abs(full_lags).max()
    assert np.abs(dec_lags - full_lags).max() <= 2e-6 * scale
    assert np.abs(pre_lags - full_lags).max() <= 2e-6 * scale
    assert np.abs(p2_lags - full_lags).max() <= 2e-6 * scale


@pytest.mark.parametrize("n1,n2,delay", [(9_000_000, 9_000_000, 4321), (8_400_001, 10_000_000, -19999)])
def test_decimated_inverse_on_the_4096x2048_plan(oracle, n1, n2, delay):
    """windows of 4 to 8 s at 2 Msps (N = 2^24, 4096 x 2048: two-sweep column pass with the 8-point finish) ran the full inverse
    until round 5; now the decimating FIR walks their 2048-row columns too (k_pair_decimate_cols<2048> for a pair call,
    k_pair_decimate_staged<2048> in batches) with a 4096 x 128 small plan.  Against the full inverse of the same context, lag
    by lag, and the peak against the f64 oracle."""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n1, max(0, -delay), 63, 1)
    b = oracle.simulate_delayed_fm(n2, max(0, delay), 63, 2)
    with tdoa_amd.Context(max_lag=20000, window_len=max(n1, n2)) as c:
        dec_lags, dec_peak = c.fm_xcorr_lags(a, b, 20000), c.fm_xcorr(a, b, 20000)
        assert tuple(c.plan_info()) == (1 << 24, 4096, 2048) and c.last_k1(0)[1] == (n1 == n2)
        c.debug_flags(no_decimate=True)
        full_lags, full_peak = c.fm_xcorr_lags(a, b, 20000), c.fm_xcorr(a, b, 20000)
    assert dec_peak[0] == full_peak[0] == delay
    assert not np.array_equal(dec_lags, full_lags)                       # (two different inverses ran)
    scale = np.abs(full_lags).max()
    assert np.abs(dec_lags - full_lags).max() <= 2e-6 * scale
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, 20000)
    assert olag == delay and abs(dec_peak[1] - ocorr) <= REL_TOL * abs(ocorr)
    _assert_lags_close(dec_lags, want)


def test_profiling_inside_the_replayed_graph(oracle):
    """tdoa_profile_enable(2): the step keeps replaying as one hipGraph and the selected scope is timed by event-record
    nodes spliced into the captured graph; the results must not change and only the selected scope may report launches"""
    import tdoa_amd
    blk, wl, ml = 30000, 10000, 300
    caps = [oracle.simulate_delayed_fm(3 * blk, d, 5, 40 + i) for i, d in enumerate((0, 21, 8))]
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        for s, cap in enumerate(caps):
            c.capture_upload(s, cap)
        plain = c.process()
        base_nodes = c.graph_info()["nodes"]
        c.profile_enable(2)
        c.profile_select(["k_fwd_row"])
        first = c.process()                                  # captures and instruments the step
        c.profile_reset()
        again = [c.process() for _ in range(3)]
        prof = c.profile()
        c.profile_enable(False)
        c.profile_select(None)
        after = c.process()                                  # back to the plain graph
        assert c.graph_info()["nodes"] == base_nodes
    for got in [first, after] + again:
        assert np.array_equal(got, plain)
    assert prof["k_fwd_row"]["launches"] == 3 and prof["k_fwd_row"]["ms"] > 0.0
    assert all(v["launches"] == 0 for k, v in prof.items() if k != "k_fwd_row")


def test_eight_stations_28_pairs(oracle):
    """BASELINE config 4 geometry in miniature: 8 collectors, 28 pairs ordered i<j."""
    import tdoa_amd
    blk, wl, ml = 12000, 6000, 120
    delays = [0, 7, 19, 33, 2, 51, 40, 11]
    caps = [oracle.simulate_delayed_fm(3 * blk, d, 808, 100 + i) for i, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        peaks = c.process_u8(caps)
    assert peaks.shape == (6, 28)
    pairs = [(i, j) for i in range(8) for j in range(i + 1, 8)]
    for wid in (0, 3, 5):
        off = (wid // 2) * blk + (wid % 2) * wl
        pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0] for cp in caps]
        for p, (i, j) in enumerate(pairs):
            olag, ocorr = oracle.b_xcorr_peak(pre[i], pre[j], ml)
            assert peaks[wid, p]["lag"] == olag == delays[j] - delays[i], (wid, i, j)
            assert abs(peaks[wid, p]["corr"] - ocorr) <= REL_TOL * abs(ocorr)


def test_capture_upload_file_roundtrip(tmp_path, oracle):
    import tdoa_amd
    rng = np.random.default_rng(21)
    raw = rng.integers(0, 256, size=2 * (40 * 1024 * 1024 // 2) + 6, dtype=np.uint8)    # several 8 MiB staging chunks + a ragged tail
    path = tmp_path / "kx0u-1754900000.dat"
    raw.tofile(path)
    with tdoa_amd.Context() as c:
        n = c.capture_upload_file(0, str(path))
        assert n == raw.size // 2
        for first, cnt in ((0, 1000), (n - 777, 777), (16 * 1024 * 1024 - 5, 10)):
            assert np.array_equal(c.capture_download(0, first, cnt), raw[2 * first:2 * (first + cnt)])
        with pytest.raises(tdoa_amd.TdoaError):
            c.capture_upload_file(1, str(tmp_path / "missing.dat"))
        # whole capture, every 8 MiB staging chunk of every copy thread
        assert np.array_equal(c.capture_download(0, 0, n), raw[:2 * n])
        # host-memory path through the same uploader; then a shorter capture into the same (reused) buffer
        c.capture_upload(1, raw)
        assert np.array_equal(c.capture_download(1, 0, n), raw[:2 * n])
        short = raw[5_000_000:5_000_000 + 2 * 3_000_001]
        c.capture_upload(1, short)
        assert np.array_equal(c.capture_download(1, 0, 3_000_001), short)
        with pytest.raises(tdoa_amd.TdoaError):
            c.capture_download(1, 3_000_001, 1)               # beyond the new, shorter capture
        c.capture_upload(1, raw[:2 * 100])                    # small copies take the plain path
        assert np.array_equal(c.capture_download(1, 0, 100), raw[:200])


def test_attach_device_buffer_from_torch(oracle):
    """captures that already live in HBM (e.g. a torch tensor) are used in place, and the peaks can be
    written into a device buffer for the RCCL all-gather."""
    torch = pytest.importorskip("torch")
    import tdoa_amd
    from tdoa_amd.capi import PEAK_DTYPE
    blk, wl, ml = 9000, 3000, 100
    caps = [oracle.simulate_delayed_fm(3 * blk, d, 99, 40 + i) for i, d in enumerate((0, 21, 5))]
    dev = [torch.from_numpy(c.copy()).cuda() for c in caps]
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        for s, t in enumerate(dev):
            c.capture_attach_device(s, t.data_ptr(), t.numel() // 2)
        wpb, W = c.num_windows()
        out_dev = torch.zeros(W * 3 * 16, dtype=torch.uint8, device="cuda")
        host = c.process(out_dev_ptr=out_dev.data_ptr())
        torch.cuda.synchronize()
        got = np.frombuffer(out_dev.cpu().numpy().tobytes(), dtype=PEAK_DTYPE).reshape(W, 3)
        assert np.array_equal(got, host)
        assert (host["lag"] == np.array([21, 5, -16])).all()
        ref = c.process_u8(caps)                       # same bytes through the upload path
        assert np.array_equal(ref, host)


@pytest.mark.parametrize("max_lag,delay", [(128, 57), (128, -127), (511, 300), (512, -511), (1000, 999), (2047, -2046),
                                           (2048, 1500), (4095, 4094), (4096, -4000)])
def test_short_lag_form_vs_oracle_and_general_form(oracle, max_lag, delay):
    """|lag| ranges below 1024 take the segment form (overlap-save in LDS, no four-step transform at all), ranges below
    4095 the short-lag inverse (no V round trip) on 4096-point rows; 4096 is the first range that takes the general
    pruned form again.  All against the f64 oracle, and against each other."""
    import tdoa_amd
    n = 300_000                                                   # N = 2^19: 4096 x 64
    a = oracle.simulate_delayed_fm(n, max(0, -delay), 77, 1)
    b = oracle.simulate_delayed_fm(n, max(0, delay), 77, 2)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, max_lag)
    assert olag == delay
    with tdoa_amd.Context(max_lag=max_lag, window_len=n) as c:
        lags = c.fm_xcorr_lags(a, b, max_lag)
        _assert_lags_close(lags, want)
        (lag, corr), fine = c.fm_xcorr_fine(a, b, max_lag, 1e9)
        assert lag == olag and abs(corr - ocorr) <= REL_TOL * abs(ocorr)
        ofine = oracle.b_refine_peak(ta, tb, lag, 1e9)
        assert np.abs(fine["y"] - ofine["y"]).max() <= REL_TOL * abs(ocorr)
        assert abs(fine["frac"] - ofine["frac"]) < 1e-4
        c.debug_flags(no_segment_form=True)                        # short-lag shares of the four-step form
        shares = c.fm_xcorr_lags(a, b, max_lag)
        _assert_lags_close(shares, want)
        _assert_lags_close(shares, lags)
        (lag_s, corr_s), fine_s = c.fm_xcorr_fine(a, b, max_lag, 1e9)
        assert lag_s == olag and abs(corr_s - ocorr) <= REL_TOL * abs(ocorr) and abs(fine_s["frac"] - ofine["frac"]) < 1e-4
        c.debug_flags(no_short_lag=True)
        general = c.fm_xcorr_lags(a, b, max_lag)
        _assert_lags_close(general, lags)
        assert c.fm_xcorr(a, b, max_lag)[0] == lag


@pytest.mark.parametrize("n,label", [(70_000, "N = 2^17: 4096 x 16"), (200_000, "N = 2^18: 4096 x 32"),
                                     (400_000, "N = 2^19: 4096 x 64"), (666_666, "cfg1 block, N = 2^20: 4096 x 128")])
def test_short_column_kernels(oracle, n, label):
    """hot forward column pass for N2 = 16 .. 128 against the f64 oracle and the any-size kernels"""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n, 11, 55, 1)
    b = oracle.simulate_delayed_fm(n - 1234, 0, 55, 2)             # unequal lengths; a is the delayed one
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, 20000)
    assert olag == -11, label
    with tdoa_amd.Context(window_len=n) as c:
        hot = c.fm_xcorr_lags(a, b, 20000)
        _assert_lags_close(hot, want)
        assert c.fm_xcorr(a, b, 20000)[0] == olag
        c.force_generic(True)
        _assert_lags_close(c.fm_xcorr_lags(a, b, 20000), hot)


@pytest.mark.parametrize("per_batch", [0, 2])
def test_short_lag_form_in_the_batched_path(oracle, per_batch):
    """tdoa_process with 4096-point rows and a 300-lag range: many pair-windows through the short-lag inverse,
    against the oracle and against the general form."""
    import tdoa_amd
    block, wl, ml = 140_000, 70_000, 300
    delays = [0, 41, -17]
    caps = [np.concatenate([oracle.simulate_delayed_fm(block, 100 + d, 640 + k, 10 * s + k) for k in range(3)])
            for s, d in enumerate(delays)]
    pairs = [(0, 1), (0, 2), (1, 2)]
    with tdoa_amd.Context(max_lag=ml, window_len=wl, windows_per_batch=per_batch) as c:
        peaks, fine = None, None
        for s, cap in enumerate(caps):
            c.capture_upload(s, cap)
        peaks, fine = c.process_fine(120.0)
        assert peaks.shape == (6, 3)
        for wid in range(6):
            off = (wid // 2) * block + (wid % 2) * wl
            pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0] for cp in caps]
            for p, (i, j) in enumerate(pairs):
                olag, ocorr = oracle.b_xcorr_peak(pre[i], pre[j], ml)
                assert peaks[wid, p]["lag"] == olag == delays[j] - delays[i]
                assert abs(peaks[wid, p]["corr"] - ocorr) <= REL_TOL * abs(ocorr)
                of = oracle.b_refine_peak(pre[i], pre[j], olag, 120.0)
                assert abs(fine[wid, p]["frac"] - of["frac"]) < 1e-4
        c.debug_flags(no_segment_form=True)
        shares = c.process()
        assert np.array_equal(shares["lag"], peaks["lag"])
        assert np.abs(shares["corr"] - peaks["corr"]).max() <= REL_TOL * np.abs(peaks["corr"]).max()
        c.debug_flags(no_short_lag=True)
        general = c.process()
        assert np.array_equal(general["lag"], peaks["lag"])
        assert np.abs(general["corr"] - peaks["corr"]).max() <= REL_TOL * np.abs(peaks["corr"]).max()


@pytest.mark.parametrize("n1,n2,max_lag", [(70_000, 70_000, 20000), (300_001, 299_999, 5000), (2_000_000, 1_999_999, 20000),
                                           (1_100_000, 1_100_000, 20000)])
def test_fused_k1_vs_materialised_codes(oracle, n1, n2, max_lag):
    """default path on the hot plans: the forward column kernels read the capture bytes and evaluate the discriminator
    themselves (k_fwd_col256_k1), the window sums come from the reduce-only pre-pass; with TDOA_DEBUG_NO_FUSED_K1 the
    codes are written to memory (int32) and read back (k_fwd_col256_c16).  Same integers and the same normalised samples;
    the two column kernels are different instruction streams (the compiler contracts multiply-adds where it likes), so
    the lags agree to float32 rounding, not bit for bit; odd lengths (a last element with one sample) included"""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n1, 0, 31, 1)
    b = oracle.simulate_delayed_fm(n2, 41, 31, 2)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, max_lag)
    with tdoa_amd.Context(max_lag=max_lag, window_len=max(n1, n2)) as c:
        c.debug_flags()
        fused = c.fm_xcorr_lags(a, b, max_lag)
        lag, corr = c.fm_xcorr(a, b, max_lag)
        c.debug_flags(no_fused_k1=True)
        stored = c.fm_xcorr_lags(a, b, max_lag)
        lag2, corr2 = c.fm_xcorr(a, b, max_lag)
    _assert_lags_close(fused, want)
    _assert_lags_close(fused, stored, 1e-6)
    assert lag == lag2 == olag == 41 and abs(corr - corr2) <= 1e-6 * abs(corr)
    assert abs(corr - ocorr) <= REL_TOL * abs(ocorr)


def test_fused_k1_in_the_batched_path(oracle):
    """3 stations / 3 pairs on simulator.go captures (exact reversals on 2 % of the samples): fused K1 against materialised
    codes through tdoa_process; block starts that are only 2-byte aligned"""
    import tdoa_amd
    block, wl, ml = 140_001, 70_000, 20000
    caps = [oracle.simulate_station(nm, block, oracle.SEED_BASE + i, tx_power=200000.0) for i, nm in enumerate(oracle.COLLECTORS)]
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        c.debug_flags()
        fused = c.process_u8(caps)
        c.debug_flags(no_fused_k1=True)
        stored = c.process()
    assert fused.shape == (6, 3)
    assert np.array_equal(fused["lag"], stored["lag"])
    assert np.abs(fused["corr"] - stored["corr"]).max() <= 1e-6 * np.abs(stored["corr"]).max()


@pytest.mark.parametrize("block,plan", [(1_100_001, (4096, 256)), (2_200_001, (4096, 512))])
def test_fused_column_kernel_on_odd_window_starts_at_the_timed_plan(oracle, block, plan):
    """The kernels of the timed path (4096 x 256 plan; 4096 x 512: config 5's; decimated inverse) on windows that start on ODD samples: blocks of an
    odd number of samples put every window of blocks 1 and 2 at a byte offset that is 2 modulo 4 -- the 1024-thread column
    kernel reads them with 2-byte-aligned dword loads, takes its boundary samples from the tile to the left, and its tiles
    are dealt XCD by XCD.  Against the f64 oracle on the same slices, and against the materialised-code path."""
    import tdoa_amd
    wl, ml = block, 20000
    caps = [oracle.simulate_delayed_fm(3 * block, d, 314, 70 + i) for i, d in enumerate((0, 29))]
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        fused = c.process_u8(caps)
        assert tuple(c.plan_info())[1:] == plan
        c.debug_flags(no_fused_k1=True)
        stored = c.process()
    assert fused.shape == (3, 1)
    assert np.array_equal(fused["lag"], stored["lag"])
    assert np.abs(fused["corr"] - stored["corr"]).max() <= 1e-6 * np.abs(stored["corr"]).max()
    for wid in range(3):
        off = wid * block
        ta, _ = oracle.b_preprocess(caps[0][2 * off:2 * (off + wl)])
        tb, _ = oracle.b_preprocess(caps[1][2 * off:2 * (off + wl)])
        olag, ocorr, _ = oracle.b_xcorr_peak_fft(ta, tb, ml)
        assert fused[wid, 0]["lag"] == olag == 29
        assert abs(fused[wid, 0]["corr"] - ocorr) <= REL_TOL * abs(ocorr)


@pytest.mark.parametrize("n1,n2,max_lag,delay", [(300_000, 300_000, 512, 77), (123_457, 99_991, 200, -150),
                                                 (50_001, 50_000, 1023, 1000), (2_000_000, 2_000_000, 512, -333)])
def test_segment_form_ragged_and_full_size(oracle, n1, n2, max_lag, delay):
    """segment form on unequal / odd lengths (frames that run off either window) and at the BASELINE window length"""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n1, max(0, -delay), 88, 1)
    b = oracle.simulate_delayed_fm(n2, max(0, delay), 88, 2)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, max_lag)
    assert olag == delay
    with tdoa_amd.Context(max_lag=max_lag, window_len=max(n1, n2)) as c:
        lags = c.fm_xcorr_lags(a, b, max_lag)
        lag, corr = c.fm_xcorr(a, b, max_lag)
    _assert_lags_close(lags, want)
    assert lag == olag and abs(corr - ocorr) <= REL_TOL * abs(ocorr)


@pytest.mark.parametrize("n1,n2,max_lag", [(300_000, 300_000, 512), (123_457, 99_991, 200), (50_001, 50_000, 1023),
                                           (2_000_000, 1_999_999, 256), (4097, 8191, 100), (2, 5000, 64)])
def test_segment_form_packed_code_rows_are_bit_identical(oracle, n1, n2, max_lag):
    """the segment form reads its code rows at 3 bytes per code (k1_store8_packed / k1_code3_at: a stored code has 24
    significant bits); the int32 rows of round 3 behind TDOA_DEBUG_NO_SEG_PACK3 hold the same codes, so every lag sum is
    the same float -- on random bytes (every code value, both signs), odd lengths and windows shorter than a frame"""
    import tdoa_amd
    rng = np.random.default_rng(n1 + max_lag)
    a = rng.integers(0, 256, size=2 * n1, dtype=np.uint8)
    b = rng.integers(0, 256, size=2 * n2, dtype=np.uint8)
    k = min(n1, n2) // 2
    b[2 * 7:2 * (7 + k)] = a[:2 * k]                                # a common stretch: a real peak at lag 7
    with tdoa_amd.Context(max_lag=max_lag, window_len=max(n1, n2)) as c:
        packed, peak_p = c.fm_xcorr_lags(a, b, max_lag), c.fm_xcorr(a, b, max_lag)
        c.debug_flags(no_seg_pack3=True)
        plain, peak_i = c.fm_xcorr_lags(a, b, max_lag), c.fm_xcorr(a, b, max_lag)
        c.debug_flags(no_segment_quads=True)
        pair_p = c.fm_xcorr_lags(a, b, max_lag)
    assert np.array_equal(packed, plain) and peak_p == peak_i
    assert np.array_equal(pair_p, packed)
    if k > 4 * max_lag:
        assert peak_p[0] == 7


@pytest.mark.parametrize("n_st,ml,per_batch", [(3, 512, 0), (3, 100, 2), (4, 200, 0), (5, 1000, 0), (8, 120, 4)])
def test_segment_quads_vs_one_pair_at_a_time(oracle, n_st, ml, per_batch):
    """segment form with station transforms shared by the pairs of a window (k_xcorr_segments_quad: two packed
    transforms per segment serve up to four pair-windows) against the same batch one pair-window at a time
    (k_xcorr_segments), against the general four-step form, and against the f64 oracle"""
    import tdoa_amd
    blk, wl = 140_000, 70_000                                      # N = 2^17: 4096 x 16; 23-35 frames per window
    delays = [0, 41, -17, 5, 50, -40, 23, 9][:n_st]
    caps = [np.concatenate([oracle.simulate_delayed_fm(blk, 100 + d, 910 + k, 10 * s + k) for k in range(3)])
            for s, d in enumerate(delays)]
    pairs = [(i, j) for i in range(n_st) for j in range(i + 1, n_st)]
    with tdoa_amd.Context(max_lag=ml, window_len=wl, windows_per_batch=per_batch) as c:
        for s, cap in enumerate(caps):
            c.capture_upload(s, cap)
        quads, fine_q = c.process_fine(120.0)
        assert quads.shape == (6, len(pairs))
        c.debug_flags(no_segment_quads=True)
        single, fine_s = c.process_fine(120.0)
        c.debug_flags(no_segment_form=True, no_short_lag=True)
        general = c.process()
        # sharded: window-major (2 ranks) and pair-major (more ranks than windows: odd pair subsets per rank)
        c.debug_flags()
        for world in (2, 7):
            merged = np.zeros_like(quads)
            for r in range(world):
                part = c.process(rank=r, world=world)
                own = part["corr"] != 0
                assert not (own & (merged["corr"] != 0)).any()
                merged[own] = part[own]
            # (the chunking of a window's segments follows the size of the rank's batch: same lags, sums to rounding)
            assert np.array_equal(merged["lag"], quads["lag"]), world
            assert np.abs(merged["corr"] - quads["corr"]).max() <= 2e-6 * np.abs(quads["corr"]).max(), world
    scale = np.abs(quads["corr"]).max()
    for other in (single, general):
        assert np.array_equal(other["lag"], quads["lag"])
        assert np.abs(other["corr"] - quads["corr"]).max() <= 2e-6 * scale
    assert np.abs(fine_q["frac"] - fine_s["frac"]).max() < 1e-4
    for wid in (0, 5):
        off = (wid // 2) * blk + (wid % 2) * wl
        pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0] for cp in caps]
        for p, (i, j) in enumerate(pairs):
            olag, ocorr = oracle.b_xcorr_peak(pre[i], pre[j], ml)
            assert quads[wid, p]["lag"] == olag == delays[j] - delays[i], (wid, i, j)
            assert abs(quads[wid, p]["corr"] - ocorr) <= REL_TOL * abs(ocorr)


@pytest.mark.parametrize("n1,n2,max_lag,delay", [(2_000_000, 2_000_000, 20000, 57), (1_234_567, 1_999_999, 20000, -19876),
                                                 (1_500_000, 1_100_000, 4096, 4001), (2_000_000, 2_000_000, 23000, 9),
                                                 (2_000_000, 2_000_000, 26000, -25001), (4_000_000, 3_999_000, 20000, 1234),
                                                 (2_200_000, 3_100_000, 20000, -19999)])
def test_decimated_inverse_vs_full_inverse(oracle, n1, n2, max_lag, delay):
    """4096 x 256 and 4096 x 512 plans, search ranges above 4095 lags: K3 + FIR decimation of the pair's spectrum + an
    Nc/16-point inverse (k_pair_decimate16) against the full inverse (k_inv_row_pair4096 + k_inv_col_pruned), every lag,
    and both against the f64 oracle.  26000 lags on the 4096 x 256 plan leave no room for the transition band: that
    range must fall back by itself."""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n1, max(0, -delay), 91, 1)
    b = oracle.simulate_delayed_fm(n2, max(0, delay), 91, 2)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, max_lag)
    assert olag == delay
    with tdoa_amd.Context(max_lag=max_lag, window_len=max(n1, n2)) as c:
        dec = c.fm_xcorr_lags(a, b, max_lag)
        lag, corr = c.fm_xcorr(a, b, max_lag)
        _, fine = c.fm_xcorr_fine(a, b, max_lag, 1e9)
        assert c.plan_info()[1:] == ((4096, 256) if max(n1, n2) <= 2_000_000 else (4096, 512))
        c.debug_flags(no_decimate=True)
        full = c.fm_xcorr_lags(a, b, max_lag)
        lag_f, corr_f = c.fm_xcorr(a, b, max_lag)
        _, fine_f = c.fm_xcorr_fine(a, b, max_lag, 1e9)
    _assert_lags_close(dec, want)
    _assert_lags_close(full, want)
    _assert_lags_close(dec, full, 2e-6)
    assert lag == lag_f == olag
    assert abs(corr - ocorr) <= REL_TOL * abs(ocorr) and abs(corr_f - ocorr) <= REL_TOL * abs(ocorr)
    # sub-sample refinement: the three neighbours come from the small plan's row-pass output, window divided out
    ofine = oracle.b_refine_peak(ta, tb, olag, 1e9)
    assert np.abs(fine["y"] - ofine["y"]).max() <= REL_TOL * abs(ocorr) and np.abs(fine_f["y"] - ofine["y"]).max() <= REL_TOL * abs(ocorr)
    assert abs(fine["frac"] - ofine["frac"]) < 1e-4 and abs(fine_f["frac"] - ofine["frac"]) < 1e-4
    if max_lag == 26000:
        assert np.array_equal(dec, full)          # same kernels: the decimated form did not apply
