"""The decimated inverse's small plan in one pass (k_small_rows_col_peak, fft_radix8.hpp): one workgroup per pair-window runs the
row transforms of G and keeps the six column sums that can hold a searched lag in registers -- V' never reaches memory.
Reference: the two kernels it replaces (k_inv_rows_plain_r8 -> V' -> k_small_col_peak<3, 3>, TDOA_DEBUG_NO_SMALL_FUSED), which the
other GPU tests hold against the oracle; the sums run in the same order with the same factors, so every lag value and every
peak must carry the same bits.  Also against the oracle's float64 FFT directly on one pair."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ML = 20000


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle as o
    o.build()
    return o


@pytest.mark.parametrize("wl", [1_100_000, 2_200_001])
def test_lag_arrays_identical_with_the_two_kernel_path(oracle, wl):
    """one pair, all 39 999 lags (tile form of the pair step, shares from E): 4096 x 16 and 4096 x 32 small plans"""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(wl, 0, 31, 1)
    b = oracle.simulate_delayed_fm(wl, 173, 31, 2)
    with tdoa_amd.Context(max_lag=ML, window_len=wl) as c:
        c.debug_flags(small_fused_always=True)                       # (the library's own rule: from 1024 pair-windows per launch on)
        fused = c.fm_xcorr_lags(a, b, ML)
        peak = c.fm_xcorr(a, b, ML)
        c.debug_flags(no_small_fused=True)
        two = c.fm_xcorr_lags(a, b, ML)
        peak2 = c.fm_xcorr(a, b, ML)
    assert fused.shape == two.shape == (2 * ML - 1,)
    assert np.array_equal(fused, two) and np.abs(fused).max() > 100.0
    assert peak == peak2 and peak[0] == 173
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, _ = oracle.b_xcorr_peak_fft(ta, tb, ML)
    assert peak[0] == olag and abs(peak[1] - ocorr) <= 1e-5 * abs(ocorr)


@pytest.mark.parametrize("n_stations,wl,staged", [(3, 1_100_000, True), (3, 1_100_000, False), (5, 1_100_000, True), (4, 2_200_001, True)])
def test_batch_peaks_identical_with_the_two_kernel_path(oracle, n_stations, wl, staged):
    """batches through tdoa_process (single-look K1: the residual-mean terms are added inside the kernel): the pair step as the
    staged column walk (shares from X, by column) and as tiles; every peak record identical, every lag the geometry's"""
    import tdoa_amd
    rng = np.random.default_rng(7 + n_stations)
    delays = [int(x) for x in rng.integers(0, 400, size=n_stations)]
    caps = [np.concatenate([oracle.simulate_delayed_fm(wl, d, 500 + k, 100 * (s + 1) + k) for k in range(3)]) for s, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=ML, window_len=wl) as c:
        c.debug_flags(no_dec_cols=not staged, small_fused_always=True)
        fused = c.process_u8(caps)
        assert c.last_k1(0)[1]
        c.debug_flags(no_dec_cols=not staged, no_small_fused=True)
        two = c.process()
        fine_two = c.process_fine(4.0)
        c.debug_flags(no_dec_cols=not staged, small_fused_always=True)
        fine_fused = c.process_fine(4.0)
    assert np.array_equal(fused, two)
    assert np.array_equal(fine_fused[0], fine_two[0]) and np.array_equal(fine_fused[1], fine_two[1])
    want = np.array([delays[j] - delays[i] for i in range(n_stations) for j in range(i + 1, n_stations)])
    assert (fused["lag"] == want[None, :]).all() and (fused["abs_corr"] > 100.0).all()
