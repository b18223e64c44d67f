"""GPU: the transform length of ten-second windows (BASELINE config 3) -- N = 5 x 2^22 instead of 2^25 (round 5).

The reference's padding rule, the next power of two (processor.go:563), is dead code there (processor.go:638) and a design
hint here: ANY N >= window + search range gives the same linear correlation.  20 000 000 samples + 20 000 lags sit in
2^25 = 33 554 432 points with 40 % zero padding that every pass moves; 5 x 2^22 = 20 971 520 (packed 4096 x 2560) holds them
with 4.5 %.  New arithmetic on that plan: a 10-point finish of the two-sweep column pass (2 x 5-point DFTs), twiddles with a
denominator 5 x 2^k (unit_root_any: the quadrant reduction on integers), the column walk of the decimated pair step down 2560
rows, a 4096 x 160 small plan (column roots W_160).  Checked here:
  * EVERY lag against the f64 oracle (ob_* codes, float64 FFT correlation) at the window lengths that select the plan --
    its fullest window, BASELINE's 20 000 000, an odd length, the shortest one (2^24 + 1 points needed), unequal lengths --
    through the single-look path, the pre-pass path and materialised codes (k_fwd_col256_c16<true> writes the sub-transforms);
  * the same windows in N = 2^25 (TDOA_DEBUG_POW2_ONLY): lag arrays equal to rounding;
  * the window lengths either side of the plans' ranges take the next plan up (2^24 below, 3 x 2^23 above 5 x 2^22, 2^25 above that);
  * N = 3 x 2^23 (4096 x 3072, a 12-point finish) for windows of 10.5 to 12.6 s, the same way;
  * the sub-sample refinement on the small plan 4096 x 160 (k_refine_peaks with W_160);
  * a batch through tdoa_process (device-generated weak-simulator captures, three windows x three pairs) on both plans and
    against the oracle and the float64 atan2 pipeline."""
import numpy as np
import pytest

from oracle import float_pipeline as fp

pytestmark = pytest.mark.gpu

ML = 20000
N5 = 5 << 22                                    # 20 971 520
N3 = 3 << 23                                    # 25 165 824: windows between 10.5 and 12.6 s at 2 Msps
PLAN5 = (N5, 4096, 2560)
PLAN3 = (N3, 4096, 3072)
PLAN25 = (1 << 25, 4096, 4096)


def _lags_close(got, want, tol):
    peak = np.abs(want).max()
    err = np.abs(got - want).max() / peak
    assert err < tol, "max lag error %.3g of the peak" % err
    return err


@pytest.mark.parametrize("n1,n2,delay,f64", [
    (20_000_000, 20_000_000, 88, True),                      # BASELINE config 3's window
    (N5 - ML, N5 - ML, -19999, True),                        # the fullest window the plan holds: window + lags = N exactly
    (18_000_001, 20_000_000, 7, True),                       # unequal lengths: no single-look K1, the pre-pass by itself
    # (the float64 oracle of a ten-second pair costs half a minute of host time; these two take the 2^25 plan -- itself held
    #  against the oracle above and in test_gpu_fm.py -- as their reference)
    (19_999_999, 19_999_999, 4097, False),                   # odd length: the last element of the window holds one sample
    ((1 << 24) - ML + 1, (1 << 24) - ML + 1, -1, False),     # the shortest: 2^24 + 1 points needed
])
def test_every_lag_vs_f64_oracle_on_the_5x2p22_plan(oracle, n1, n2, delay, f64, capsys):
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n1, max(0, -delay), 61, 1)
    b = oracle.simulate_delayed_fm(n2, max(0, delay), 61, 2)
    if f64:
        ta, _ = oracle.b_preprocess(a)
        tb, _ = oracle.b_preprocess(b)
        olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, ML)
        assert olag == delay
    else:
        with tdoa_amd.Context(max_lag=ML, window_len=max(n1, n2)) as c:
            c.debug_flags(pow2_only=True)
            want = c.fm_xcorr_lags(a, b, ML)
            olag, ocorr = c.fm_xcorr(a, b, ML)
            assert tuple(c.plan_info()) == PLAN25 and olag == delay
    with tdoa_amd.Context(max_lag=ML, window_len=max(n1, n2)) as c:
        lags, peak = c.fm_xcorr_lags(a, b, ML), c.fm_xcorr(a, b, ML)
        assert tuple(c.plan_info()) == PLAN5
        assert c.last_k1(0)[1] == (n1 == n2)
        e0 = _lags_close(lags, want, 1e-5)
        assert peak[0] == delay and abs(peak[1] - ocorr) <= 1e-5 * abs(ocorr)
        c.debug_flags(no_k1_once=True)                              # statistics pre-pass + discriminator in the column pass
        e1 = _lags_close(c.fm_xcorr_lags(a, b, ML), want, 1e-5)
        assert tuple(c.plan_info()) == PLAN5 and not c.last_k1(0)[1]
        c.debug_flags(no_fused_k1=True)                             # materialised codes: k_fwd_col256_c16<true> + the 10-point finish
        e2 = _lags_close(c.fm_xcorr_lags(a, b, ML), want, 1e-5)
        assert tuple(c.plan_info()) == PLAN5
        c.debug_flags(pow2_only=True)                               # the same window in N = 2^25
        p2 = c.fm_xcorr_lags(a, b, ML)
        assert tuple(c.plan_info()) == PLAN25
        e3 = _lags_close(p2, want, 1e-5)
        _lags_close(lags, p2, 2e-6)
    with capsys.disabled():
        print("\n  L = %d / %d on 4096 x 2560: all %d lags vs %s %.2e (single look) %.2e (pre-pass) %.2e (codes); "
              "2^25 plan %.2e" % (n1, n2, 2 * ML - 1, "f64 oracle" if f64 else "the 2^25 plan", e0, e1, e2, e3))


@pytest.mark.parametrize("n1,n2,delay,f64", [(23_000_000, 23_000_000, -4097, True), (N3 - ML, N3 - ML, 19999, False),
                                             (N5 - ML + 1, 22_222_221, 77, False)])
def test_every_lag_on_the_3x2p23_plan(oracle, n1, n2, delay, f64, capsys):
    """N = 3 x 2^23 (4096 x 3072: twelve 256-point sub-transforms, a 12-point finish as 3 x 4, denominators 3 x 2^k, a
    4096 x 192 small plan) for windows that need between 5 x 2^22 + 1 and 25 165 824 points: every lag against the f64 oracle
    (once: half a minute of host time) or against the same window in N = 2^25"""
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n1, max(0, -delay), 67, 1)
    b = oracle.simulate_delayed_fm(n2, max(0, delay), 67, 2)
    with tdoa_amd.Context(max_lag=ML, window_len=max(n1, n2)) as c:
        lags, peak = c.fm_xcorr_lags(a, b, ML), c.fm_xcorr(a, b, ML)
        assert tuple(c.plan_info()) == PLAN3 and c.last_k1(0)[1] == (n1 == n2)
        c.debug_flags(no_fused_k1=True)
        codes = c.fm_xcorr_lags(a, b, ML)
        assert tuple(c.plan_info()) == PLAN3
        c.debug_flags(pow2_only=True)
        p2, p2_peak = c.fm_xcorr_lags(a, b, ML), c.fm_xcorr(a, b, ML)
        assert tuple(c.plan_info()) == PLAN25
    assert peak[0] == p2_peak[0] == delay
    e_p2 = _lags_close(lags, p2, 2e-6)
    _lags_close(codes, p2, 2e-6)
    if f64:
        ta, _ = oracle.b_preprocess(a)
        tb, _ = oracle.b_preprocess(b)
        olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, ML)
        assert olag == delay and abs(peak[1] - ocorr) <= 1e-5 * abs(ocorr)
        e = _lags_close(lags, want, 1e-5)
        with capsys.disabled():
            print("\n  L = %d on 4096 x 3072: all %d lags vs f64 oracle %.2e, vs the 2^25 plan %.2e" % (n1, 2 * ML - 1, e, e_p2))


@pytest.mark.parametrize("n,plan", [((1 << 24) - ML, (1 << 24, 4096, 2048)), (N5 - ML + 1, PLAN3), (N3 - ML + 1, PLAN25)])
def test_window_lengths_either_side_keep_their_power_of_two(oracle, n, plan):
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n, 0, 13, 1)
    b = oracle.simulate_delayed_fm(n, 55, 13, 2)
    with tdoa_amd.Context(max_lag=ML, window_len=n) as c:
        assert c.fm_xcorr(a, b, ML)[0] == 55
        assert tuple(c.plan_info()) == plan


def test_where_the_5x2p22_plan_does_not_apply(oracle):
    """short search ranges, TDOA_LAGS_GO, the full inverse and the any-size kernels have no 5 x 2^k form: they keep N = 2^25"""
    import tdoa_amd
    n = 17_000_000
    a = oracle.simulate_delayed_fm(n, 0, 14, 1)
    b = oracle.simulate_delayed_fm(n, 300, 14, 2)
    with tdoa_amd.Context(max_lag=ML, window_len=n) as c:
        assert c.fm_xcorr(a, b, 600)[0] == 300                       # segment form
        assert tuple(c.plan_info()) == PLAN25
        assert c.fm_xcorr(a, b, 3000)[0] == 300                      # short-lag rows
        assert tuple(c.plan_info()) == PLAN25
        assert c.fm_xcorr(a, b, ML)[0] == 300
        assert tuple(c.plan_info()) == PLAN5
        c.debug_flags(no_decimate=True)
        assert c.fm_xcorr(a, b, ML)[0] == 300 and tuple(c.plan_info()) == PLAN25
        c.debug_flags(no_dec_cols=True)
        assert c.fm_xcorr(a, b, ML)[0] == 300 and tuple(c.plan_info()) == PLAN25
    with tdoa_amd.Context(max_lag=ML, window_len=n, lag_mode=1) as c:          # TDOA_LAGS_GO: equal lengths, lag 0 only
        assert c.fm_xcorr(a, b, ML)[0] == 0 and tuple(c.plan_info()) == PLAN25


def test_go_lag_set_and_wide_ranges_on_ten_second_windows(oracle):
    """TDOA_LAGS_GO with unequal lengths searches [0, min(maxLag, Ls - Lt)): 20 000 non-negative lags -- the decimated path on
    the 5 x 2^22 plan with a one-sided output set (k_small_col_peak's run-time form, W_160 by index); a search range too wide
    for the pruned column outputs (40 000 lags: ten of them) has no 5 x 2^k form and keeps N = 2^25.  Both against the same
    call in N = 2^25."""
    import tdoa_amd
    nt, ns, delay = 18_000_000, 20_000_000, 12345
    a = oracle.simulate_delayed_fm(nt, 0, 19, 1)
    b = oracle.simulate_delayed_fm(ns, delay, 19, 2)
    with tdoa_amd.Context(max_lag=ML, window_len=ns, lag_mode=tdoa_amd.capi.LAGS_GO) as c:
        lag, corr = c.fm_xcorr(a, b, ML)
        lags = c.fm_xcorr_lags(a, b, ML)
        assert tuple(c.plan_info()) == PLAN5
        c.debug_flags(pow2_only=True)
        lag2, corr2 = c.fm_xcorr(a, b, ML)
        lags2 = c.fm_xcorr_lags(a, b, ML)
        assert tuple(c.plan_info()) == PLAN25
    assert lag == lag2 == delay and abs(corr - corr2) <= 2e-6 * abs(corr2)
    assert not lags[:ML - 1].any() and np.abs(lags - lags2).max() <= 2e-6 * np.abs(lags2).max()      # no negative lag is reported
    wide = 40000
    with tdoa_amd.Context(max_lag=wide, window_len=ns) as c:
        lagw, corrw = c.fm_xcorr(a[:2 * 17_000_000], b[:2 * 17_000_000], wide)
        assert tuple(c.plan_info()) == PLAN25
    assert lagw == delay


def test_refinement_on_the_small_plan_4096x160(oracle):
    """tdoa_fm_xcorr_fine_u8 on a ten-second window: the three neighbours come from the 4096 x 160 small plan's row-pass output
    (k_refine_peaks: 160 values per column, W_160); against ob_refine_peak's parabola on values from the f64 FFT and against
    the same window in N = 2^25"""
    import tdoa_amd
    n, delay = 20_000_000, -12345
    a = oracle.simulate_delayed_fm(n, -delay, 17, 1)
    b = oracle.simulate_delayed_fm(n, 0, 17, 2)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    _, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, ML)
    y = want[delay + ML - 2:delay + ML + 1] * np.sign(ocorr)
    with tdoa_amd.Context(max_lag=ML, window_len=n) as c:
        (lag, corr), fine = c.fm_xcorr_fine(a, b, ML, 25000.0)
        assert tuple(c.plan_info()) == PLAN5
        c.debug_flags(pow2_only=True)
        (lag2, corr2), fine2 = c.fm_xcorr_fine(a, b, ML, 25000.0)
        assert tuple(c.plan_info()) == PLAN25
    assert lag == lag2 == delay and abs(corr - ocorr) <= 1e-5 * abs(ocorr)
    assert np.abs(np.asarray(fine["y"], dtype=np.float64) - y).max() <= 1e-5 * abs(ocorr)
    assert abs(float(fine["frac"]) - oracle.b_parabola_vertex(*y)) < 1e-4
    assert np.abs(np.asarray(fine["y"]) - np.asarray(fine2["y"])).max() <= 2e-6 * abs(ocorr)
    assert abs(float(fine["delay"]) - float(fine2["delay"])) < 1e-4 and fine["plausible"]


def test_cfg3_batch_on_both_plans(oracle, capsys):
    """BASELINE config 3 in miniature: three weak-simulator stations generated in HBM, ONE ten-second window per block -- three
    windows x three pairs through tdoa_process (single-look K1, two-sweep column pass with the 10-point finish, row pass,
    column walk down 2560 rows, 4096 x 160 inverse, K5) -- against the same batch in N = 2^25 and, for the target-block window,
    against the oracle and the float64 atan2 pipeline on the downloaded bytes"""
    import tdoa_amd
    L = 20_000_000
    with tdoa_amd.Context(max_lag=ML, window_len=L) as c:
        for s, nm in enumerate(oracle.COLLECTORS):
            c.synth_weak_capture(s, L, oracle.STATIONS[nm], oracle.DEFAULT_TX, oracle.SEED_BASE + s, tgt_power=20000.0)
        peaks = c.process()
        assert tuple(c.plan_info()) == PLAN5 and c.last_k1(0)[1]
        again = c.process()                                          # the replayed graph
        fpk, fine = c.process_fine(25000.0)                          # + the refinement on the 4096 x 160 small plan, batched
        c.debug_flags(pow2_only=True)
        p2 = c.process()
        assert tuple(c.plan_info()) == PLAN25
        fpk2, fine2 = c.process_fine(25000.0)
        tgt = [c.capture_download(s, L, L) for s in range(3)]
    assert peaks.shape == (3, 3) and np.array_equal(peaks, again)
    assert np.array_equal(fpk, peaks) and np.array_equal(fpk2, p2)     # the integer peaks are untouched by the refinement
    assert np.abs(fine["delay"][1] - fine2["delay"][1]).max() < 1e-4 and fine["plausible"][1].all()
    assert np.abs(fine["y"][1] - fine2["y"][1]).max() <= 2e-6 * np.abs(p2["corr"][1]).max()
    assert np.array_equal(peaks["lag"], p2["lag"])
    assert (np.abs(peaks["corr"] - p2["corr"]) <= 2e-6 * np.abs(p2["corr"])).all()
    assert not peaks[[0, 2]]["lag"].any() and not peaks[[0, 2]]["corr"].any()          # constant reference blocks: (0, 0.0)
    pre = [oracle.b_preprocess(x)[0] for x in tgt]
    rows = []
    for p, (i, j) in enumerate(((0, 1), (0, 2), (1, 2))):
        olag, ocorr, _ = oracle.b_xcorr_peak_fft(pre[i], pre[j], ML)
        g = peaks[1, p]
        assert int(g["lag"]) == olag and abs(float(g["corr"]) - ocorr) <= 1e-5 * abs(ocorr)
        rows.append((i, j, olag, float(g["corr"]), abs(float(g["corr"]) - ocorr) / abs(ocorr)))
    flag, fcorr, _ = fp.xcorr_peak_u8(tgt[0], tgt[1], ML)
    fdev = abs(float(peaks[1, 0]["corr"]) - fcorr) / abs(fcorr)
    assert int(peaks[1, 0]["lag"]) == flag and fdev < 1e-5
    with capsys.disabled():
        print("\n  cfg3 batch on 4096 x 2560 (target-block window):")
        for r in rows:
            print("    pair %d-%d lag %6d corr %12.5f  vs ob_* %.2e" % r)
        print("    pair 0-1 vs float64 atan2 pipeline %.2e" % fdev)
