"""GPU parity tests of the sub-sample refinement and plausibility gate,
through the C ABI (tdoa_fm_xcorr_fine_u8, tdoa_process_fine) against oracle ob_refine_peak."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5       # neighbours, relative to the peak (north_star tolerance for magnitudes)
FRAC_TOL = 1e-4      # samples; f32 FFT against the f64 time-domain oracle


def _check(fine, ofine, lag):
    peak = abs(ofine["y"][1])
    assert np.abs(np.asarray(fine["y"], dtype=np.float64) - ofine["y"]).max() <= REL_TOL * peak
    assert abs(float(fine["frac"]) - ofine["frac"]) < FRAC_TOL
    assert abs(float(fine["delay"]) - (lag + float(fine["frac"]))) < 1e-6
    assert bool(fine["plausible"]) == ofine["plausible"]


@pytest.mark.parametrize("n,delay,max_lag", [
    (20000, 37, 500),
    (20000, -113, 500),
    (20000, 0, 500),
    (20000, 499, 500),      # peak on the edge of the searched range: neighbour +1 is outside it
    (20000, -499, 500),
    (3001, 5, 64),          # small plan (generic kernels)
    (70000, -1, 600),       # 4096-point rows
])
def test_single_pair_fine_vs_oracle(oracle, n, delay, max_lag):
    import tdoa_amd
    a = oracle.simulate_delayed_fm(n, max(0, -delay), 4242, 1)
    b = oracle.simulate_delayed_fm(n, max(0, delay), 4242, 2)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    with tdoa_amd.Context(max_lag=max_lag, window_len=n) as c:
        for gate in (120.0, 3.0):
            (lag, corr), fine = c.fm_xcorr_fine(a, b, max_lag, gate)
            assert (lag, corr) == c.fm_xcorr(a, b, max_lag)         # the integer peak is untouched
            assert lag == delay
            _check(fine, oracle.b_refine_peak(ta, tb, lag, gate), lag)
            assert fine["plausible"] == (abs(fine["delay"]) <= gate)


def test_fine_negative_peak_and_flat_capture(oracle):
    import tdoa_amd
    n = 20000
    a = oracle.simulate_delayed_fm(n, 0, 77, 1)
    b = oracle.simulate_delayed_fm(n, 9, 77, 2)
    binv = b.copy()
    binv[0::2], binv[1::2] = b[1::2], b[0::2]            # swap I and Q: phase runs backwards, c < 0
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(binv)
    with tdoa_amd.Context(max_lag=100, window_len=n) as c:
        (lag, corr), fine = c.fm_xcorr_fine(a, binv, 100, 120.0)
        olag, ocorr = oracle.b_xcorr_peak(ta, tb, 100)
        assert lag == olag and corr < 0 and ocorr < 0
        assert fine["y"][1] > 0                           # y is sign-normalised
        _check(fine, oracle.b_refine_peak(ta, tb, lag, 120.0), lag)
        flat = np.full(2 * 64, 128, np.uint8)
        (lag, corr), fine = c.fm_xcorr_fine(flat, flat, 10, 120.0)
        assert (lag, corr) == (0, 0.0)
        assert fine["delay"] == 0.0 and fine["frac"] == 0.0 and not fine["y"].any() and fine["plausible"]
        (lag, corr), fine = c.fm_xcorr_fine(np.zeros(0, np.uint8), a, 10, 120.0)     # processor.go:622-625
        assert (lag, corr) == (0, 0.0) and fine["delay"] == 0.0 and fine["plausible"]


@pytest.mark.parametrize("per_batch", [0, 2])
def test_process_fine_batched(oracle, per_batch):
    import tdoa_amd
    block, wl, ml = 30000, 10000, 300
    delays = [0, 41, 17]
    blocks = []
    for st, d in enumerate(delays):
        blocks.append(np.concatenate([oracle.simulate_delayed_fm(block, d, 500 + k, 10 * st + k) for k in range(3)]))
    pairs = [(0, 1), (0, 2), (1, 2)]
    with tdoa_amd.Context(max_lag=ml, window_len=wl, windows_per_batch=per_batch) as c:
        for s, cap in enumerate(blocks):
            c.capture_upload(s, cap)
        base = c.process()
        peaks, fine = c.process_fine(30.0)
        assert np.array_equal(peaks, base)
        assert fine.shape == (9, 3)
        for wid in range(9):
            off = (wid // 3) * block + (wid % 3) * wl
            pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0] for cp in blocks]
            for p, (i, j) in enumerate(pairs):
                lag = int(peaks[wid, p]["lag"])
                assert lag == delays[j] - delays[i]
                _check(fine[wid, p], oracle.b_refine_peak(pre[i], pre[j], lag, 30.0), lag)
        assert (fine[:, 0]["plausible"] == 0).all()       # pair (0,1): 41 samples > gate 30
        assert (fine[:, 1]["plausible"] == 1).all()       # pair (0,2): 17
        assert (fine[:, 2]["plausible"] == 1).all()       # pair (1,2): -24
        # replay of the captured graph gives the same records; a different gate re-captures
        peaks2, fine2 = c.process_fine(30.0)
        assert np.array_equal(fine2, fine) and np.array_equal(peaks2, peaks)
        _, fine3 = c.process_fine(50.0)
        assert (fine3["plausible"] == 1).all()
        assert np.array_equal(fine3["frac"], fine["frac"])
        # sharded: a rank refines only the windows it owns
        for r in range(2):
            pk, fn = c.process_fine(30.0, rank=r, world=2)
            for wid in range(9):
                if wid % 2 == r:
                    assert np.array_equal(fn[wid], fine[wid])
                else:
                    assert not fn[wid]["delay"].any() and not fn[wid]["y"].any()


def test_full_size_fine_hot_kernels(oracle):
    """BASELINE config 2 geometry (N = 2^21, pruned inverse): neighbours against the f64 FFT oracle."""
    import tdoa_amd
    n = 2_000_000
    a = oracle.simulate_delayed_fm(n, 0, 99, 3)
    b = oracle.simulate_delayed_fm(n, 57, 99, 4)
    ta, _ = oracle.b_preprocess(a)
    tb, _ = oracle.b_preprocess(b)
    olag, ocorr, want = oracle.b_xcorr_peak_fft(ta, tb, 20000)
    with tdoa_amd.Context() as c:
        (lag, corr), fine = c.fm_xcorr_fine(a, b, 20000, 120.0)
    assert lag == olag == 57
    sg = 1.0 if ocorr >= 0 else -1.0
    oy = sg * want[[56 + 19999, 57 + 19999, 58 + 19999]]
    assert np.abs(fine["y"] - oy).max() <= REL_TOL * abs(ocorr)
    assert abs(fine["frac"] - oracle.b_parabola_vertex(*oy)) < FRAC_TOL
    assert fine["plausible"]
