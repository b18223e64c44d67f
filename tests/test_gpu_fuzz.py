"""Seeded sweep over window lengths, length mismatches and lag ranges: every plan variant the dispatcher can pick
(short / long column kernels, short-lag and pruned inverse, any-size fallbacks) against the any-size kernels, which the
small-size tests pin to the time-domain oracle.  A disagreement here is an indexing bug in one of the hot variants."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def _cases():
    rng = np.random.default_rng(20260101)
    out = []
    for _ in range(36):
        logn = rng.integers(16, 21)                       # FFT sizes 2^17 .. 2^21 (4096-point rows) and one below
        n = int(rng.integers((1 << logn) // 2 + 10, (1 << logn) - 10))
        cut = int(rng.integers(0, min(5000, n // 4)))     # second window shorter by `cut` samples
        max_lag = int(rng.choice([3, 64, 127, 511, 512, 1023, 1024, 2047, 2048, 4095, 4096, 9000, 20000]))
        max_lag = min(max_lag, (1 << logn) - n)           # keep N = nextpow2(n + max_lag) inside the bucket
        if max_lag >= 2:
            out.append((n, cut, max_lag, int(rng.integers(0, 2 ** 31))))
    for logn in (21, 22, 22, 23, 23, 24):                 # the long-column variants (N2 = 256 .. 2048), fewer cases
        n = int(rng.integers((1 << logn) // 2 + 10, (1 << logn) - 20010))
        max_lag = int(rng.choice([300, 4000, 20000]))
        out.append((n, int(rng.integers(0, 3000)), max_lag, int(rng.integers(0, 2 ** 31))))
    return out


@pytest.mark.parametrize("n,cut,max_lag,seed", _cases())
def test_hot_variants_agree_with_any_size_kernels(oracle, n, cut, max_lag, seed):
    import tdoa_amd
    delay = int(seed % (2 * max_lag - 1)) - (max_lag - 1)
    a = oracle.simulate_delayed_fm(n, max(0, -delay), seed & 0xffff, 1)
    b = oracle.simulate_delayed_fm(n, max(0, delay), seed & 0xffff, 2)[:2 * (n - cut)]
    with tdoa_amd.Context(max_lag=max_lag, window_len=n) as c:
        hot = c.fm_xcorr_lags(a, b, max_lag)
        hot_peak = c.fm_xcorr(a, b, max_lag)
        (_, _), fine = c.fm_xcorr_fine(a, b, max_lag, 1e9)
        c.force_generic(True)
        gen = c.fm_xcorr_lags(a, b, max_lag)
        gen_peak = c.fm_xcorr(a, b, max_lag)
        (_, _), gfine = c.fm_xcorr_fine(a, b, max_lag, 1e9)
    peak = np.abs(gen).max()
    assert peak > 0
    assert np.abs(hot - gen).max() <= REL_TOL * peak
    assert hot_peak[0] == gen_peak[0] == delay
    assert abs(hot_peak[1] - gen_peak[1]) <= REL_TOL * abs(gen_peak[1])
    assert abs(fine["frac"] - gfine["frac"]) < 1e-4


@pytest.mark.parametrize("n_st,block,wlen,per_batch,max_lag,seed", [
    (2, 70_000, 70_000, 0, 300, 1), (4, 150_000, 70_000, 1, 5000, 2), (5, 33_000, 11_000, 2, 100, 3),
    (6, 262_144 + 4000, 131_072, 3, 2047, 4), (3, 9_001, 3_000, 0, 50, 5),
])
def test_batched_path_equals_pair_calls(oracle, n_st, block, wlen, per_batch, max_lag, seed):
    """every (window, pair) of tdoa_process is the same arithmetic as one tdoa_fm_xcorr_u8 call on the same bytes:
    the peak records must be identical bit for bit, whatever the station count, launch grouping or plan"""
    import tdoa_amd
    rng = np.random.default_rng(seed)
    delays = [int(x) for x in rng.integers(0, max_lag // 3 + 1, size=n_st)]
    caps = [np.concatenate([oracle.simulate_delayed_fm(block, d, 900 + 7 * seed + k, 100 * s + k) for k in range(3)])
            for s, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=max_lag, window_len=wlen, windows_per_batch=per_batch) as c:
        # flags == 0: the library's default kernels on both sides (the batch and the single pair call)
        peaks = c.process_u8(caps)
        wpb = max(1, block // wlen)
        wl = min(wlen, block)
        assert peaks.shape == (3 * wpb, n_st * (n_st - 1) // 2)
        for wid in range(3 * wpb):
            off = (wid // wpb) * block + (wid % wpb) * wl
            p = 0
            for i in range(n_st):
                for j in range(i + 1, n_st):
                    lag, corr = c.fm_xcorr(caps[i][2 * off:2 * (off + wl)], caps[j][2 * off:2 * (off + wl)], max_lag)
                    assert (int(peaks[wid, p]["lag"]), float(peaks[wid, p]["corr"])) == (lag, corr), (wid, i, j)
                    assert lag == delays[j] - delays[i]
                    p += 1


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_decimated_inverse_fuzz_batched(oracle, seed):
    """seeded sweep of the batched path on 4096 x 256 / 4096 x 512 plans: stations, window length, search range, batch
    split and sharding drawn at random; decimated inverse (default) against the full inverse, lag for lag, and a
    sample of (window, pair) units against the f64 oracle"""
    import tdoa_amd
    rng = np.random.default_rng(seed)
    n_st = int(rng.integers(2, 5))
    wl = int(rng.integers(1_060_000, 2_000_001)) if seed % 3 else int(rng.integers(2_100_000, 3_000_001))
    ml = int(rng.integers(4096, 23001))
    per_batch = int(rng.integers(0, 3))
    half = min((ml - 1) // 2, 9000)                               # every pair's delay difference stays inside the search range
    delays = [0] + [int(x) for x in rng.integers(-half, half, size=n_st - 1)]
    base = 10_000
    caps = [np.concatenate([oracle.simulate_delayed_fm(wl, base + d, 500 + seed + k, 10 * s + k) for k in range(3)])
            for s, d in enumerate(delays)]
    pairs = [(i, j) for i in range(n_st) for j in range(i + 1, n_st)]
    with tdoa_amd.Context(max_lag=ml, window_len=wl, windows_per_batch=per_batch) as c:
        for s, cap in enumerate(caps):
            c.capture_upload(s, cap)
        dec = c.process()
        n_fft, n1, n2 = c.plan_info()
        assert (n1, n2) in ((4096, 256), (4096, 512))
        world = int(rng.integers(2, 5))
        merged = np.zeros_like(dec)
        for r in range(world):
            part = c.process(rank=r, world=world)
            own = part["corr"] != 0
            merged[own] = part[own]
        c.debug_flags(no_decimate=True)
        full = c.process()
    assert dec.shape == (3, len(pairs))
    scale = np.abs(full["corr"]).max()
    for got in (dec, merged):
        assert np.array_equal(got["lag"], full["lag"])
        assert np.abs(got["corr"] - full["corr"]).max() <= 3e-6 * scale
    wid = int(rng.integers(0, 3))
    pre = [oracle.b_preprocess(cp[2 * wid * wl:2 * (wid + 1) * wl])[0] for cp in caps]
    p = int(rng.integers(0, len(pairs)))
    i, j = pairs[p]
    olag, ocorr, _ = oracle.b_xcorr_peak_fft(pre[i], pre[j], ml)
    assert dec[wid, p]["lag"] == olag == delays[j] - delays[i]
    assert abs(dec[wid, p]["corr"] - ocorr) <= 1e-5 * abs(ocorr)
