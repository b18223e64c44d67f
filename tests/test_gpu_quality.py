"""GPU parity of the capture-quality statistics through the C ABI:
tdoa_window_quality_u8 / tdoa_window_quality_all / tdoa_fast_analyze_u8 against the oracle's
restatement of fastAnalyzeSamples (fast_analyzer.go:113-161) and of validateDataFile's block
power (collector.go:219-224).  Integer sums: every float64 field must match bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("i_avg", "q_avg", "i_std", "q_std", "power_level")


def _check(got, iq, oracle):
    n = iq.size // 2
    want = oracle.fast_analyze(iq, n)
    for f in FIELDS:
        assert np.float64(got[f]).tobytes() == np.float64(getattr(want, f)).tobytes(), f
    assert got["n_samples"] == n
    assert bool(got["has_clipping"]) == bool(want.has_clipping)
    assert bool(got["has_overload"]) == bool(want.has_overload)
    assert (got["i_min"], got["i_max"]) == (int(iq[0::2].min()), int(iq[0::2].max()))
    assert (got["q_min"], got["q_max"]) == (int(iq[1::2].min()), int(iq[1::2].max()))
    assert np.float64(got["mean_power"]).tobytes() == np.float64(oracle.block_power(iq)).tobytes()


@pytest.mark.parametrize("n", [1, 7, 8, 9, 1000, 16384, 16385, 65536, 100003])
def test_window_quality_u8_bit_exact(oracle, n):
    import tdoa_amd
    rng = np.random.default_rng(n)
    iq = rng.integers(100, 156, size=2 * n, dtype=np.uint8)
    with tdoa_amd.Context() as c:
        got = c.window_quality(iq)
        _check(got, iq, oracle)


def test_flags_and_extremes(oracle):
    import tdoa_amd
    with tdoa_amd.Context() as c:
        flat = np.full(2 * 5000, 128, np.uint8)                     # sigma 0: overload flag, power floor
        got = c.window_quality(flat)
        _check(got, flat, oracle)
        assert got["has_overload"] == 1 and got["has_clipping"] == 0 and got["power_level"] == -100.0
        clip = oracle.simulate_station("kx0u", 3000, oracle.SEED_BASE)
        clip[4001] = 255                                            # one clipped Q byte
        got = c.window_quality(clip)
        _check(got, clip, oracle)
        assert got["has_clipping"] == 1 and got["q_max"] == 255
        clip[4001] = 128
        clip[10] = 0                                                # one clipped I byte at the low end
        got = c.window_quality(clip)
        assert got["has_clipping"] == 1 and got["i_min"] == 0
        # odd byte offsets inside a capture (window starts are only 2-byte aligned)
        for lo in (2, 6, 14):
            _check(c.window_quality(clip[lo:lo + 2 * 2001]), clip[lo:lo + 2 * 2001], oracle)


def test_window_quality_all_and_sharding(oracle):
    import tdoa_amd
    block, wl = 30000, 10000
    caps = [oracle.simulate_station(nm, block, oracle.SEED_BASE + i, tx_power=200000.0)
            for i, nm in enumerate(oracle.COLLECTORS)]
    with tdoa_amd.Context(window_len=wl, max_lag=300) as c:
        for s, cap in enumerate(caps):
            c.capture_upload(s, cap)
        peaks = c.process()
        q = c.window_quality_all()
        assert q.shape == (9, 3)
        for wid in range(9):
            off = (wid // 3) * block + (wid % 3) * wl
            for s, cap in enumerate(caps):
                _check(q[wid, s], cap[2 * off:2 * (off + wl)], oracle)
        # validateDataFile's check: the two reference blocks have consistent power, within 2x (collector.go:231-237)
        ratio = q[6, 0]["mean_power"] / q[0, 0]["mean_power"]
        assert 0.5 <= ratio <= 2.0
        for r in range(2):
            part = c.window_quality_all(rank=r, world=2)
            for wid in range(9):
                if wid % 2 == r:
                    assert part[wid].tobytes() == q[wid].tobytes()
                else:
                    assert not part[wid]["n_samples"].any()
        assert np.array_equal(c.process(), peaks)          # the captured correlation graph is still valid


def test_fast_analyze_uses_device_statistics(oracle):
    import tdoa_amd
    raw = oracle.simulate_station("n3pay", 70000, oracle.SEED_BASE + 1)
    with tdoa_amd.Context() as c:
        got = c.fast_analyze(raw[:2 * 65536], 65536)
    want = oracle.fast_analyze(raw[:2 * 65536], 65536)
    for f in FIELDS + ("snr_estimate",):
        assert np.float64(getattr(got, f)).tobytes() == np.float64(getattr(want, f)).tobytes(), f
    assert (got.has_clipping, got.has_overload) == (want.has_clipping, want.has_overload)
