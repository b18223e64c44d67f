"""CPU tests of the sub-sample refinement oracle.
The reference has no interpolation code; the anchors are PROJECT_NOTES.md:29-32 (max |TDOA| about
57 us, i.e. 114 samples at 2 Msps) and the 120-sample bound of the prebuilt binary's second search."""
import numpy as np


def test_parabola_vertex_known_values(oracle):
    v = oracle.b_parabola_vertex
    assert v(1.0, 2.0, 1.0) == 0.0                                # symmetric
    assert abs(v(1.0, 2.0, 1.5) - (0.5 * (1.0 - 1.5) / (1.0 - 4.0 + 1.5))) < 1e-15
    assert v(2.0, 2.0, 1.0) == -0.5                               # plateau on the left: clamped
    assert v(1.0, 2.0, 2.0) == 0.5
    assert v(3.0, 2.0, 3.0) == 0.0                                # convex: no vertex
    assert v(1.0, 1.0, 1.0) == 0.0                                # flat
    assert v(0.0, 0.0, 0.0) == 0.0
    assert v(float("nan"), 1.0, 0.5) == 0.0                       # NaN never moves the peak


def test_parabola_recovers_sampled_vertex(oracle):
    for x0 in (-0.49, -0.2, 0.0, 0.123, 0.5):
        f = lambda x: 7.0 - 3.0 * (x - x0) ** 2
        assert abs(oracle.b_parabola_vertex(f(-1.0), f(0.0), f(1.0)) - x0) < 1e-12


def test_refine_peak_sign_gate_and_neighbours(oracle):
    rng = np.random.default_rng(5)
    t = rng.standard_normal(4000).astype(np.float32)
    s = np.zeros(4100, np.float32)
    s[37:4037] = -t                                               # anti-correlated at lag 37
    c = oracle.b_xcorr_all_lags(t, s, 100)
    lag, corr = oracle.b_pick_peak(c, 100)
    assert lag == 37 and corr < 0
    f = oracle.b_refine_peak(t, s, lag, 50.0)
    want_y = -c[[36 + 99, 37 + 99, 38 + 99]]                      # y = sign(c[lag]) * c
    assert np.allclose(f["y"], want_y, rtol=0, atol=1e-12)
    assert f["y"][1] > 0 and abs(f["frac"]) <= 0.5
    assert abs(f["delay"] - (37 + f["frac"])) < 1e-15
    assert f["plausible"]
    assert not oracle.b_refine_peak(t, s, lag, 36.0)["plausible"]
    # neighbours beyond the searched range are still the linear correlation values
    f99 = oracle.b_refine_peak(t, s, 99, 1e9)
    full = oracle.b_xcorr_all_lags(t, s, 101)
    sg = 1.0 if full[99 + 100] >= 0 else -1.0
    assert np.allclose(f99["y"], sg * full[[98 + 100, 99 + 100, 100 + 100]], rtol=0, atol=1e-12)


def test_refine_peak_all_zero(oracle):
    z = np.zeros(100, np.float32)
    f = oracle.b_refine_peak(z, z, 0, 120.0)
    assert f["delay"] == 0.0 and f["frac"] == 0.0 and not f["y"].any() and f["plausible"]


def test_fractional_delay_is_recovered(oracle):
    """band-limited signal delayed by a fraction of a sample: the parabola lands within 0.1 sample"""
    n = 8192
    rng = np.random.default_rng(9)
    spec = np.fft.rfft(rng.standard_normal(n))
    k = np.arange(spec.size)
    spec[k > n // 8] = 0.0                                        # band limit: 1/4 of Nyquist
    for true in (12.3, -7.75, 0.4):
        x = np.fft.irfft(spec, n)
        y = np.fft.irfft(spec * np.exp(-2j * np.pi * k * true / n), n)
        t = x[1000:7000].astype(np.float32)
        s = y[1000:7000].astype(np.float32)
        lag, _ = oracle.b_xcorr_peak(t, s, 64)
        f = oracle.b_refine_peak(t, s, lag, 120.0)
        assert abs(f["delay"] - true) < 0.1, (true, f)
