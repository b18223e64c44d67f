"""Pins the CPU oracle against every known answer the reference holds for the
hot path.  The reference has no numeric golden
vectors for correlation outputs; these are the acceptance thresholds and
known-answer constants it does contain.  CPU only."""
import math

import numpy as np
import pytest


# PROJECT_NOTES.md:25-27 -- 3-D baselines between the three collectors
def test_baselines_project_notes(oracle):
    st = oracle.STATIONS
    ecef = {k: oracle.latlon_to_ecef(*v) for k, v in st.items()}
    d = lambda a, b: oracle.distance3d(ecef[a], ecef[b]) / 1000.0
    assert round(d("kx0u", "n3pay"), 2) == 12.29
    assert round(d("kx0u", "kf0mtl"), 2) == 17.02
    assert round(d("n3pay", "kf0mtl"), 2) == 10.02
    # PROJECT_NOTES.md:29-32 -- max |TDOA| ~57 us = 17 km / c  (~114 samples at 2 Msps)
    tmax = d("kx0u", "kf0mtl") * 1000.0 / 299792458.0
    assert abs(tmax * 1e6 - 57.0) < 0.5
    assert int(tmax * 2e6) in (113, 114)


def test_ecef_roundtrip(oracle):
    for lat, lon, el in oracle.STATIONS.values():
        x, y, z = oracle.latlon_to_ecef(lat, lon, el)
        lle = oracle.ecef_to_latlon(x, y, z)
        assert abs(lle[0] - lat) < 1e-9 and abs(lle[1] - lon) < 1e-9 and abs(lle[2] - el) < 1e-4


# docs/usage.md:125-128 -- capture size = duration * 2 Msps * 2 B
def test_file_size_arithmetic():
    assert 30 * 2_000_000 * 2 == 120_000_000
    assert 100 * 2_000_000 * 2 == 400_000_000


def _simple_corr_signal(oracle, n=10000, seed=1234):
    # simple_corr.go:17-27: 0.5*sin(2*pi*1000*t) at 100 kHz + 0.1*(U-0.5), imag 0
    i = np.arange(n)
    t = i / 100000.0
    sine = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    noise = np.array([0.1 * (oracle.rand_float64(seed, k) - 0.5) for k in range(n)]).astype(np.float32)
    return (sine + noise).astype(np.float32).astype(np.complex64)


# simple_corr.go:31-76 -- the reference's three executable acceptance checks
def test_simple_corr_acceptance(oracle):
    sig = _simple_corr_signal(oracle)
    delay, corr = oracle.simple_correlate(sig, sig)          # :33-36
    assert delay == 0 and corr > 0.8
    assert abs(corr - 1.0) < 1e-3
    shift = 100                                               # :46-55
    delayed = np.zeros_like(sig)
    delayed[shift:] = sig[:-shift]
    delay, corr = oracle.simple_correlate(sig[:len(sig) - shift], delayed[shift:])
    assert corr > 0.8 and -10 <= delay <= 10
    n = len(sig)                                              # :62-72
    noise = np.array([complex(oracle.rand_float64(77, 2 * k) - 0.5, oracle.rand_float64(77, 2 * k + 1) - 0.5)
                      for k in range(n)], dtype=np.complex64)
    delay, corr = oracle.simple_correlate(sig, noise)
    assert abs(corr) < 0.2


def test_simple_correlate_finds_shift_when_lengths_differ(oracle):
    sig = _simple_corr_signal(oracle, n=6000)
    # template = sig[137 : 137+4000] appears in the longer signal at delay 137
    delay, corr = oracle.simple_correlate(sig[137:4137], sig)
    assert delay == 137 and corr > 0.99


# correlation_sanity.go:35-58 -- crossCorrelate(x, x) on the first 100000 samples of
# the reference and target blocks of one capture must exceed 0.5 (it is ~sqrt(99000)
# because of the coherent-gain factor, processor.go:719-720)
def test_correlation_sanity_flow(oracle):
    raw = oracle.simulate_station("kx0u", 40000, oracle.SEED_BASE)
    data = oracle.iq_u8_to_c64(raw)
    ref = oracle.extract_reference(data)[:100000]
    tgt = oracle.extract_target(data)[:100000]
    assert ref.size == 80000 and tgt.size == 40000
    for s in (ref[:20000], tgt[:20000]):
        delay, corr = oracle.cross_correlate(s, s)
        assert delay == 0
        assert corr > 0.5
        nblocks = math.ceil((s.size - 1000) / 1000)
        assert abs(corr - math.sqrt(nblocks * 1000)) / corr < 0.05


# SURVEY finding 3 / processor.go:668-675,686 -- equal lengths => only delay 0 is evaluated
def test_equal_length_only_lag_zero(oracle):
    rng = np.random.default_rng(5)
    a = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
    b = np.roll(a, 7)
    d, c = oracle.time_domain_correlation(a, b, 20000)
    assert d == 0
    lags = oracle.time_domain_all_lags(a, b, 20000)
    assert lags.size == 1 and lags[0] == c


def test_time_domain_finds_delay_when_lengths_differ(oracle):
    rng = np.random.default_rng(6)
    s = (rng.standard_normal(9000) + 1j * rng.standard_normal(9000)).astype(np.complex64)
    t = s[321:321 + 5000].copy()
    d, c = oracle.time_domain_correlation(t, s, 20000)
    assert d == 321
    # template longer than 1000 but the last 1..1000 samples are ignored (:691)
    nb = math.ceil((5000 - 1000) / 1000)
    p = np.mean(np.abs(t[:nb * 1000]) ** 2)
    assert abs(c - p * math.sqrt(nb * 1000)) / c < 1e-5
    # swapping arguments keeps the shorter one as template (:650-655)
    d2, c2 = oracle.time_domain_correlation(s, t, 20000)
    assert (d2, c2) == (d, c)


def test_short_template_returns_zero(oracle):
    a = np.ones(1000, dtype=np.complex64)
    d, c = oracle.time_domain_correlation(a, a, 20000)   # no block fits: Lt-1000 == 0
    assert (d, c) == (0, 0.0)
    d, c = oracle.cross_correlate(np.zeros(0, np.complex64), a)  # :622-625
    assert (d, c) == (0, 0.0)


# processor.go:400-406 window sizes implied by the hard-coded cut-offs
def test_filter_window_sizes(oracle):
    assert oracle.cutoff_window(57.5) == 1000
    assert oracle.cutoff_window(62.5) == 1000
    assert oracle.cutoff_window(975000.0) == 3
    assert oracle.cutoff_window(100.0) == 1000
    assert oracle.cutoff_window(40000.0) == 25
    assert oracle.cutoff_window(500.0) == 1000
    assert oracle.cutoff_window(50000.0) == 20


def test_lowpass_matches_definition(oracle):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(300) + 1j * rng.standard_normal(300)).astype(np.complex64)
    y = oracle.lowpass(x, 20)          # half window 10 -> 21 taps, edge-truncated
    for i in (0, 3, 150, 299):
        j0, j1 = max(0, i - 10), min(299, i + 10)
        sr = np.float32(0)
        si = np.float32(0)
        for j in range(j0, j1 + 1):
            sr = np.float32(sr + x[j].real)
            si = np.float32(si + x[j].imag)
        cnt = np.float32(j1 - j0 + 1)
        assert y[i].real == np.float32(sr / cnt) and y[i].imag == np.float32(si / cnt)
    assert np.array_equal(oracle.lowpass(x, 1), x)     # windowSize <= 1 returns the input


def test_u8_conversion_is_true_division(oracle):
    raw = np.arange(256, dtype=np.uint8).repeat(2)
    c = oracle.iq_u8_to_c64(raw)
    want = ((np.arange(256, dtype=np.float32) - np.float32(127.5)) / np.float32(127.5)).astype(np.float32)
    assert np.array_equal(c.real, want) and np.array_equal(c.imag, want)
    # a reciprocal multiply differs for some codes -- the reason for the LUT on the GPU
    recip = (np.arange(256, dtype=np.float32) - np.float32(127.5)) * np.float32(1.0 / 127.5)
    assert np.any(recip != want)


def test_block_extraction(oracle):
    data = (np.arange(10) + 0j).astype(np.complex64)    # 10 samples -> block 3
    ref = oracle.extract_reference(data)
    tgt = oracle.extract_target(data)
    assert np.array_equal(ref.real, [0, 1, 2, 6, 7, 8]) and np.array_equal(tgt.real, [3, 4, 5])
    small = (np.arange(2) + 0j).astype(np.complex64)    # block size 0 -> returned unchanged
    assert np.array_equal(oracle.extract_reference(small), small)
    assert np.array_equal(oracle.extract_target(small), small)


def test_preprocess_gate_and_unit_power(oracle):
    raw = oracle.simulate_station("n3pay", 20000, oracle.SEED_BASE + 1)
    data = oracle.iq_u8_to_c64(raw)
    ref = oracle.extract_reference(data)[:20000]
    p0 = oracle.signal_power(ref)
    assert p0 < 0.001                  # simulator amplitudes land in the weak chain
    out, weak = oracle.preprocess(ref)
    assert weak
    assert abs(oracle.signal_power(out) - 1.0) < 1e-4
    strong = (ref * np.complex64(50)).astype(np.complex64)
    out2, weak2 = oracle.preprocess(strong)
    assert not weak2 and abs(oracle.signal_power(out2) - 1.0) < 1e-4


# fast_analyzer.go:146-151, :222-226 fallbacks
def test_fast_analyzer_fallbacks(oracle):
    flat = np.full(2 * 9000, 128, dtype=np.uint8)
    fa = oracle.fast_analyze(flat, 9000)
    assert fa.power_level == -100.0
    # a constant capture is a pure (windowed) DC line: huge "SNR", not the fallback
    assert fa.snr_estimate > 100.0
    assert fa.has_overload == 1 and fa.has_clipping == 0
    # one sample: the Hann window is 0/0 = NaN, no bin qualifies, fallback -20 dB (:222-226)
    assert oracle.fast_snr(np.array([128, 128], np.uint8), 1) == -20.0
    raw = oracle.simulate_station("kx0u", 40000, oracle.SEED_BASE, tx_power=200000.0)
    rc, ref, tgt = oracle.fast_analyze_capture(raw)
    assert rc == 0 and ref.total_samples == 65536 and tgt.total_samples == 32768
    assert tgt.snr_estimate > ref.snr_estimate        # strong target tone vs 0.01-amplitude reference
    assert oracle.fast_analyze_capture(np.zeros(4, np.uint8))[0] == -1


def test_fast_dft_matches_numpy(oracle):
    from oracle import pyoracle
    import ctypes as C
    rng = np.random.default_rng(3)
    x = rng.standard_normal(256) + 1j * rng.standard_normal(256)
    xin = np.ascontiguousarray(x.astype(np.complex128)).view(np.float64)
    out = np.empty(512)
    pyoracle.lib().o_fast_dft(pyoracle._d(xin), C.c_int(256), pyoracle._d(out))
    got = out.view(np.complex128)
    assert np.allclose(got, np.fft.fft(x), atol=1e-9)
    y = oracle.simple_dft(x.astype(np.complex64)[:64])
    assert np.allclose(y, np.fft.fft(x.astype(np.complex64)[:64]), atol=2e-4)


def test_dead_frequency_domain_path(oracle):
    a = np.ones(2048, dtype=np.complex64)
    rc, d, c = oracle.frequency_domain_correlation(a, a, 20000)
    assert rc == -1                                    # would index correlation[fftSize], processor.go:605-606
    assert oracle.next_pow2(1024 + 20000) == 32768     # processor.go:563


# processor.go:932-1020 solver: 2x2 damped Newton in ECEF X,Y with Z frozen (:1004), so it
# can only move inside the plane Z = Z(centroid); pin the cases that plane contains.
def test_solve_tdoa(oracle):
    st = [oracle.STATIONS[k] for k in oracle.COLLECTORS]
    c = np.mean(np.array(st), axis=0)                          # centroid start (:950-955)
    ce = oracle.latlon_to_ecef(*c)
    r = [oracle.distance3d(oracle.latlon_to_ecef(*s), ce) for s in st]
    rc, lle, iters = oracle.solve_tdoa(st, [r[1] - r[0], r[2] - r[0]])
    assert rc == 0 and iters == 0                              # residuals < 1 m at the start (:970)
    assert np.allclose(lle, c, atol=1e-6)
    tx = (c[0], c[1] - 0.05, c[2])                             # due west: (almost) the same ECEF Z
    txe = oracle.latlon_to_ecef(*tx)
    r = [oracle.distance3d(oracle.latlon_to_ecef(*s), txe) for s in st]
    rc, lle, iters = oracle.solve_tdoa(st, [r[1] - r[0], r[2] - r[0], r[2] - r[1]])
    assert rc == 0 and iters == 10                             # 0.5 damping: never reaches 1 m in 10 steps
    assert abs(lle[0] - tx[0]) < 1e-4 and abs(lle[1] - tx[1]) < 1e-3
    # the reference's own outcome on its simulator files: all delays 0 (processor.go:868)
    rc0, lle0, it0 = oracle.solve_tdoa(st, [0.0, 0.0, 0.0])
    assert rc0 == 0 and 41.1 < lle0[0] < 41.4 and -96.2 < lle0[1] < -95.9


def test_simulator_shape_and_determinism(oracle):
    a = oracle.simulate_station("kx0u", 5000, 42)
    b = oracle.simulate_station("kx0u", 5000, 42)
    c = oracle.simulate_station("kx0u", 5000, 43)
    assert a.size == 30000 and np.array_equal(a, b) and not np.array_equal(a, c)
    # tones are tiny: amplitude 0.01 -> ~1.3 LSB about 127.5 (simulator.go:126-128)
    assert 120 <= a.min() and a.max() <= 135
    w = oracle.simulate_weak_station("kf0mtl", 5000, 44)
    assert w.size == 30000 and abs(float(w.mean()) - 127.0) < 1.0


def test_k1_codes_agree_with_the_elf_discriminator_definition(oracle):
    """SURVEY section 8, row K1 (read off the prebuilt binary): x = ((I-127.5)/127.5, (Q-127.5)/127.5),
    p = x_i conj(x_{i-1}), y_i = atan2(Im p, Re p) unless |p|^2 <= 1e-10, y_0 := y_1.  The phase code is that
    angle in units of pi/2^23 AS A REAL NUMBER in (-pi, +pi] (no modulo 2 pi: an exactly reversed sample is +pi,
    a near reversal keeps the sign of Im p), up to the two roundings of the angle codes it is the difference of."""
    from oracle import float_pipeline as fp
    unit = np.pi / 2.0 ** 23
    rng = np.random.default_rng(2024)
    iq = rng.integers(0, 256, size=2 * 200000, dtype=np.uint8)
    iq[:12] = [0, 0, 255, 255, 0, 255, 255, 0, 127, 128, 128, 127]       # corners and the centre
    # exactly reversed samples: equal magnitude, and collinear with different magnitudes (Im p = 0, Re p < 0)
    iq[20:28] = [129, 129, 126, 126, 128, 126, 127, 129]                    # (3,3)->(-3,-3), (1,-3)->(-1,3)
    iq[28:36] = [128, 128, 126, 126, 129, 129, 127, 127]                    # (1,1)->(-3,-3), (3,3)->(-1,-1)
    # the nearest a pair of NON-collinear samples gets to a reversal is 1/(|x_i||x_{i-1}|) rad (full scale: 7.7e-6 rad =
    # 20 code steps): the sign of Im p decides the side, and the code follows it without a special case
    # (sample 21: Im p < 0 -> -pi side; sample 23: Im p > 0 -> +pi side)
    iq[40:48] = [254, 253, 0, 1, 234, 229, 0, 6]
    code = oracle.b_discriminate(iq).astype(np.float64)
    assert code.min() > -2.0 ** 23 and code.max() <= 2.0 ** 23
    y = fp.discriminate(iq)                                                # exact-product float64 statement
    diff = code * unit - y                                                 # NOT wrapped
    assert np.abs(diff).max() <= 1.01 * unit                               # two half-step roundings
    assert code[0] == code[1]
    assert code[11] == 2 ** 23 and code[13] == 2 ** 23 and code[15] == 2 ** 23 and code[17] == 2 ** 23
    assert -2 ** 23 < code[21] < -2 ** 23 + 4096 and -np.pi < y[21] < -3.1415
    assert 2 ** 23 - 4096 < code[23] < 2 ** 23 and 3.1415 < y[23] < np.pi
    # a small-amplitude capture (simulator.go: I/Q of +-1, +-3 LSB) is full of exact reversals
    sim = oracle.simulate_station("kx0u", 20000, 7)
    cs, ys = oracle.b_discriminate(sim).astype(np.float64), fp.discriminate(sim)
    assert (ys == np.pi).sum() > 100 and (ys == -np.pi).sum() == 0
    assert ((cs == 2.0 ** 23) == (ys == np.pi)).all()
    assert np.abs(cs * unit - ys).max() <= 1.01 * unit


def test_k1_exact_reversal_is_the_only_half_turn(oracle):
    """a_i - a_{i-1} = -+2^23 must mean 'exactly reversed' (then Im p = 0 and atan2 gives +pi): over ALL pairs of the
    65536 byte pairs, the directions whose angle codes are exactly opposite are collinear and opposite."""
    code = np.array([[oracle.b_angle_code(2 * i - 255, 2 * q - 255) for q in range(256)] for i in range(256)], dtype=np.int64)
    ii, qq = np.meshgrid(2 * np.arange(256) - 255, 2 * np.arange(256) - 255, indexing="ij")
    flat, fi, fq = code.ravel(), ii.ravel(), qq.ravel()
    order = np.argsort(flat, kind="stable")
    sc = flat[order]
    # partner codes: c + 2^23 (for c <= 0) -- every sample whose code equals that must be collinear and opposite
    for lo in range(0, sc.size, 8192):
        blk = order[lo:lo + 8192]
        c = flat[blk]
        tgt = np.where(c <= 0, c + 2 ** 23, c - 2 ** 23)
        left, right = np.searchsorted(sc, tgt, "left"), np.searchsorted(sc, tgt, "right")
        assert (right > left).all()                                          # (-I, -Q) always exists
        for k in np.nonzero(right - left > 0)[0]:
            other = order[left[k]:right[k]]
            cross = fq[blk[k]] * fi[other] - fi[blk[k]] * fq[other]
            dot = fi[blk[k]] * fi[other] + fq[blk[k]] * fq[other]
            assert (cross == 0).all() and (dot < 0).all()


def test_k1_angle_table_properties(oracle):
    """every one of the 65536 byte pairs: the correctly rounded atan2 in units of pi/2^23,
    collinear samples share a code, opposite samples differ by exactly half a turn"""
    from math import gcd
    code = np.array([[oracle.b_angle_code(2 * i - 255, 2 * q - 255) for q in range(256)] for i in range(256)])
    bi, bq = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    want = np.arctan2(2.0 * bq - 255.0, 2.0 * bi - 255.0) * (2.0 ** 23 / np.pi)
    assert np.abs(code - want).max() <= 0.5 + 1e-6
    assert np.abs(code).max() < 2 ** 23
    opp = code[::-1, ::-1]                                                  # (b_I, b_Q) -> (255 - b_I, 255 - b_Q)
    assert (np.abs(code - opp) == 2 ** 23).all()
    for (i, q, k) in [(1, 1, 3), (1, 3, 5), (3, 5, 7), (7, 1, 9), (5, 11, 21), (1, 1, 255)]:
        for si in (1, -1):
            for sq in (1, -1):
                assert oracle.b_angle_code(si * i, sq * q) == oracle.b_angle_code(si * i * k, sq * q * k)
    assert gcd(3, 9) == 3 and oracle.b_angle_code(1, 1) == 2 ** 21 and oracle.b_angle_code(-1, 1) == 3 * 2 ** 21
    assert oracle.b_angle_code(-1, -1) == -3 * 2 ** 21 and oracle.b_angle_code(1, -1) == -2 ** 21
    # the first-octant table the product keeps: 8256 directions, strictly inside (0, 2^21]
    tab = np.array([oracle.b_octant_code(2 * mn + 1, 2 * mx + 1) for mx in range(128) for mn in range(mx + 1)])
    assert tab.size == 8256 and tab.min() > 0 and tab.max() == 2 ** 21
