"""GPU parity tests for mode A -- the reference's executed call surface (processor.go
loadIQData / preprocessSignal / timeDomainCorrelation / crossCorrelate, simple_corr.go
simpleCorrelate, fast_analyzer.go fastSNRCalculation) through the C ABI, against the CPU
oracle.  Bit-exact wherever the Go evaluation order is reproduced; 1e-12 where only an f64
sum is re-associated."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import tdoa_amd
    c = tdoa_amd.Context()
    yield c
    c.close()


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_load_iq_bit_exact(ctx, oracle):
    raw = np.arange(256, dtype=np.uint8).repeat(2)
    rng = np.random.default_rng(0)
    raw = np.concatenate([raw, rng.integers(0, 256, 100001 * 2, dtype=np.uint8)])
    assert np.array_equal(_bits(ctx.load_iq_u8(raw)), _bits(oracle.iq_u8_to_c64(raw)))


@pytest.mark.parametrize("n", [5, 1000, 2049, 30000])
def test_preprocess_weak_chain_bit_exact(ctx, oracle, n):
    raw = oracle.simulate_station("kx0u", max(n, 8), oracle.SEED_BASE)
    sig = oracle.iq_u8_to_c64(raw)[:n]
    want, oweak = oracle.preprocess(sig)
    got, weak = ctx.preprocess(sig)
    assert weak and oweak
    assert np.array_equal(_bits(got), _bits(want))


@pytest.mark.parametrize("n", [7, 4097, 30000])
def test_preprocess_standard_chain_bit_exact(ctx, oracle, n):
    rng = np.random.default_rng(n)
    sig = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    sig += np.complex64(0.3 - 0.2j)
    want, oweak = oracle.preprocess(sig)
    got, weak = ctx.preprocess(sig)
    assert not weak and not oweak
    assert np.array_equal(_bits(got), _bits(want))


def test_time_domain_equal_lengths_only_lag_zero(ctx, oracle):
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(50000) + 1j * rng.standard_normal(50000)).astype(np.complex64)
    b = (a + 0.5 * (rng.standard_normal(50000) + 1j * rng.standard_normal(50000))).astype(np.complex64)
    d, c = ctx.time_domain_correlation(a, b, 20000)
    od, oc = oracle.time_domain_correlation(a, b, 20000)
    assert d == od == 0                                   # processor.go:668-675
    assert abs(c - oc) <= 1e-12 * abs(oc)
    assert ctx.time_domain_correlation(a[:1000], a[:1000], 20000) == (0, 0.0)   # no block fits
    assert ctx.time_domain_correlation(a[:0], a, 20000) == (0, 0.0)


@pytest.mark.parametrize("nt,ns,shift,max_lag", [
    (5000, 9000, 321, 20000),      # 4000 lags: thread-per-lag kernel, sequential in-block sums
    (5000, 5040, 17, 20000),       # 40 lags: wave-per-(lag, block) kernel
    (12345, 20000, 4000, 3000),    # peak outside the searched range
    (3000, 30000, 19999, 20000),   # last searched lag
])
def test_time_domain_lag_search(ctx, oracle, nt, ns, shift, max_lag):
    rng = np.random.default_rng(nt)
    s = (rng.standard_normal(ns) + 1j * rng.standard_normal(ns)).astype(np.complex64)
    t = s[shift:shift + nt].copy() if shift + nt <= ns else s[:nt].copy()
    d, c = ctx.time_domain_correlation(t, s, max_lag)
    od, oc = oracle.time_domain_correlation(t, s, max_lag)
    assert d == od
    assert abs(c - oc) <= 1e-12 * abs(oc)
    if min(max_lag, ns - nt) >= 64:
        assert c == oc                                    # same summation order -> same bits
    d2, c2 = ctx.time_domain_correlation(s, t, max_lag)   # shorter input is the template
    assert (d2, c2) == (d, c)


def test_periodic_near_ties_resolve_to_lowest_lag(ctx, oracle):
    # 5-sample-period tone (400 kHz alias at 2 Msps): lags 0, 5, 10, ... are near ties
    n = 6000
    k = np.arange(n + 200)
    s = (0.3 * np.exp(2j * np.pi * 0.2 * k)).astype(np.complex64)
    d, c = ctx.time_domain_correlation(s[:n], s, 20000)
    od, oc = oracle.time_domain_correlation(s[:n], s, 20000)
    assert (d, c) == (od, oc)


def test_cross_correlate_reference_call_pattern(ctx, oracle):
    # ProcessTDOA: 3 stations, pairs i<j, reference-block and target-block signals of equal length
    blk = 12000
    caps = [oracle.simulate_station(nm, blk, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    data = [oracle.iq_u8_to_c64(c) for c in caps]
    refs = [oracle.extract_reference(d) for d in data]
    tgts = [oracle.extract_target(d) for d in data]
    for sigs in (refs, tgts):
        for i in range(3):
            for j in range(i + 1, 3):
                d, c = ctx.cross_correlate(sigs[i], sigs[j])
                od, oc = oracle.cross_correlate(sigs[i], sigs[j])
                assert d == od == 0
                assert abs(c - oc) <= 1e-9 * max(abs(oc), 1e-3)


def test_cross_correlate_sanity_flow(ctx, oracle):
    # correlation_sanity.go:35-58: crossCorrelate(x, x) must exceed 0.5 at delay 0
    raw = oracle.simulate_station("kx0u", 40000, oracle.SEED_BASE)
    data = oracle.iq_u8_to_c64(raw)
    for s in (oracle.extract_reference(data)[:20000], oracle.extract_target(data)[:20000]):
        d, c = ctx.cross_correlate(s, s)
        od, oc = oracle.cross_correlate(s, s)
        assert d == 0 and c > 0.5
        assert abs(c - oc) <= 1e-9 * abs(oc)
    assert ctx.cross_correlate(np.zeros(0, np.complex64), data[:10]) == (0, 0.0)


def test_cross_correlate_with_lag_search(ctx, oracle):
    a = oracle.iq_u8_to_c64(oracle.simulate_delayed_fm(9000, 0, 5, 1))
    b = oracle.iq_u8_to_c64(oracle.simulate_delayed_fm(14000, 777, 5, 2))
    d, c = ctx.cross_correlate(a, b)
    od, oc = oracle.cross_correlate(a, b)
    assert d == od
    assert abs(c - oc) <= 1e-9 * abs(oc)


def _simple_signal(oracle, n=10000, seed=1234):
    t = np.arange(n) / 100000.0
    sine = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    noise = np.array([0.1 * (oracle.rand_float64(seed, k) - 0.5) for k in range(n)]).astype(np.float32)
    return (sine + noise).astype(np.float32).astype(np.complex64)


def test_simple_correlate_acceptance_and_bits(ctx, oracle):
    sig = _simple_signal(oracle)
    d, c = ctx.simple_correlate(sig, sig)                          # simple_corr.go:33-36
    assert d == 0 and c > 0.8 and (d, c) == oracle.simple_correlate(sig, sig)
    shift = 100                                                    # :46-55
    delayed = np.zeros_like(sig)
    delayed[shift:] = sig[:-shift]
    a, b = sig[:len(sig) - shift], delayed[shift:]
    d, c = ctx.simple_correlate(a, b)
    assert c > 0.8 and -10 <= d <= 10 and (d, c) == oracle.simple_correlate(a, b)
    noise = np.array([complex(oracle.rand_float64(77, 2 * k) - 0.5, oracle.rand_float64(77, 2 * k + 1) - 0.5)
                      for k in range(len(sig))], dtype=np.complex64)
    d, c = ctx.simple_correlate(sig, noise)                        # :62-72
    assert abs(c) < 0.2 and (d, c) == oracle.simple_correlate(sig, noise)
    d, c = ctx.simple_correlate(sig[137:4137], sig)                # lag search, 1000 lags
    assert (d, c) == oracle.simple_correlate(sig[137:4137], sig) and d == 137


def test_fast_snr_matches_oracle(ctx, oracle):
    raw = oracle.simulate_station("kx0u", 40000, oracle.SEED_BASE, tx_power=200000.0)
    ref = np.concatenate([raw[:2 * 32768], raw[4 * 40000:4 * 40000 + 2 * 32768]])
    for samples, total in ((ref, 65536), (raw[2 * 40000:2 * 40000 + 2 * 32768], 32768), (raw[:2 * 3000], 3000)):
        assert ctx.fast_snr(samples, total) == oracle.fast_snr(samples, total)
    assert ctx.fast_snr(np.array([128, 128], np.uint8), 1) == -20.0


def test_fast_analyzer_capture_matches_oracle(ctx, oracle):
    # fast_analyzer.go main: "REF,<snr>,<power>,<clip>,<overload>" / "TGT,..." (:44-50)
    raw = oracle.simulate_station("n3pay", 40000, oracle.SEED_BASE + 1, tx_power=60000.0)
    ref, tgt = ctx.fast_analyze_capture(raw)
    rc, oref, otgt = oracle.fast_analyze_capture(raw)
    assert rc == 0
    for g, o in ((ref, oref), (tgt, otgt)):
        assert g.total_samples == o.total_samples
        assert (g.i_avg, g.q_avg, g.i_std, g.q_std) == (o.i_avg, o.q_avg, o.i_std, o.q_std)
        assert g.power_level == o.power_level and g.snr_estimate == o.snr_estimate
        assert (g.has_clipping, g.has_overload) == (o.has_clipping, o.has_overload)
        line = "%.1f,%.1f,%s,%s" % (g.snr_estimate, g.power_level, bool(g.has_clipping), bool(g.has_overload))
        assert line == "%.1f,%.1f,%s,%s" % (o.snr_estimate, o.power_level, bool(o.has_clipping), bool(o.has_overload))
    flat = np.full(2 * 9000, 128, dtype=np.uint8)
    fa = ctx.fast_analyze(flat, 9000)
    assert fa.power_level == -100.0 and fa.has_overload == 1 and fa.has_clipping == 0


def test_reference_call_size_two_million_samples(oracle):
    """processor.go:772: the reference correlates the first 2 000 000 samples.  Same size here: the preprocessed
    signal bit for bit (2 M sequential f32 DC sum, 1123-tap filter chain), delay identical, corr to 1e-12."""
    import tdoa_amd
    n = 2_000_000
    rng = np.random.default_rng(77)
    raw = [rng.integers(118, 138, size=2 * n, dtype=np.uint8) for _ in range(2)]
    with tdoa_amd.Context() as c:
        sig = [c.load_iq_u8(r) for r in raw]
        assert np.array_equal(sig[0].view(np.uint32), oracle.iq_u8_to_c64(raw[0]).view(np.uint32))
        got, weak = c.preprocess(sig[0])
        want, oweak = oracle.preprocess(sig[0])
        assert weak == oweak
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        d, corr = c.cross_correlate(sig[0], sig[1])
        od, ocorr = oracle.cross_correlate(sig[0], sig[1])
        assert d == od == 0
        assert abs(corr - ocorr) <= 1e-12 * abs(ocorr)


def test_cross_correlate_batch_equals_pair_calls(ctx, oracle):
    """tdoa_cross_correlate_batch_c64: every signal preprocessed once (the reference re-does it per pair,
    processor.go:629-630), every pair i < j correlated -- bit-identical to the per-pair calls and to the oracle's delay"""
    raw = [oracle.simulate_station(nm, 20000, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    sigs = [oracle.extract_target(oracle.iq_u8_to_c64(r)) for r in raw]
    sigs.append(sigs[0][:12000])                                  # a shorter one: real lag searches against the others
    got = ctx.cross_correlate_batch(sigs)
    assert len(got) == 6
    p = 0
    for i in range(4):
        for j in range(i + 1, 4):
            d, c = ctx.cross_correlate(sigs[i], sigs[j])
            assert got[p] == (d, c), (i, j)
            od, oc = oracle.cross_correlate(sigs[i], sigs[j])
            assert d == od and abs(c - oc) <= 1e-9 * max(abs(oc), 1e-300)
            p += 1
    # empty members give (0, 0.0) for their pairs, like processor.go:622-625
    got = ctx.cross_correlate_batch([sigs[0], np.zeros(0, np.complex64), sigs[1]])
    assert got[0] == (0, 0.0) and got[2] == (0, 0.0) and got[1] == ctx.cross_correlate(sigs[0], sigs[1])
