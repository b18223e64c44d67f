"""GPU: the BASELINE configurations that round 1 left unexercised ("configs not exercised").

cfg3 -- weak_signal_simulator.go captures (weak reference blocks, strong target block): oracle bytes through mode A
        (bit-exact) and mode B (vs the ob_* oracle and the float64 atan2 pipeline) including one full 10 s window
        (L = 20 000 000, N = 2^25), and the device-side generator bench.py uses for the 1024-pair-window stream;
cfg5 -- 16 collectors / 120 pairs in miniature, every pair against the oracle;
cfg4 -- sharded ingest: a rank that uploads only the windows it owns gets the same peaks as a full upload."""
import numpy as np
import pytest

from oracle import float_pipeline as fp

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_cfg3_weak_capture_mode_a_bit_exact(oracle):
    """the reference's executed chain (power gate -> weak-signal filter chain -> time-domain correlation) on
    weak_signal_simulator.go bytes: preprocessed samples bit for bit, delay exact, corr to 1e-12"""
    import tdoa_amd
    blk = 30000
    caps = [oracle.simulate_weak_station(nm, blk, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
    with tdoa_amd.Context() as c:
        sigs = [c.load_iq_u8(cp) for cp in caps]
        for cp, sg in zip(caps, sigs):
            assert np.array_equal(_bits(sg), _bits(oracle.iq_u8_to_c64(cp)))
        refs = [oracle.extract_reference(s) for s in sigs]
        tgts = [oracle.extract_target(s) for s in sigs]
        for kind in (refs, tgts):
            for s in kind:
                want, oweak = oracle.preprocess(s)
                got, weak = c.preprocess(s)
                assert weak == oweak and np.array_equal(_bits(got), _bits(want))
            for (i, j) in ((0, 1), (0, 2), (1, 2)):
                d, corr = c.cross_correlate(kind[i], kind[j])
                od, ocorr = oracle.cross_correlate(kind[i], kind[j])
                assert d == od == 0                                        # equal lengths: processor.go:668-675
                assert abs(corr - ocorr) <= 1e-12 * max(abs(ocorr), 1e-300)
        # unequal lengths: a real lag search through the weak chain
        d, corr = c.cross_correlate(tgts[0][:20000], tgts[1])
        od, ocorr = oracle.cross_correlate(tgts[0][:20000], tgts[1])
        assert d == od and abs(corr - ocorr) <= 1e-9 * abs(ocorr)


def test_cfg3_weak_capture_mode_b_full_window(oracle, capsys):
    """one full cfg3 unit: a 10 s window (20 000 000 samples, N = 5 x 2^22 = 2 x 4096 x 2560, two-sweep column plan) of the strong target
    block of two weak-simulator stations, against the ob_* oracle (f64 FFT evaluation) and the float64 atan2 pipeline;
    and a reference-block window, whose bytes are constant (amplitude 1.4e-4 -> 127): zero phase, (0, 0.0)"""
    import tdoa_amd
    L, ml = 20_000_000, 20000
    caps = [oracle.simulate_weak_station(nm, L, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS[:2])]
    tgt = [cp[2 * L:4 * L] for cp in caps]
    ref = [cp[:2 * L] for cp in caps]
    with tdoa_amd.Context(max_lag=ml, window_len=L) as c:
        lag, corr = c.fm_xcorr(tgt[0], tgt[1], ml)
        n, n1, n2 = c.plan_info()
        assert (n, n1, n2) == (5 << 22, 4096, 2560)
        assert c.fm_xcorr(ref[0], ref[1], ml) == (0, 0.0)
    pre = [oracle.b_preprocess(x)[0] for x in tgt]
    olag, ocorr, _ = oracle.b_xcorr_peak_fft(pre[0], pre[1], ml)
    assert lag == olag and abs(corr - ocorr) <= REL_TOL * abs(ocorr)
    flag, fcorr, _ = fp.xcorr_peak_u8(tgt[0], tgt[1], ml)
    dev = abs(corr - fcorr) / abs(fcorr)
    assert lag == flag and dev < 1e-5            # north_star's tolerance, against the float64 atan2 definition
    with capsys.disabled():
        print("\n  cfg3 window (L = 2e7): lag %d corr %.6f, vs ob_* %.2e, vs float64 atan2 pipeline %.2e"
              % (lag, corr, abs(corr - ocorr) / abs(ocorr), dev))


def test_cfg3_device_generator_matches_the_model(oracle):
    """k_synth_weak_block follows the same model and the same counter-based generator as the oracle's restatement of
    weak_signal_simulator.go; device and host libm differ in the last bit of sin/cos/log, which moves a sample across
    a quantisation boundary now and then -- never by more than one code"""
    import tdoa_amd
    blk = 200_000
    lle = oracle.STATIONS["n3pay"]
    want = oracle.simulate_weak_station("n3pay", blk, 99, tgt_power=1000.0)
    strong = oracle.simulate_weak_station("n3pay", blk, 99, ref_power=2000.0, tgt_power=30000.0)   # both blocks well above 1 LSB
    with tdoa_amd.Context(window_len=blk) as c:
        c.synth_weak_capture(0, blk, lle, oracle.DEFAULT_TX, 99)
        c.synth_weak_capture(1, blk, lle, oracle.DEFAULT_TX, 99, ref_power=2000.0, tgt_power=30000.0)
        got = c.capture_download(0, 0, 3 * blk)
        got_strong = c.capture_download(1, 0, 3 * blk)
    for g, w in ((got, want), (got_strong, strong)):
        diff = g.astype(np.int16) - w.astype(np.int16)
        assert np.abs(diff).max() <= 1
        assert (diff != 0).mean() < 2e-3
    assert strong.std() > 2.0 and np.array_equal(want[:2 * blk], np.full(2 * blk, 127, np.uint8))   # default REF block is below 1 LSB


def test_cfg3_batch_of_device_generated_windows(oracle):
    """a small cfg3-shaped batch (3 weak-simulator stations generated in HBM, all windows x 3 pairs): every peak against
    the oracle run on the downloaded bytes"""
    import tdoa_amd
    blk, wl, ml = 400_000, 200_000, 20000
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        for s, nm in enumerate(oracle.COLLECTORS):
            c.synth_weak_capture(s, blk, oracle.STATIONS[nm], oracle.DEFAULT_TX, oracle.SEED_BASE + s, tgt_power=20000.0)
        peaks = c.process()
        caps = [c.capture_download(s, 0, 3 * blk) for s in range(3)]
    assert peaks.shape == (6, 3)
    for wid in (2, 3):                                                    # the two target-block windows
        off = (wid // 2) * blk + (wid % 2) * wl
        pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0] for cp in caps]
        for p, (i, j) in enumerate(((0, 1), (0, 2), (1, 2))):
            olag, ocorr, _ = oracle.b_xcorr_peak_fft(pre[i], pre[j], ml)
            assert peaks[wid, p]["lag"] == olag and abs(peaks[wid, p]["corr"] - ocorr) <= REL_TOL * abs(ocorr)
    assert not peaks[[0, 1, 4, 5]]["lag"].any() and not peaks[[0, 1, 4, 5]]["corr"].any()      # constant reference blocks


def test_cfg5_sixteen_stations_120_pairs(oracle):
    """BASELINE config 5 geometry in miniature: 16 collectors, 120 pairs ordered i<j, 4 Msps (the sample rate only
    scales the time axis), two launch groups"""
    import tdoa_amd
    blk, wl, ml = 24000, 12000, 150
    rng = np.random.default_rng(16)
    delays = [int(x) for x in rng.integers(0, 60, size=16)]
    caps = [oracle.simulate_delayed_fm(3 * blk, d, 1616, 300 + i) for i, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=ml, window_len=wl, sample_rate=4e6, windows_per_batch=4) as c:
        peaks = c.process_u8(caps)
        assert c.num_pairs() == 120
    assert peaks.shape == (6, 120)
    pairs = [(i, j) for i in range(16) for j in range(i + 1, 16)]
    for wid in (0, 5):
        off = (wid // 2) * blk + (wid % 2) * wl
        pre = [oracle.b_preprocess(cp[2 * off:2 * (off + wl)])[0] for cp in caps]
        for p, (i, j) in enumerate(pairs):
            olag, ocorr = oracle.b_xcorr_peak(pre[i], pre[j], ml)
            assert peaks[wid, p]["lag"] == olag == delays[j] - delays[i], (wid, i, j)
            assert abs(peaks[wid, p]["corr"] - ocorr) <= REL_TOL * abs(ocorr)
    # all windows agree on the delays (the signal is stationary)
    assert (peaks["lag"] == np.array([delays[j] - delays[i] for i, j in pairs])).all()


def test_many_stations_batch_is_split_below_the_grid_limit(oracle):
    """windows x pairs of one launch group become gridDim.y (limit 65535): 40 stations = 780 pairs, 90
    windows -> 70 200 pair-windows, more than one group even though the caller asked for a single one"""
    import tdoa_amd
    S, blk, wl, ml = 40, 3000, 100, 20
    base = oracle.simulate_delayed_fm(3 * blk + 64, 0, 4040, 1)
    caps = [base[2 * (s % 7):2 * (s % 7) + 6 * blk].copy() for s in range(S)]         # station s delayed by -(s % 7)
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        peaks = c.process_u8(caps)
    assert peaks.shape == (90, 780)
    pairs = [(i, j) for i in range(S) for j in range(i + 1, S)]
    want = np.array([(i % 7) - (j % 7) for i, j in pairs])
    assert (peaks["lag"] == want).all()


def test_sharded_ingest_uploads_only_owned_windows(oracle):
    """tdoa_capture_upload_range: each of two ranks uploads the sample runs of the windows it owns (half the bytes) and
    the merged result equals the single-rank result on fully uploaded captures, byte for byte"""
    import tdoa_amd
    from tdoa_amd import sharding
    blk, wl, ml = 30000, 10000, 300
    caps = [oracle.simulate_station(nm, blk, oracle.SEED_BASE + i, tx_power=200000.0) for i, nm in enumerate(oracle.COLLECTORS)]
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        full = c.process_u8(caps)
    parts = []
    for r in range(2):
        with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
            sent = [c.capture_upload_owned(s, cp, r, 2, wl) for s, cp in enumerate(caps)]
            assert sent == [2 * len(sharding.owned_windows(r, 2, 9)) * wl] * 3            # 5 or 4 of the 9 windows
            parts.append(c.process(rank=r, world=2))
    merged = sharding.merge_sharded(np.stack([sharding.peaks_as_bytes(p) for p in parts]), 9, 3)
    assert np.array_equal(merged, full)
