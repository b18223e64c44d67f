"""GPU: the BATCHED column walk of the decimated pair step -- the form that dominates BASELINE configs 4 and 5 -- directly
against the oracle (until round 5 it was only compared with the tile form, the full inverse and the
geometry's lags).  Since round 5 the library runs it with the stations' rows staged in LDS (k_pair_decimate_staged,
csrc/dec_staged.hpp: one workgroup per window and 64-column block, a loader wave and one walk per compute wave);
k_pair_decimate_cols<256> / <512> (one pair-window per wave straight from memory, four per workgroup) is what
TDOA_DEBUG_NO_DEC_STAGED leaves.  BOTH are held against the oracle here.

Every (window, pair) of a tdoa_process batch, in the reference's pair order i < j (processor.go:816-850), is held against
oracle.b_xcorr_peak_fft (ob_* codes, float64 FFT correlation) on the bytes downloaded from the device: lag identical, corr
within north_star's 1e-5; and two units of every batch against the independent float64 atan2 pipeline
(oracle/float_pipeline.py).  The batches are sized so that the LAST workgroup of the walk is partial where the pair count
allows it (pair-windows % 4 != 0: waves 2-3 of the last workgroup have nothing to walk)."""
import numpy as np
import pytest

from oracle import float_pipeline as fp

pytestmark = pytest.mark.gpu

ML = 20000


def _check_batch(oracle, c, peaks, n_stations, wl, blk, float_units, capsys, tag):
    pairs = [(i, j) for i in range(n_stations) for j in range(i + 1, n_stations)]
    wpb, n_windows = c.num_windows()
    assert peaks.shape == (n_windows, len(pairs))
    worst, rows = 0.0, []
    for wid in range(n_windows):
        first = (wid // wpb) * blk + (wid % wpb) * wl
        raw = [c.capture_download(s, first, wl) for s in range(n_stations)]
        pre = [oracle.b_preprocess(x)[0] for x in raw]
        for p, (i, j) in enumerate(pairs):
            olag, ocorr, _ = oracle.b_xcorr_peak_fft(pre[i], pre[j], ML)
            g = peaks[wid, p]
            assert int(g["lag"]) == olag, (tag, wid, (i, j), int(g["lag"]), olag)
            dev = abs(float(g["corr"]) - ocorr) / abs(ocorr)
            worst = max(worst, dev)
            assert dev < 1e-5, (tag, wid, (i, j), dev)
            if (wid, p) in float_units:
                flag, fcorr, _ = fp.xcorr_peak_u8(raw[i], raw[j], ML)
                fdev = abs(float(g["corr"]) - fcorr) / abs(fcorr)
                assert int(g["lag"]) == flag and fdev < 1e-5, (tag, wid, (i, j), fdev)
                rows.append((wid, i, j, olag, ocorr, fdev))
    with capsys.disabled():
        print("\n  %s: %d pair-windows vs ob_* (f64 FFT): lag identical, worst |dcorr|/|corr| %.2e" % (tag, peaks.size, worst))
        for r in rows:
            print("    window %d pair %d-%d lag %6d corr %12.5f  vs float64 atan2 pipeline %.2e" % r)


@pytest.mark.parametrize("n_stations", [8, 5])
def test_column_walk_batch_4096x256_vs_oracle(oracle, n_stations, capsys):
    """8 stations: 28 pairs x 3 windows = 84 pair-windows -- staged: two groups of 14 walks per window and column block;
    per-pair walk: 21 full workgroups of four --; 5 stations: 10 x 3 = 30 -- staged: one group of ten walks; per-pair walk: the
    eighth workgroup half empty.  Delayed FM content per block, every station its own delay and noise."""
    import tdoa_amd
    wl = blk = 1_100_000
    delays = [0, 41, -17, 203, -350, 19, 77, -5][:n_stations]
    caps = [np.concatenate([oracle.simulate_delayed_fm(blk, 400 + d, 900 + k, 100 * (s + 1) + k) for k in range(3)])
            for s, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=ML, window_len=wl) as c:
        peaks = c.process_u8(caps)
        assert tuple(c.plan_info())[1:] == (4096, 256) and c.last_k1(0)[1]
        assert peaks.size % 4 == (0 if n_stations == 8 else 2)
        # (more pairs than stations: the library walks the columns by itself -- asserted through the tile form differing in
        # the last bits, not being the same numbers)
        c.debug_flags(no_dec_cols=True)
        tiles = c.process()
        c.debug_flags(no_dec_staged=True)                            # one pair-window per wave, rows straight from memory
        walk = c.process()
        c.debug_flags()
        assert np.array_equal(tiles["lag"], peaks["lag"]) and not np.array_equal(tiles["corr"], peaks["corr"])
        # the staged walk and the per-pair walk run the same arithmetic in the same order: the same bits
        assert np.array_equal(walk, peaks)
        _check_batch(oracle, c, peaks, n_stations, wl, blk, {(0, 0), (2, len(delays))}, capsys,
                     "k_pair_decimate_staged<256>, %d stations" % n_stations)
    want = np.array([delays[j] - delays[i] for i in range(n_stations) for j in range(i + 1, n_stations)])
    assert (peaks["lag"] == want[None, :]).all()


def test_column_walk_batch_4096x512_vs_oracle(oracle, capsys):
    """4 stations x 3 windows of 2 200 001 samples (odd: the last element of a window holds one sample):
    2 200 001 + 20 000 -> N = 2^22 = 2 x 4096 x 512, k_pair_decimate_staged<512, 8>; 6 pairs x 3 windows = 18 pair-windows (one
    group of six walks; the per-pair walk's fifth workgroup half empty)"""
    import tdoa_amd
    wl = blk = 2_200_001
    delays = [0, -123, 64, 1999]
    caps = [np.concatenate([oracle.simulate_delayed_fm(blk, 2100 + d, 1900 + k, 100 * (s + 1) + k) for k in range(3)])
            for s, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=ML, window_len=wl) as c:
        peaks = c.process_u8(caps)
        assert tuple(c.plan_info())[1:] == (4096, 512) and c.last_k1(0)[1]
        assert peaks.size == 18
        c.debug_flags(no_dec_staged=True)
        walk = c.process()
        c.debug_flags()
        assert np.array_equal(walk, peaks)
        _check_batch(oracle, c, peaks, 4, wl, blk, {(0, 1), (1, 5)}, capsys, "k_pair_decimate_staged<512>, 4 stations")
    want = np.array([delays[j] - delays[i] for i in range(4) for j in range(i + 1, 4)])
    assert (peaks["lag"] == want[None, :]).all()


@pytest.mark.parametrize("n_stations,wl", [(7, 1_100_000), (16, 1_100_000), (16, 2_200_001), (2, 1_100_000), (11, 1_100_000),
                                           (13, 1_100_000), (9, 2_200_001)])
def test_staged_walk_group_geometries_vs_the_per_pair_walk(oracle, n_stations, wl):
    """the workgroup geometries of the staged walk: 7 stations = 21 pairs in two groups of 11 and 10 (an idle compute wave in the
    second), 2 stations = one pair (a two-wave workgroup); more than eight stations: the greedy share-out of build_stg_groups --
    groups of at most 15 pairs that touch at most eight stations, each station at its rank among the group's stations in the LDS
    ring (16 stations: nine groups, the first three the 15 pairs of six stations; 9, 11, 13: with merged leftovers), on the
    4096 x 256 and the 4096 x 512 plan.  Reference: the per-pair walk (TDOA_DEBUG_NO_DEC_STAGED), itself held against the oracle
    above -- the same bits -- and the geometry's lags."""
    import tdoa_amd
    blk = wl
    rng = np.random.default_rng(100 + n_stations)
    delays = [int(x) for x in rng.integers(0, 300, size=n_stations)]
    caps = [np.concatenate([oracle.simulate_delayed_fm(blk, d, 700 + k, 100 * (s + 1) + k) for k in range(3)])
            for s, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=ML, window_len=wl) as c:
        c.debug_flags(dec_cols_always=True)                          # (two stations: the library's own choice is the tile form)
        peaks = c.process_u8(caps)
        assert c.last_k1(0)[1]
        c.debug_flags(dec_cols_always=True, no_dec_staged=True)
        walk = c.process()
    assert peaks.shape == (3, n_stations * (n_stations - 1) // 2)
    assert np.array_equal(walk, peaks)
    want = np.array([delays[j] - delays[i] for i in range(n_stations) for j in range(i + 1, n_stations)])
    assert (peaks["lag"] == want[None, :]).all() and (peaks["abs_corr"] > 100.0).all()


@pytest.mark.parametrize("n_stations,folded", [(5, True), (8, True), (12, True), (16, False)])
def test_staged_walk_with_and_without_a_loader_wave(oracle, n_stations, folded, monkeypatch):
    """the two forms of the staged walk's workgroup: a loader wave next to at most fifteen walks, or up to sixteen walks whose
    last waves each bring one station's rows (the library takes the second where it makes fewer workgroups: 13 to 16 stations).
    Each forced on geometries the library would give the other (TDOA_STG_FOLDED_ALWAYS / TDOA_NO_STG_FOLDED, read when the
    context is made): the same bits as the per-pair walk"""
    import tdoa_amd
    monkeypatch.setenv("TDOA_STG_FOLDED_ALWAYS" if folded else "TDOA_NO_STG_FOLDED", "1")
    wl = 1_100_000
    rng = np.random.default_rng(300 + n_stations)
    delays = [int(x) for x in rng.integers(0, 300, size=n_stations)]
    caps = [np.concatenate([oracle.simulate_delayed_fm(wl, d, 800 + k, 100 * (s + 1) + k) for k in range(3)])
            for s, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=ML, window_len=wl) as c:
        peaks = c.process_u8(caps)
        c.debug_flags(dec_cols_always=True, no_dec_staged=True)
        walk = c.process()
    assert np.array_equal(walk, peaks)
    want = np.array([delays[j] - delays[i] for i in range(n_stations) for j in range(i + 1, n_stations)])
    assert (peaks["lag"] == want[None, :]).all() and (peaks["abs_corr"] > 100.0).all()


def test_staged_walk_in_launch_groups_and_shards(oracle):
    """the staged walk behind tdoa_process's batching and sharding: two windows per launch group (three windows: a group of two
    and a group of one), and the windows dealt to two ranks (wid % 2: rank 0 two windows, rank 1 one) -- every variant must
    return the bytes of the one-group, one-rank run (each pair-window's arithmetic does not depend on its neighbours)"""
    import tdoa_amd
    from tdoa_amd import sharding
    wl = blk = 1_100_000
    delays = [0, 41, -17, 203, -350]
    caps = [np.concatenate([oracle.simulate_delayed_fm(blk, 400 + d, 900 + k, 100 * (s + 1) + k) for k in range(3)])
            for s, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=ML, window_len=wl) as c:
        full = c.process_u8(caps)
        parts = [c.process(rank=r, world=2) for r in range(2)]
    with tdoa_amd.Context(max_lag=ML, window_len=wl, windows_per_batch=2) as c:
        grouped = c.process_u8(caps)
    assert np.array_equal(grouped, full)
    merged = sharding.merge_sharded(np.stack([sharding.peaks_as_bytes(p) for p in parts]), 3, 10)
    assert np.array_equal(merged, full)
    assert not parts[0][1]["lag"].any() and not parts[1][[0, 2]]["lag"].any()      # a rank leaves the windows it does not own at zero
