"""GPU: the captures bench.py gives its multi-rank jobs (--sim fmdelay: one FM carrier from TX, every station's copy
delayed by its propagation time) through the batched path at the timed geometry -- every window of every block must put
every pair's peak exactly at delay_j - delay_i, and the least-squares fix on the ellipsoid must land within 150 m of the
transmitter (one sample = 150 m of range, PROJECT_NOTES.md:29-32).  This is the check bench.py applies to its own timed
result; here it runs as a test, on three stations and on eight."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_stations", [3, 8])
def test_fm_captures_with_true_delays_locate_the_transmitter(n_stations):
    import torch
    import tdoa_amd
    import bench
    fs, wl, blk = 2e6, 1_100_000, 2_200_000
    stations = bench.station_table(n_stations)
    delays = bench.propagation_delays(stations, fs)
    pairs = [(i, j) for i in range(n_stations) for j in range(i + 1, n_stations)]
    want = np.array([delays[j] - delays[i] for i, j in pairs])
    assert np.abs(want).max() < 400 and len(set(delays)) > 1
    with tdoa_amd.Context(max_lag=20000, window_len=wl, sample_rate=fs) as c:
        bufs = bench.synth_torch_captures(c, torch, "fmdelay", n_stations, blk, bench.SEED_BASE, delays)
        peaks = c.process()
        assert c.last_k1(0)[1]                                       # the single-look path
        again = c.process()
        # the other form of the decimated pair step (8 stations: the library walks the spectrum columns, 3: 4096-bin tiles)
        c.debug_flags(no_dec_cols=True) if n_stations > 3 else c.debug_flags(dec_cols_always=True)
        other = c.process()
        del bufs
    assert np.array_equal(other["lag"], peaks["lag"])
    assert np.abs(other["corr"] - peaks["corr"]).max() <= 5e-7 * np.abs(peaks["corr"]).max()
    assert peaks.shape == (6, len(pairs)) and np.array_equal(peaks, again)
    assert (peaks["lag"] == want[None, :]).all(), peaks["lag"]
    assert (peaks["abs_corr"] > 100.0).all()                         # strong peaks: sqrt(1.1e6) = 1049 at full correlation
    rd = np.median(peaks["lag"][2:4], axis=0) / fs * 299792458.0     # target-block windows, like processor.go:892-903
    wgt = np.median(peaks["abs_corr"][2:4], axis=0).astype(np.float64)
    rc, lle, it = tdoa_amd.capi.solve_surface(stations, rd, weights=wgt, height_m=sum(s[2] for s in stations) / n_stations)
    err = math.dist(tdoa_amd.capi.latlon_to_ecef(float(lle[0]), float(lle[1]), bench.TX[2]), tdoa_amd.capi.latlon_to_ecef(*bench.TX))
    assert rc == 0 and err < 150.0, (rc, lle, err)
