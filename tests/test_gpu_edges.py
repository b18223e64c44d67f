"""Edge cases and error behaviour of the batched path at the C ABI: call order, ragged captures,
tiny captures, lag ranges longer than the window, argument checks.  Status codes instead of
aborts (the reference log.Fatalf's, processor.go:1067-1075; a library must not)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5
ERR_INVALID, ERR_UNSUPPORTED, ERR_STATE = 1, 5, 6


def test_call_order_and_argument_errors(oracle):
    import tdoa_amd
    with tdoa_amd.Context(max_lag=50, window_len=1000) as c:
        with pytest.raises(tdoa_amd.TdoaError) as e:
            c.process()                                              # nothing uploaded
        assert e.value.status == ERR_STATE
        cap = oracle.simulate_delayed_fm(3000, 0, 1, 1)
        c.capture_upload(0, cap)
        with pytest.raises(tdoa_amd.TdoaError) as e:
            c.process()                                              # one station: no pair
        assert e.value.status == ERR_STATE
        c.capture_upload(2, cap)                                     # station 1 left empty
        with pytest.raises(tdoa_amd.TdoaError) as e:
            c.process()
        assert e.value.status == ERR_STATE
        c.capture_upload(1, cap)
        assert c.process().shape == (3, 3)                           # blocks of 1000 samples = one window each
        for rank, world in ((0, 0), (2, 2), (-1, 1)):
            with pytest.raises(tdoa_amd.TdoaError) as e:
                c.process(rank=rank, world=world)
            assert e.value.status == ERR_INVALID
        with pytest.raises(tdoa_amd.TdoaError) as e:
            c.capture_upload(5000, cap)                              # station index out of range
        assert e.value.status == ERR_INVALID
        c.capture_clear()
        c.capture_upload(0, cap[:2 * 5])                             # 5 samples: blocks of 1 sample
        c.capture_upload(1, cap[:2 * 5])
        with pytest.raises(tdoa_amd.TdoaError) as e:
            c.process()
        assert e.value.status == ERR_UNSUPPORTED
    L = tdoa_amd.capi.load()
    assert L.tdoa_strerror(ERR_STATE) and L.tdoa_strerror(12345)      # every code has a message
    prm = tdoa_amd.capi.Params()
    L.tdoa_default_params(C.byref(prm))
    prm.max_lag = 0
    h = C.c_void_p()
    assert L.tdoa_create(C.byref(prm), C.byref(h)) == ERR_INVALID and not h.value


def test_ragged_captures_use_their_own_thirds(oracle):
    """processor.go:214 cuts every file into its own thirds; the window grid comes from the shortest capture."""
    import tdoa_amd
    wl, ml = 4000, 100
    lens = [12000, 13500, 12600]                                      # blocks of 4000, 4500, 4200 samples
    delays = [0, 13, 31]
    caps = []
    for s, (n, d) in enumerate(zip(lens, delays)):
        b = n // 3
        caps.append(np.concatenate([oracle.simulate_delayed_fm(b, d, 70 + k, 10 * s + k) for k in range(3)]
                                   + [np.zeros(2 * (n - 3 * b), np.uint8)]))
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as c:
        peaks = c.process_u8(caps)
        assert peaks.shape == (3, 3)                                 # one window per block
        pairs = [(0, 1), (0, 2), (1, 2)]
        for k in range(3):
            pre = [oracle.b_preprocess(cp[2 * k * (n // 3):2 * (k * (n // 3) + wl)])[0] for cp, n in zip(caps, lens)]
            for p, (i, j) in enumerate(pairs):
                olag, ocorr = oracle.b_xcorr_peak(pre[i], pre[j], ml)
                assert peaks[k, p]["lag"] == olag == delays[j] - delays[i]
                assert abs(peaks[k, p]["corr"] - ocorr) <= REL_TOL * abs(ocorr)
        q = c.window_quality_all()
        for k in range(3):
            for s, (cp, n) in enumerate(zip(caps, lens)):
                w = cp[2 * k * (n // 3):2 * (k * (n // 3) + wl)]
                assert q[k, s]["i_avg"] == w[0::2].astype(np.float64).sum() / wl


def test_block_shorter_than_window_and_lag_longer_than_window(oracle):
    import tdoa_amd
    n = 900                                                          # blocks of 300 samples < window_len
    a = oracle.simulate_delayed_fm(n, 0, 5, 1)
    b = oracle.simulate_delayed_fm(n, 4, 5, 2)
    with tdoa_amd.Context(max_lag=1000, window_len=2_000_000) as c:  # lags far beyond the 300-sample windows
        peaks = c.process_u8([a, b])
        assert peaks.shape == (3, 1)
        for k in range(3):
            ta, _ = oracle.b_preprocess(a[600 * k:600 * (k + 1)])
            tb, _ = oracle.b_preprocess(b[600 * k:600 * (k + 1)])
            olag, ocorr = oracle.b_xcorr_peak(ta, tb, 1000)
            assert peaks[k, 0]["lag"] == olag
            assert abs(peaks[k, 0]["corr"] - ocorr) <= REL_TOL * abs(ocorr)
        lags = c.fm_xcorr_lags(a[:600], b[:600], 1000)
        want = oracle.b_xcorr_all_lags(*[oracle.b_preprocess(x[:600])[0] for x in (a, b)], 1000)
        assert np.abs(lags - want).max() <= REL_TOL * np.abs(want).max()
        assert not want[:1000 - 300].any() and not want[1000 + 300:].any()    # no overlap beyond +-299 samples
        assert np.abs(lags[:700]).max() <= REL_TOL * np.abs(want).max()        # FFT rounding noise only


def test_two_sample_blocks(oracle):
    """the smallest capture the path accepts: blocks of 2 samples"""
    import tdoa_amd
    a = np.array([10, 200, 250, 3, 128, 127, 90, 91, 7, 255, 0, 64], np.uint8)
    b = a[::-1].copy()
    with tdoa_amd.Context(max_lag=4, window_len=1000) as c:
        peaks = c.process_u8([a, b])
        assert peaks.shape == (3, 1)
        for k in range(3):
            ta, _ = oracle.b_preprocess(a[4 * k:4 * k + 4])
            tb, _ = oracle.b_preprocess(b[4 * k:4 * k + 4])
            olag, ocorr = oracle.b_xcorr_peak(ta, tb, 4)
            assert peaks[k, 0]["lag"] == olag
            assert abs(peaks[k, 0]["corr"] - ocorr) <= REL_TOL * max(abs(ocorr), 1e-30)


def test_pair_major_sharding_when_fewer_windows_than_ranks(oracle):
    """W < G falls back to dealing (window, pair) units; every unit is computed exactly once and
    equals the single-rank result."""
    import tdoa_amd
    from tdoa_amd import sharding
    blk, ml = 6000, 120
    delays = [0, 7, 19, 33]
    caps = [oracle.simulate_delayed_fm(3 * blk, d, 808, 100 + i) for i, d in enumerate(delays)]
    with tdoa_amd.Context(max_lag=ml, window_len=blk) as c:
        for s, cap in enumerate(caps):
            c.capture_upload(s, cap)
        full = c.process()
        W, P = full.shape
        assert (W, P) == (3, 6)
        for world in (4, 8, 18, 19):                      # 18 units: world 19 leaves one rank without work
            parts = [c.process(rank=r, world=world) for r in range(world)]
            for wid in range(W):
                for p in range(P):
                    owner = sharding.unit_owner(wid, p, world, W, P)
                    for r in range(world):
                        if r == owner:
                            assert parts[r][wid, p] == full[wid, p]
                        else:
                            assert parts[r][wid, p]["lag"] == 0 and parts[r][wid, p]["corr"] == 0.0
            merged = sharding.merge_sharded(np.stack([sharding.peaks_as_bytes(x) for x in parts]), W, P)
            assert np.array_equal(merged, full)
        _, fine = c.process_fine(60.0)
        pk, fn = c.process_fine(60.0, rank=5, world=8)
        for wid in range(W):
            for p in range(P):
                if sharding.unit_owner(wid, p, 8, W, P) == 5:
                    assert fn[wid, p] == fine[wid, p]
                else:
                    assert fn[wid, p]["delay"] == 0.0



def test_two_contexts_interleaved_and_threads(oracle):
    """A Go host may hold one context per GPU in one process.  Two contexts on the
    same device, used alternately and from two threads, must not disturb each other (separate streams, buffers and
    captured graphs)."""
    import threading
    import tdoa_amd
    blk, wl, ml = 30000, 10000, 200
    caps_a = [oracle.simulate_delayed_fm(3 * blk, d, 31, 10 + i) for i, d in enumerate((0, 21, 55))]
    caps_b = [oracle.simulate_delayed_fm(3 * blk, d, 32, 20 + i) for i, d in enumerate((0, 8, 90))]
    with tdoa_amd.Context(max_lag=ml, window_len=wl) as ca, tdoa_amd.Context(max_lag=ml, window_len=wl) as cb:
        ra = ca.process_u8(caps_a)
        rb = cb.process_u8(caps_b)
        assert (ra[:, 0]["lag"] == 21).all() and (rb[:, 1]["lag"] == 90).all()
        for _ in range(3):                                   # alternate replays of the two captured graphs
            assert ca.process().tobytes() == ra.tobytes()
            assert cb.process().tobytes() == rb.tobytes()
        out = {}

        def run(name, ctx, want):
            ok = True
            for _ in range(10):
                ok = ok and ctx.process().tobytes() == want.tobytes()
            out[name] = ok

        ts = [threading.Thread(target=run, args=("a", ca, ra)), threading.Thread(target=run, args=("b", cb, rb))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert out == {"a": True, "b": True}
