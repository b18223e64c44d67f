"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): window-major sharding, one
all-gather of the per-pair peak records, merge, downstream solve.  The per-rank compute is the
CPU oracle standing in for the HIP kernels (no GPU here); the sharding / collective / merge code
is the product's (tdoa_amd.sharding), the same code bench.py and a multi-GPU host use."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


BLOCK, WLEN, MAX_LAG = 6000, 2000, 100


def _captures(oracle):
    return [oracle.simulate_delayed_fm(3 * BLOCK, d, 77, 10 + i) for i, d in enumerate((0, 13, 40))]


def _peaks_for(oracle, caps, wids):
    from tdoa_amd.capi import PEAK_DTYPE
    n_windows = 3 * (BLOCK // WLEN)
    out = np.zeros((n_windows, 3), dtype=PEAK_DTYPE)
    wpb = BLOCK // WLEN
    for wid in wids:
        off = (wid // wpb) * BLOCK + (wid % wpb) * WLEN
        pre = [oracle.b_preprocess(c[2 * off:2 * (off + WLEN)])[0] for c in caps]
        for p, (i, j) in enumerate([(0, 1), (0, 2), (1, 2)]):
            lag, corr = oracle.b_xcorr_peak(pre[i], pre[j], MAX_LAG)
            out[wid, p] = (lag, abs(corr), corr)
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tdoa-geolocation_amd"))
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as oracle
    from tdoa_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    caps = _captures(oracle)
    n_windows = 3 * (BLOCK // WLEN)
    mine = sharding.owned_windows(rank, world, n_windows)
    local = _peaks_for(oracle, caps, mine)                     # other ranks' windows stay zero
    buf = torch.from_numpy(sharding.peaks_as_bytes(local).copy())
    gathered = sharding.all_gather_peaks(buf, dist)
    merged = sharding.merge_sharded(gathered.numpy(), n_windows, 3)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, merged.tobytes()))


@pytest.mark.timeout(300)
def test_two_rank_sharding_allgather_merge(oracle):
    import torch.multiprocessing as mp
    from tdoa_amd.capi import PEAK_DTYPE
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        results = dict(q.get(timeout=240) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:                                          # no child outlives the test
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=30)
    n_windows = 3 * (BLOCK // WLEN)
    want = _peaks_for(oracle, _captures(oracle), range(n_windows))
    for r in range(2):
        got = np.frombuffer(results[r], dtype=PEAK_DTYPE).reshape(n_windows, 3)
        assert np.array_equal(got, want)                      # every rank holds the complete result
    # the gathered peaks carry the true delays (13, 40, 27 samples) on every window
    assert (want["lag"] == np.array([13, 40, 27])).all()


# ---- world size 8: BASELINE config 4 / 5 sharding (window-major) and the pair-major fallback -----------------------

def _fake_peaks(n_windows, n_pairs, units):
    """deterministic peak records for the given (wid, pair) units; the others stay zero"""
    from tdoa_amd.capi import PEAK_DTYPE
    out = np.zeros((n_windows, n_pairs), dtype=PEAK_DTYPE)
    for wid, p in units:
        corr = (-1.0) ** (wid + p) * (1.0 + wid * 0.25 + p * 1e-3)
        out[wid, p] = ((wid * 7 + p * 3) % 41 - 20, abs(corr), corr)
    return out


def _worker8(rank, world, port, n_windows, n_pairs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tdoa-geolocation_amd"))
    import torch
    import torch.distributed as dist
    from tdoa_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    units = [(w, p) for w in range(n_windows) for p in range(n_pairs)
             if sharding.unit_owner(w, p, world, n_windows, n_pairs) == rank]
    local = _fake_peaks(n_windows, n_pairs, units)
    gathered = sharding.all_gather_peaks(torch.from_numpy(sharding.peaks_as_bytes(local).copy()), dist)
    merged = sharding.merge_sharded(gathered.numpy(), n_windows, n_pairs)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, len(units), merged.tobytes()))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n_windows,n_pairs,label", [(99, 28, "cfg4: 8 stations, window-major"),
                                                     (3, 28, "fewer windows than ranks: pair-major"),
                                                     (300, 120, "cfg5: 16 stations, 300 windows")])
def test_eight_rank_ownership_allgather_merge(n_windows, n_pairs, label):
    import torch.multiprocessing as mp
    from tdoa_amd.capi import PEAK_DTYPE
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, n_windows, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=240) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:                                          # no child outlives the test, whatever happened above
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=30)
    want = _fake_peaks(n_windows, n_pairs, [(w, p) for w in range(n_windows) for p in range(n_pairs)])
    assert sum(n for _, n, _ in results) == n_windows * n_pairs, label          # every unit has exactly one owner
    counts = sorted(n for _, n, _ in results)
    assert counts[-1] - counts[0] <= max(n_pairs, 1), label                       # balanced to within one window
    for rank, _, blob in results:
        got = np.frombuffer(blob, dtype=PEAK_DTYPE).reshape(n_windows, n_pairs)
        assert np.array_equal(got, want), (label, rank)                            # every rank holds the complete result


def test_owned_sample_runs_cover_exactly_the_owned_windows():
    from tdoa_amd import sharding
    for n_samples, wlen, world in ((200_000_000, 2_000_000, 8), (60_000, 10_000, 2), (1_200_000_000, 4_000_000, 8),
                                   (2_000_000, 2_000_000, 8), (90_001, 7_000, 3)):
        block, wl, wpb = sharding.window_grid(n_samples, wlen)
        n_windows = 3 * wpb
        seen = np.zeros(n_samples, dtype=np.int8) if n_samples <= 2_000_000 else None
        total = 0
        for r in range(world):
            runs = sharding.owned_sample_runs(r, world, n_samples, wlen)
            assert all(a >= 0 and a + c <= n_samples for a, c in runs)
            assert all(runs[i][0] + runs[i][1] < runs[i + 1][0] for i in range(len(runs) - 1))   # merged and sorted
            if n_windows < world:
                assert runs == [(0, n_samples)]                  # pair-major fallback: any window may be needed
                continue
            owned = sharding.owned_windows(r, world, n_windows)
            assert sum(c for _, c in runs) == len(owned) * wl
            for wid in owned:                                     # every owned window lies inside one run
                first = (wid // wpb) * block + (wid % wpb) * wl
                assert any(a <= first and first + wl <= a + c for a, c in runs)
            total += sum(c for _, c in runs)
            if seen is not None:
                for a, c in runs:
                    seen[a:a + c] += 1
        if n_windows >= world:
            assert total == n_windows * wl                        # the ranks' runs partition the windowed samples
            if seen is not None:
                assert seen.max() == 1
