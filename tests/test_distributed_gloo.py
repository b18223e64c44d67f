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
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_windows = 3 * (BLOCK // WLEN)
    want = _peaks_for(oracle, _captures(oracle), range(n_windows))
    for r in range(2):
        got = np.frombuffer(results[r], dtype=PEAK_DTYPE).reshape(n_windows, 3)
        assert np.array_equal(got, want)                      # every rank holds the complete result
    # the gathered peaks carry the true delays (13, 40, 27 samples) on every window
    assert (want["lag"] == np.array([13, 40, 27])).all()
