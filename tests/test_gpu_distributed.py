"""Two ranks sharing the one GPU of the test box (gloo for the collective, as bench.py's rehearsal mode does): the
product's sharded path end to end with the real kernels -- tdoa_process(rank, world) per rank, one all-gather of the
peak records, merge -- must reproduce the single-rank result exactly.  (The pair-major fallback for fewer windows
than ranks is checked rank by rank in test_gpu_edges.py.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCK, WLEN, MAX_LAG = 30000, 10000, 300


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _captures(oracle):
    return [oracle.simulate_delayed_fm(3 * BLOCK, d, 77, 10 + i) for i, d in enumerate((0, 13, 40))]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tdoa-geolocation_amd"))
    import torch                                            # torch first: one HIP runtime in the process
    import torch.distributed as dist
    import tdoa_amd
    from oracle import pyoracle as oracle                   # test-side: only generates the input bytes
    from tdoa_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    caps = _captures(oracle)
    with tdoa_amd.Context(max_lag=MAX_LAG, window_len=WLEN) as c:
        for s, cap in enumerate(caps):
            c.capture_upload(s, cap)
        _, n_windows = c.num_windows()
        local = c.process(rank=rank, world=world)           # other ranks' windows stay zero
        # graph replay, in lock step with the other rank (round 2: un-owned records came back non-zero from the second
        # replay on while the step graph still held hipMemsetAsync nodes): the records of the windows
        # this rank does not own must be all-zero BYTES after a replay, and the captured step one chain of kernel nodes
        dist.barrier()
        again = c.process(rank=rank, world=world)
        info = c.graph_info()
        assert info["memsets"] == 0 and info["roots"] == 1 and info["edges"] >= info["nodes"] - 1, info
        raw = again.view(np.uint8).reshape(n_windows, -1)
        for wid in range(n_windows):
            if wid % world != rank:
                assert not raw[wid].any(), (rank, wid)
        assert np.array_equal(again, local)
        full = c.process() if rank == 0 else None
    buf = torch.from_numpy(sharding.peaks_as_bytes(local).copy())
    gathered = sharding.all_gather_peaks(buf, dist)
    merged = sharding.merge_sharded(gathered.numpy(), n_windows, 3)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, merged.tobytes(), full.tobytes() if full is not None else b""))


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_reproduce_the_single_rank_result():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=480) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:                                          # no child outlives the test
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=30)
    res = {r: (m, f) for r, m, f in got}
    full = res[0][1]
    assert full and res[0][0] == full and res[1][0] == full
    from tdoa_amd.capi import PEAK_DTYPE
    peaks = np.frombuffer(full, dtype=PEAK_DTYPE).reshape(9, 3)
    assert (peaks[:, 0]["lag"] == 13).all() and (peaks[:, 1]["lag"] == 40).all() and (peaks[:, 2]["lag"] == 27).all()


def _worker_rccl(port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tdoa-geolocation_amd"))
    import torch
    import torch.distributed as dist
    import tdoa_amd
    from oracle import pyoracle as oracle                   # test-side: only generates the input bytes
    from tdoa_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    caps = _captures(oracle)
    with tdoa_amd.Context(max_lag=MAX_LAG, window_len=WLEN) as c:
        for s, cap in enumerate(caps):
            c.capture_upload(s, cap)
        _, n_windows = c.num_windows()
        dev = torch.zeros(n_windows * 3 * 16, dtype=torch.uint8, device="cuda")
        gathered = torch.zeros_like(dev)
        c.process(rank=0, world=1, out_dev_ptr=dev.data_ptr(), want_host=False)      # peak records stay in HBM
        dist.all_gather_into_tensor(gathered, dev)                                   # the RCCL call of bench.py --gpus N
        merged = sharding.bytes_as_peaks(gathered.view(1, -1).amax(dim=0).cpu().numpy(), n_windows, 3)
        full = c.process()
    dist.barrier()
    dist.destroy_process_group()
    q.put((merged.tobytes(), full.tobytes()))


@pytest.mark.timeout(600)
def test_rccl_all_gather_of_the_peak_records_at_world_size_one():
    """no 8-GPU node is available to the build: the RCCL branch of bench.py (tdoa_process writes the peak records to a device
    buffer, dist.all_gather_into_tensor on it, byte-wise owner merge) runs here on one GPU with a world of one rank"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl, args=(_free_port(), q))
    p.start()
    try:
        merged, full = q.get(timeout=480)
        p.join(timeout=60)
        assert p.exitcode == 0
    finally:
        if p.is_alive():
            p.terminate()
        p.join(timeout=30)
    assert merged == full
    from tdoa_amd.capi import PEAK_DTYPE
    peaks = np.frombuffer(full, dtype=PEAK_DTYPE).reshape(9, 3)
    assert (peaks[:, 0]["lag"] == 13).all() and (peaks[:, 1]["lag"] == 40).all() and (peaks[:, 2]["lag"] == 27).all()
