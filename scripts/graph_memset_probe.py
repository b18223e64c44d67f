#!/usr/bin/env python3
"""One-off probe for DESIGN.md section 7 (VERDICT r02 item 6a / ADVICE): what does a captured tdoa_process step look like
with hipMemsetAsync nodes in it, and does the round-2 anomaly (two processes on one card, lock-stepped graph replays,
un-owned peak records not zero from the second replay on) come back with them?  ONE run, no repetition loops.
  part 1 (one process): default step graph -> node / edge / root / memset counts;
                        TDOA_DEBUG_MEMSET_NODES=1 -> the same + Graphviz dump (memset node parameters, edges)
  part 2 (two processes, gloo barrier between replays): 4 lock-stepped replays per rank with memset nodes, then 4 with kernel
                        nodes; after every replay each rank checks that the windows it does not own are all-zero bytes
usage (GPU box, repo root):  python3 scripts/graph_memset_probe.py > gpurun_out/graph_memset_probe.txt"""
import os
import re
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tdoa-geolocation_amd")):
    sys.path.insert(0, p)
BLOCK, WLEN, MAX_LAG = 30000, 10000, 300


def captures():
    from oracle import pyoracle as oracle               # input bytes only
    return [oracle.simulate_delayed_fm(3 * BLOCK, d, 77, 10 + i) for i, d in enumerate((0, 13, 40))]


def part1(memset):
    os.environ["TDOA_DEBUG_MEMSET_NODES"] = "1" if memset else "0"
    import tdoa_amd
    with tdoa_amd.Context(max_lag=MAX_LAG, window_len=WLEN) as c:
        for s, cap in enumerate(captures()):
            c.capture_upload(s, cap)
        a = c.process()
        b = c.process()                                  # replay
        dot = os.path.join(ROOT, "gpurun_out", "step_graph_%s.dot" % ("memset" if memset else "kernels"))
        info = c.graph_info(dot)
    print("part 1, memset nodes %s: %r; replay identical: %s" % (memset, info, bool((a == b).all())))
    txt = open(dot).read()
    # hipGraphDebugDotPrint's node lines are not of one shape across ROCm versions: count "label" lines and arrows, and print
    # every line that mentions a memset verbatim (its parameters -- element size, width, height, value -- are in the label)
    labels = [ln for ln in txt.splitlines() if "label" in ln and "->" not in ln]
    edges = re.findall(r'->', txt)
    print("  dot file: %d node lines, %d edges" % (len(labels), len(edges)))
    if os.path.exists(dot + ".memsets"):                  # hipGraphMemsetNodeGetParams of every memset node (tdoa_debug_graph_info)
        for ln in open(dot + ".memsets").read().splitlines():
            print("  " + ln)
    return info


def worker(rank, world, port, memset, q):
    os.environ["TDOA_DEBUG_MEMSET_NODES"] = "1" if memset else "0"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch                                         # noqa: F401  (one HIP runtime in the process)
    import torch.distributed as dist
    import tdoa_amd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bad = []
    with tdoa_amd.Context(max_lag=MAX_LAG, window_len=WLEN) as c:
        for s, cap in enumerate(captures()):
            c.capture_upload(s, cap)
        _, n_windows = c.num_windows()
        for it in range(5):                              # call 0 captures, calls 1..4 replay
            dist.barrier()
            pk = c.process(rank=rank, world=world)
            raw = pk.view(np.uint8).reshape(n_windows, -1)
            for wid in range(n_windows):
                if wid % world != rank and raw[wid].any():
                    bad.append((it, wid))
        info = c.graph_info()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, info, bad))


def part2(memset):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=worker, args=(r, 2, port, memset, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=300) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for rank, info, bad in res:
        print("part 2, memset nodes %s, rank %d: graph %r; (replay, window) pairs with non-zero un-owned records: %r"
              % (memset, rank, info, bad))


if __name__ == "__main__":
    part1(False)
    part1(True)
    part2(True)
    part2(False)
