#!/usr/bin/env python3
"""Randomised sweep of the staged column walk against the per-pair walk (must be bit-identical) and the tile form (5e-7 of the
peak): station counts 2..16, window lengths on both small plans (odd lengths, windows that leave rows of zero padding), launch
groups, two-rank sharding.  Not a test (minutes of GPU time); run on the GPU box:  python3 scripts/fuzz_staged.py [seed] [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tdoa-geolocation_amd")):
    sys.path.insert(0, p)
import torch  # noqa: F401,E402  (one HIP runtime per process: torch's)
import tdoa_amd  # noqa: E402
from oracle import pyoracle as o  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed)
bad = 0
for case in range(n_cases):
    S = int(rng.integers(2, 17))
    plan512 = bool(rng.integers(0, 2)) and S <= 8
    wl = int(rng.integers(2_100_000, 4_150_000)) if plan512 else int(rng.integers(1_060_000, 2_070_000))
    wpb = int(rng.integers(1, 3)) if wl < 1_500_000 and S <= 8 else 1
    blk = wl * wpb + int(rng.integers(0, 1000))
    per_batch = int(rng.integers(0, 3))
    delays = [int(x) for x in rng.integers(0, 500, size=S)]
    base = [o.simulate_delayed_fm(blk + 600, 0, 3000 + 10 * case + k, 1) for k in range(3)]       # one content per block
    caps = []
    for s, d in enumerate(delays):     # station s: the content delayed by d samples (a shifted copy) + nothing else: exact lags
        caps.append(np.concatenate([b[2 * (500 - d):2 * (500 - d + blk)] for b in base]))
    with tdoa_amd.Context(max_lag=20000, window_len=wl, windows_per_batch=per_batch) as c:
        c.debug_flags(dec_cols_always=True)
        got = c.process_u8(caps)
        plan = c.plan_info()
        c.debug_flags(dec_cols_always=True, no_dec_staged=True)
        walk = c.process()
        c.debug_flags(no_dec_cols=True)
        tiles = c.process()
        c.debug_flags(dec_cols_always=True)
        parts = [c.process(rank=r, world=2) for r in range(2)]
    want = np.array([delays[j] - delays[i] for i in range(S) for j in range(i + 1, S)])
    ok = np.array_equal(got, walk) and np.array_equal(got["lag"], tiles["lag"]) and (got["lag"] == want[None, :]).all()
    ok = ok and np.abs(got["corr"] - tiles["corr"]).max() <= 5e-7 * np.abs(tiles["corr"]).max()
    from tdoa_amd import sharding
    W, P = got.shape
    merged = sharding.merge_sharded(np.stack([sharding.peaks_as_bytes(p) for p in parts]), W, P)
    ok = ok and np.array_equal(merged, got)
    print("case %2d: S %2d wl %7d x %d windows/block, per_batch %d, plan %r: %s" % (case, S, wl, wpb, per_batch, tuple(plan), "ok" if ok else "MISMATCH"),
          flush=True)
    bad += 0 if ok else 1
sys.exit(1 if bad else 0)
