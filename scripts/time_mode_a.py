#!/usr/bin/env python3
"""Wall time of the drop-in crossCorrelate (mode A, the reference's executed chain, bit-exact) on the
GPU for the reference's own call size: two 2 000 000-sample complex64 signals (processor.go:772)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tdoa-geolocation_amd"))
import numpy as np
import tdoa_amd

rng = np.random.default_rng(0)
n = 2_000_000
raw = [rng.integers(120, 136, size=2 * n, dtype=np.uint8) for _ in range(2)]
with tdoa_amd.Context() as c:
    sig = [c.load_iq_u8(r) for r in raw]
    c.cross_correlate(sig[0][:50000], sig[1][:50000])
    for label, a, b in (("equal lengths (lag 0 only)", sig[0], sig[1]),
                        ("lag search over 20000 lags", sig[0][:n - 20000], sig[1])):
        t0 = time.perf_counter()
        d, corr = c.cross_correlate(a, b)
        dt = time.perf_counter() - t0
        print("crossCorrelate %-30s %8.1f ms  (delay %d, corr %.6f)" % (label, dt * 1e3, d, corr), flush=True)

    # weak-signal chain (power < 0.001: three notches, band-pass, smoothing -- 5084 complex adds per sample)
    weak = [rng.integers(127, 129, size=2 * n, dtype=np.uint8) for _ in range(2)]
    wsig = [c.load_iq_u8(r) for r in weak]
    c.cross_correlate(wsig[0][:50000], wsig[1][:50000])
    t0 = time.perf_counter()
    d, corr = c.cross_correlate(wsig[0], wsig[1])
    dt = time.perf_counter() - t0
    print("crossCorrelate %-30s %8.1f ms  (delay %d, corr %.6f)" % ("weak chain, equal lengths", dt * 1e3, d, corr), flush=True)
