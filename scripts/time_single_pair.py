#!/usr/bin/env python3
"""Latency of one mode-B pair call on host buffers (tdoa_fm_xcorr_u8: upload two windows, demodulate, correlate,
peak back on the host) at the reference's call size, 2 000 000 samples per window."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tdoa-geolocation_amd"))
import numpy as np
import tdoa_amd

rng = np.random.default_rng(5)
n = 2_000_000
a = rng.integers(96, 160, size=2 * n, dtype=np.uint8)
b = np.roll(a, 2 * 57)
with tdoa_amd.Context() as c:
    for max_lag in (20000, 128):
        c.fm_xcorr(a, b, max_lag)
        t0 = time.perf_counter()
        for _ in range(20):
            lag, corr = c.fm_xcorr(a, b, max_lag)
        dt = (time.perf_counter() - t0) / 20
        print("tdoa_fm_xcorr_u8 2 x %d samples, max_lag %5d: %.2f ms per call (lag %d)" % (n, max_lag, dt * 1e3, lag), flush=True)
