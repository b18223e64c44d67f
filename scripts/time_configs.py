#!/usr/bin/env python3
"""Times tdoa_process on the window geometries of BASELINE configs 3, 4 and 5 (one GPU, small
window counts) -- these are parity-test configurations, not the bench line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tdoa-geolocation_amd"))
import numpy as np
import tdoa_amd

ST = [(41.18660274289527, -95.96064116595667, 355.69), (41.24669616513154, -96.08366304481238, 329.0),
      (41.32916620016985, -96.03513381562004, 373.18)]
TX = (41.20, -96.00, 400.0)


def run(name, n_st, fs, wlen, block, steps=3):
    c = tdoa_amd.Context(sample_rate=fs, window_len=wlen, max_lag=20000)
    rng = np.random.default_rng(1)
    for s in range(n_st):
        lle = ST[s] if s < 3 else (41.25 + 0.1 * rng.standard_normal(), -96.0 + 0.1 * rng.standard_normal(), 350.0)
        c.synth_capture(s, block, lle, TX, 0x5D0A0000 + s)
    wpb, W = c.num_windows()
    P = c.num_pairs()
    c.process(want_host=False)
    t0 = time.perf_counter()
    for _ in range(steps):
        c.process(want_host=False)
    dt = (time.perf_counter() - t0) / steps
    n, n1, n2 = c.plan_info()
    wlen = min(wlen, block)                                   # a block shorter than window_len is one window
    print(json.dumps({"config": name, "stations": n_st, "pairs": P, "windows": W, "window_len": wlen, "fft": [n, n1, n2],
                      "ms_per_step": round(dt * 1e3, 3), "Msamples_per_s": round(n_st * W * wlen / dt / 1e6, 1)}), flush=True)
    c.close()


if __name__ == "__main__":
    run("cfg1 (3 st, 2 000 000 samples each, one 666 666-sample window per block)", 3, 2e6, 2_000_000, 666_666, steps=20)
    run("cfg2 slice (3 st, 1 s windows)", 3, 2e6, 2_000_000, 8_000_000)
    run("cfg4 slice (8 st, 28 pairs)", 8, 2e6, 2_000_000, 8_000_000)
    run("cfg5 slice (16 st, 4 Msps, 1 s windows)", 16, 4e6, 4_000_000, 8_000_000)
    run("cfg3 slice (3 st, 10 s windows)", 3, 2e6, 20_000_000, 40_000_000, steps=1)
