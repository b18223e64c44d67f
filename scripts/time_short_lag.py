#!/usr/bin/env python3
"""cfg2 (3 stations x 100 s, 99 windows, 3 pairs) with search ranges shorter than the reference's 20000 lags:
the segment form (ranges up to 1024 lags; with and without station transforms shared by the pairs of a window) and the
short-lag inverse (|lag| < 4095, no V round trip) against the general pruned form (TDOA_NO_SHORT_LAG=1); graph replay."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tdoa-geolocation_amd"))
import tdoa_amd

ST = [(41.18660274289527, -95.96064116595667, 355.69), (41.24669616513154, -96.08366304481238, 329.0),
      (41.32916620016985, -96.03513381562004, 373.18)]
TX = (41.20, -96.00, 400.0)

for max_lag in (128, 512, 1023, 2047, 4095, 20000):
    row = {"max_lag": max_lag}
    for mode in ("segment", "segment_pairwise", "short", "general"):
        with tdoa_amd.Context(max_lag=max_lag) as c:
            c.debug_flags(no_short_lag=(mode == "general"), no_segment_form=not mode.startswith("segment"),
                          no_segment_quads=(mode == "segment_pairwise"))
            for s in range(3):
                c.synth_capture(s, 66_666_666, ST[s], TX, 0x5D0A0000 + s)
            c.process(want_host=False)
            c.process(want_host=False)
            t0 = time.perf_counter()
            for _ in range(5):
                c.process(want_host=False)
            dt = (time.perf_counter() - t0) / 5
            row[mode + "_ms"] = round(dt * 1e3, 3)
            row[mode + "_Gsamples_per_s"] = round(3 * 99 * 2e6 / dt / 1e9, 1)
    print(json.dumps(row), flush=True)
