#!/usr/bin/env python3
"""PCIe-inclusive rate of BASELINE config 2: host .dat bytes (3 x 400 MB, pageable numpy buffers and files
through the pinned double-buffered reader) -> HBM -> peaks on the host.  Never the bench value; DESIGN.md
section 6 quotes it next to the HBM-resident rate."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tdoa-geolocation_amd"))
import numpy as np
import tdoa_amd

S, BLOCK = 3, 66_666_666
rng = np.random.default_rng(3)
caps = [rng.integers(96, 160, size=2 * 3 * BLOCK, dtype=np.uint8) for _ in range(S)]
with tdoa_amd.Context() as c:
    c.process_u8([x[:2 * 3 * 2_000_000] for x in caps])          # warm-up: plans, tables
    for rep in range(2):
        t0 = time.perf_counter()
        peaks = c.process_u8(caps)
        dt = time.perf_counter() - t0
    n = S * peaks.shape[0] * 2_000_000
    print(json.dumps({"path": "tdoa_process_u8 (pageable host buffers)", "seconds": round(dt, 4),
                      "Msamples_per_s": round(n / dt / 1e6, 1), "GB_per_s_h2d": round(S * caps[0].size / dt / 1e9, 2)}))
    t0 = time.perf_counter()
    c.process(want_host=True)
    dt2 = time.perf_counter() - t0
    print(json.dumps({"path": "tdoa_process (bytes resident in HBM)", "seconds": round(dt2, 4),
                      "Msamples_per_s": round(n / dt2 / 1e6, 1)}))
    d = tempfile.mkdtemp(dir="/dev/shm")
    paths = []
    for i, x in enumerate(caps):
        p = os.path.join(d, "st%d-1754900000.dat" % i)
        x.tofile(p)
        paths.append(p)
    try:
        for rep in range(2):
            t0 = time.perf_counter()
            for i, p in enumerate(paths):
                c.capture_upload_file(i, p)
            peaks = c.process()
            dt3 = time.perf_counter() - t0
        print(json.dumps({"path": "tdoa_capture_upload_file x3 (page cache -> pinned staging -> HBM) + tdoa_process",
                          "seconds": round(dt3, 4), "Msamples_per_s": round(n / dt3 / 1e6, 1),
                          "GB_per_s_h2d": round(S * caps[0].size / dt3 / 1e9, 2)}))
    finally:
        for p in paths:
            os.remove(p)
        os.rmdir(d)
