#!/usr/bin/env python3
"""Stop-band attenuation of the decimation filter (k_pair_decimate16) against the error it leaves on the searched lags,
in numpy float64 on the CPU (no GPU needed): for each Kaiser design, taps per side and max |error| of the packed lags
relative to the peak, on an FM pair (strong peak) and on noise-level simulator.go / weak-simulator peaks.
Round 4 used it to pick 12 steps per phase (T <= 95: 126 dB on cfg2's transition band) instead of 14 (T = 106, 140 dB)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as oracle
oracle.build()
L, N, D, ML = 2_000_000, 1 << 21, 16, 20000
nc, r = N // 2, N // 2 // D
mp = ML // 2 + 2
sim = [oracle.simulate_station(nm, L, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
weak = [oracle.simulate_weak_station(nm, L, oracle.SEED_BASE + i) for i, nm in enumerate(oracle.COLLECTORS)]
cases = [("delayed_fm", oracle.simulate_delayed_fm(L, 0, 4242, 1), oracle.simulate_delayed_fm(L, 37, 4242, 2)),
         ("simulator.go ref 0-1", sim[0][:2 * L], sim[1][:2 * L]),
         ("simulator.go ref 1-2", sim[1][:2 * L], sim[2][:2 * L]),
         ("simulator.go tgt 0-2", sim[0][2*L:4 * L], sim[2][2*L:4 * L]),
         ("weak tgt 0-1", weak[0][2*L:4 * L], weak[1][2*L:4 * L])]
pre = []
for name, a, b in cases:
    ta, tb = oracle.b_preprocess(a)[0].astype(np.float64), oracle.b_preprocess(b)[0].astype(np.float64)
    c = np.fft.irfft(np.conj(np.fft.rfft(ta, N)) * np.fft.rfft(tb, N), N)
    q = c[0::2] + 1j * c[1::2]
    pre.append((name, q, np.fft.fft(q)))
m = np.arange(-mp, mp + 1)
idx = np.arange(r) * D
for att in (90, 100, 105, 110, 115, 120, 130, 140):
    dw = 2 * np.pi * (r - 2 * mp) / nc
    th = int(np.ceil((att - 8.0) / (2.285 * dw) / 2.0))
    beta = 0.1102 * (att - 8.7)
    t = np.arange(-th, th + 1)
    h = (np.sinc(t / D) * np.i0(beta * np.sqrt(1.0 - (t / th) ** 2)) / np.i0(beta)).astype(np.float32).astype(np.float64)
    w = (h[None, :] * np.cos(2 * np.pi * t[None, :] * m[:, None] / nc)).sum(axis=1) / D
    out = []
    for name, q, Q in pre:
        G = np.zeros(r, dtype=complex)
        for tt, ht in zip(t, h):
            G += ht * Q[(idx + tt) % nc]
        est = (np.fft.ifft(G) * r)[m % r] / w
        ref = (q * nc)[m % nc]
        # error relative to the PEAK among the searched lags (what the 1e-5 bar refers to), real & imag parts are lags
        pk = max(np.abs(ref.real).max(), np.abs(ref.imag).max())
        err = max(np.abs((est - ref).real).max(), np.abs((est - ref).imag).max()) / pk
        out.append(err)
    print("att %3d dB  T %3d (%3d taps, steps %2d)  flat %.1e   " % (att, th, 2*th+1, (2*th+1+15)//16, np.abs(w-1).max()) + "  ".join("%.1e" % e for e in out), flush=True)
