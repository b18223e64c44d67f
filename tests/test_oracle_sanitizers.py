"""The CPU oracle under AddressSanitizer + UBSan (CPU build only; GPU ASan is not available on
this pool).  Runs a compact tour of the oracle in a subprocess with libasan preloaded."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import pyoracle as o
o._SO = os.path.join(%(root)r, "oracle", "libtdoa_oracle_asan.so")
raw = o.simulate_station("kx0u", 3001, 7)
data = o.iq_u8_to_c64(raw)
ref, tgt = o.extract_reference(data), o.extract_target(data)
o.preprocess(ref)
o.cross_correlate(ref[:2500], ref)
o.cross_correlate(tgt, tgt)
o.simple_correlate(ref[:1500], ref)
o.fast_analyze_capture(raw)
o.time_domain_correlation(ref[:5], ref[:7], 20000)
w = o.simulate_weak_station("n3pay", 2000, 9)
p, st = o.b_preprocess(w[:2 * 1999])
o.b_xcorr_peak(p, p[:1500], 64)
o.b_refine_peak(p, p[:1500], 0, 120.0)
o.b_refine_peak(p[:3], p[:2], 63, 1.0)
o.b_xcorr_all_lags(p[:100], p[:90], 64)
o.block_power(w[:2000])
o.b_discriminate(w[:2])
o.simulate_delayed_fm(1000, 17, 1, 2)
o.b_preprocess_smooth(w[:2 * 777], 10)
o.b_preprocess_gate(w[:2 * 777], window=10, gate=1)
o.b_preprocess_gate(o.simulate_delayed_fm(500, 0, 1, 1), window=0, gate=1)
o.b_preprocess_gate(w[:2], gate=1)
o.b_envelope_code(0, 255)
o.solve_tdoa([o.STATIONS[k] for k in o.COLLECTORS], [10.0, -20.0, 0.0])
print("SANITIZED_OK")
"""


@pytest.mark.timeout(300)
def test_oracle_under_asan_ubsan():
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, env=env,
                       timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "SANITIZED_OK" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
