"""The C ABI from plain C99 (what cgo compiles against): examples/pair_from_c.c must build with
-std=c99 -pedantic -Wall -Wextra -Werror against include/tdoa_mi355x.h, fail loudly without a device and find the
delay with one."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tdoa-geolocation_amd")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import tdoa_amd
    tdoa_amd.build.build()
    out = str(tmp_path_factory.mktemp("cexample") / "pair_from_c")
    cmd = [shutil.which("gcc") or "gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "pair_from_c.c"),
           "-L" + PKG, "-ltdoa_mi355x", "-Wl,-rpath," + PKG, "-o", out]
    subprocess.check_call(cmd)
    return out


def test_c99_example_builds_and_reports_a_missing_device(exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no HIP device" in r.stderr          # no CPU fallback: the library says so


@pytest.mark.gpu
def test_c99_example_finds_the_delay(exe):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "lag 7 samples" in r.stdout and "plausible 1" in r.stdout
