"""Builds libtdoa_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG_ROOT, "csrc")
# TDOA_LIB_VARIANT=<name>: load libtdoa_mi355x_<name>.so instead -- a build of the same sources with other -D switches
# (build_variant below), for same-box A/B measurements of compile-time choices; never set by tests, bench.py or the driver
_VARIANT = os.environ.get("TDOA_LIB_VARIANT", "")
LIB = os.path.join(PKG_ROOT, "libtdoa_mi355x%s.so" % ("_" + _VARIANT if _VARIANT else ""))
CLI = os.path.join(PKG_ROOT, "tdoa_processor")
CLI_SRC = os.path.join(CSRC, "host", "tdoa_processor.cpp")
HEADER = os.path.join(os.path.dirname(PKG_ROOT), "include", "tdoa_mi355x.h")


def _sources():
    out = [HEADER]
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".inc", ".cpp", ".h")):
            out.append(os.path.join(CSRC, f))
    return out


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built")


def needs_build():
    if _VARIANT:
        if not os.path.exists(LIB):
            raise RuntimeError("TDOA_LIB_VARIANT=%s: %s has not been built (tdoa_amd.build.build_variant)" % (_VARIANT, LIB))
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-pthread", "-Xarch_host", "-ffp-contract=off",
           "-o", LIB, os.path.join(CSRC, "tdoa_mi355x.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_cli(force=True, verbose=verbose)
    return LIB


def build_variant(name, defines, verbose=False):
    """libtdoa_mi355x_<name>.so from the same sources with extra -D switches (measurement builds, see TDOA_LIB_VARIANT)"""
    out = os.path.join(PKG_ROOT, "libtdoa_mi355x_%s.so" % name)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-pthread", "-Xarch_host", "-ffp-contract=off"] + ["-D" + d for d in defines] + ["-o", out, os.path.join(CSRC, "tdoa_mi355x.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_cli(force=False, verbose=False):
    """tdoa_processor: C++ host harness with the reference processor's command line, linked
    against the C-ABI library only (no HIP headers)."""
    if (not force and os.path.exists(CLI) and os.path.getmtime(CLI) >= max(os.path.getmtime(CLI_SRC), os.path.getmtime(HEADER))):
        return CLI
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-Wall", "-o", CLI, CLI_SRC,
           "-L" + PKG_ROOT, "-ltdoa_mi355x", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # build.py --variant dec14 TDOA_DEC_STEPS=14
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=True))
    else:
        print(build(force=True, verbose=True))
