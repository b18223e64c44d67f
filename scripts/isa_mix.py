#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels (no GPU needed): compiles tdoa_mi355x.hip to assembly and counts, per
kernel, the vector / packed-vector / LDS / global / scratch instructions.   usage: scripts/isa_mix.py [regex]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tdoa-geolocation_amd", "csrc", "tdoa_mi355x.hip")
OUT = "/tmp/tdoa_isa.s"


def main():
    pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           "-o", OUT, SRC], stderr=subprocess.DEVNULL)
    txt = open(OUT).read()
    for f in re.split(r"\n(?=_Z\w+:)", txt):
        m = re.match(r"(_Z\w+):", f)
        if not m or "k_" not in m.group(1) or not pat.search(m.group(1)):
            continue
        body = f.split(".end_amdhsa_kernel")[0]
        ins = []
        for line in body.split("\n"):
            s = line.strip()
            if not line.startswith("\t") or not s or s[0] in ".;":
                continue
            ins.append(s.split()[0])
        c = collections.Counter(ins)
        tot = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        print("%-100s valu %5d pk %4d ds %4d glob %4d scratch %3d salu %4d" % (
            m.group(1)[:100], tot("v_"), tot("v_pk_"), tot("ds_"), tot("global_") + tot("buffer_") + tot("flat_"),
            tot("scratch_"), tot("s_")))


if __name__ == "__main__":
    main()
