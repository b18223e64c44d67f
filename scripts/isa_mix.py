#!/usr/bin/env python3
"""Instruction mix of the hot kernels from a --save-temps gfx950 .s file (static counts, whole kernel body):
   python3 scripts/isa_mix.py /tmp/isa/tdoa_mi355x-hip-amdgcn-amd-amdhsa-gfx950.s [name-substring ...]"""
import collections
import re
import sys

HOT = ("k_fwd_col256_k1ILb0ELb1", "k_pair_decimate16", "k_fwd_row4096_unpack", "k_inv_rows_plain_r8", "k_small_col_peak", "k_once_edges")


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2:] or HOT
    parts = re.split(r"\n(_Z\w+): *;[^\n]*\n", txt)
    for i in range(1, len(parts), 2):
        name, body = parts[i], re.split(r"\n\.Lfunc_end\d+:", parts[i + 1])[0]
        if not any(k in name for k in want):
            continue
        c, top = collections.Counter(), collections.Counter()
        for line in body.split("\n"):
            t = line.strip()
            if not line.startswith("\t") or not t or t[0] in ".;":
                continue
            op = t.split()[0]
            top[op] += 1
            c["v_pk" if op.startswith("v_pk_") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
              "salu" if op.startswith("s_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"] += 1
        print(name[:90])
        print("   ", dict(c))
        print("   ", ", ".join("%s %d" % kv for kv in top.most_common(24)))


if __name__ == "__main__":
    main()
