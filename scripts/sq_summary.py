#!/usr/bin/env python3
"""Per-kernel table of the SQ counters collected by collect_sq.sh (summed over the launches of the pass).
usage: sq_summary.py <dir prefix> <n groups>"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (hash of the kernel sources the counters were taken on)


def main():
    prefix, n = sys.argv[1], int(sys.argv[2])
    table = collections.defaultdict(dict)
    for i in range(n):
        for path in glob.glob("%s%d/**/*counter_collection.csv" % (prefix, i), recursive=True):
            for r in csv.DictReader(open(path)):
                # template instances of one kernel are summed under its plain name (their argument lists hold commas)
                name = r["Kernel_Name"].split("(")[0].replace("tdoa::", "").replace("void ", "").split("<")[0]
                if name.startswith("k_") and not name.startswith("k_synth"):
                    # summed over every launch of the pass (a step may take several launch groups; every ratio the tables are
                    # read for -- issue share, instructions per wave, conflict share -- is a ratio of two such sums)
                    table[name][r["Counter_Name"]] = table[name].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ctrs = sorted({c for v in table.values() for c in v})
    print("# source_sha16=" + source_hash())
    print("kernel," + ",".join(ctrs))
    for k in sorted(table):
        print(k + "," + ",".join("%.6g" % table[k].get(c, float("nan")) for c in ctrs))


if __name__ == "__main__":
    main()
