"""Per-kernel averages of every counter found under one or more rocprofv3 --pmc output dirs.
usage: python tools/pmc_show.py DIR [DIR ...] [--kernel SUBSTR]"""
import collections
import csv
import glob
import os
import sys

import re


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]


def main():
    args = sys.argv[1:]
    filt = None
    if "--kernel" in args:
        i = args.index("--kernel")
        filt = args[i + 1]
        del args[i:i + 2]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if filt and filt not in k:
                    continue
                c = agg[k][r["Counter_Name"]]
                c[0] += float(r["Counter_Value"])
                c[1] += 1
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            s, n = agg[k][c]
            print(f"    {c:28s} {s / n:16.1f}  (n={n})")


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
