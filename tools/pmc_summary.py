#!/usr/bin/env python
"""Per-kernel mean of rocprofv3 --pmc counters from the CSV output (`--output-format csv`):
   python tools/pmc_summary.py <dir-with-*_counter_collection.csv> [kernel-substring]"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    root, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
            if sub in name:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        print(name)
        for c, v in sorted(cs.items()):
            print(f"    {c:36s} n={len(v):5d} mean={sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main()
