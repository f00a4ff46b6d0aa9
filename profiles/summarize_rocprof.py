#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output to small per-kernel summaries (kept under profiles/).

usage: summarize_rocprof.py <rocprof output dir> <out prefix>
 - *_kernel_stats.csv is copied as is (already a per-kernel summary)
 - *_counter_collection.csv (--pmc runs) is reduced to per-kernel mean/sum per counter
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def main():
    src, out = sys.argv[1], sys.argv[2]
    for f in glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, out + "_kernel_stats.csv")
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]
                c = row.get("Counter_Name", "?")
                v = float(row.get("Counter_Value", 0) or 0)
                a = agg[k][c]
                a[0] += 1
                a[1] += v
    if agg:
        res = {k: {c: {"dispatches": a[0], "sum": a[1], "mean_per_dispatch": a[1] / max(a[0], 1)}
                   for c, a in cs.items()} for k, cs in agg.items()}
        json.dump(res, open(out + "_pmc.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
