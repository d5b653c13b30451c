#!/usr/bin/env python3
"""Condense a tools/prof_moe.sh output directory: per-kernel stats + per-kernel mean of each PMC counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12:
            print("  ", ",".join(c[:60] for c in row))
for pdir in sorted(glob.glob(os.path.join(d, "pmc_*"))):
    for f in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")[:40]
            acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
        print("== pmc:", os.path.basename(pdir))
        for k, cs in acc.items():
            if "gemm" in k or "act_" in k or "gemv" in k:
                print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
