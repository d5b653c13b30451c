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

# ---- launch gaps inside a step (kernel-trace timestamps): pre-pass end -> GEMM start, GEMM end -> next pre-pass start
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if ("gemm_i8" in r.get("Kernel_Name", "") or "gemm_w4" in r.get("Kernel_Name", "") or "act_fused" in r.get("Kernel_Name", "") or "act_f8" in r.get("Kernel_Name", ""))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    g1, g2, ka, kg = [], [], [], []
    for a, b in zip(rows, rows[1:]):
        gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        if "act_" in a["Kernel_Name"] and "gemm" in b["Kernel_Name"]:
            g1.append(gap); ka.append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
            kg.append((int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3)
        elif "gemm" in a["Kernel_Name"] and "act_" in b["Kernel_Name"] and gap < 200:
            g2.append(gap)
    med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
    print(f"== step timeline (us, medians over {len(g1)} steps): pre-pass {med(ka):.1f} | gap {med(g1):.1f} | GEMM {med(kg):.1f} | gap to the next step {med(g2):.1f}")
