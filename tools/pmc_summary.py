#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per-kernel sums of each counter (render kernel only by default)."""
import csv, glob, sys, collections
root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else "rt_render_kernel<false>"
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if match in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k}: total {tot[k]:.6g} over {n[k]} dispatches -> per dispatch {tot[k]/n[k]:.6g}")
