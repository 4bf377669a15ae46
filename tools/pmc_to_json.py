#!/usr/bin/env python3
"""Collect the rocprofv3 outputs of tools/round_profile.sh into profiles/<tag>_pmc_<workload>.json (what bench.py's
`roofline` reads) and profiles/<tag>_kernel_stats_<config>.csv.

    python3 tools/pmc_to_json.py <tag> <config> [bench.py arguments of the workload]

Per-launch values = sum of a counter over the dispatches of the dominant kernel / number of dispatches, from the
separate --pmc passes gpurun_out/pmc_<config>_{sq,fetch,write,tatd}; FETCH_SIZE / WRITE_SIZE are stored as rocprofv3
reports them (KB, uncorrected: bench.py doubles FETCH_SIZE as MI355X_MICROARCH.md prescribes for gfx950)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, cfg = sys.argv[1], sys.argv[2]
sys.argv = ["bench.py", "--config", cfg] + sys.argv[3:]
import bench  # noqa: E402

args = bench.parse_args()
key = bench.workload_key(args)
kernel = {"megakernel": "rt_render_paths_kernel<false", "lockstep": "rt_render_kernel<false", "wavefront": "rt_trace_kernel<false"}[args.mode]
tot, cnt, names = collections.defaultdict(float), collections.defaultdict(int), set()
for p in ("sq", "fetch", "write", "tatd"):
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{cfg}_{p}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[row["Counter_Name"]] += 1
                names.add(row["Kernel_Name"])
per_launch = {k: tot[k] / cnt[k] for k in sorted(tot)}
avg_ms = None
for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{cfg}", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{cfg}.csv"))
    for row in csv.DictReader(open(f)):
        if kernel in row["Name"]:
            avg_ms = float(row["AverageNs"]) / 1e6
try:
    head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    head = os.environ.get("RT_GIT_HEAD")
out = {"workload_key": key, "config": cfg, "kernel": sorted(names), "dispatches_per_counter": dict(cnt),
       "per_launch": per_launch, "kernel_trace_avg_ms": avg_ms, "source_sha": bench.source_sha(), "git_head": head,
       "date": time.strftime("%Y-%m-%d %H:%M:%S"),
       "how": "rocprofv3 --pmc, one pass per counter block (tools/round_profile.sh), bench.py --steps 2 --warmup 0; "
              "kernel_trace_avg_ms from rocprofv3 --kernel-trace --stats of the default bench command"}
path = os.path.join(ROOT, "profiles", f"{tag}_pmc_{key}.json")
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps(per_launch))
