#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by scripts/profile_env.sh for one kernel.

    python scripts/pmc_summary.py gpurun_out/prof_<tag> [kernel-substring] > profiles/rNN/env_step_<tag>_pmc_summary.json

Per counter: number of launches seen and the mean counter value per launch (FETCH_SIZE / WRITE_SIZE in KiB as
rocprofv3 reports them; bench.py multiplies by 1024).  Also copies nothing: the kernel-stats CSV of the trace pass is
what `profiles/` keeps beside this JSON.
"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "vine_step_kernel"
out = {}
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        if needle not in r["Kernel_Name"]:
            continue
        key = (r["Counter_Name"], r["Dispatch_Id"])
        acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
    per = {}
    for (name, _d), v in acc.items():
        per.setdefault(name, []).append(v)
    for name, vals in per.items():
        out[name] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals)}
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if needle in r["Name"]:
            out["kernel_stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                   "max_ns": float(r["MaxNs"]), "name": r["Name"][:80]}
print(json.dumps(out, indent=1))
