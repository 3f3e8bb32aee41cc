#!/usr/bin/env python3
"""Per-kernel means of the counters collected by scripts/profile_ppo_pmc.sh:  python scripts/pmc_kernels.py <dir>"""
import csv
import glob
import os
import sys

root = sys.argv[1]
data = {}      # kernel -> counter -> {dispatch: value}
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0][:90]
        data.setdefault(k, {}).setdefault(r["Counter_Name"], {})
        d = data[k][r["Counter_Name"]]
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
names = sorted({c for k in data.values() for c in k})
print("kernel, launches, " + ", ".join(names))
for k, cs in sorted(data.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", {0: 0}).values())):
    n = max(len(v) for v in cs.values())
    print("%s, %d, %s" % (k, n, ", ".join("%.4g" % (sum(cs[c].values()) / len(cs[c])) if c in cs else "-" for c in names)))
