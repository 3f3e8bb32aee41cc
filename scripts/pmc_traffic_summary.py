#!/usr/bin/env python3
"""Per-kernel HBM-side traffic of real PPO iterations, from the passes of scripts/profile_ppo_traffic.sh:

    python scripts/pmc_traffic_summary.py gpurun_out/ppo_traffic_<tag>     (writes <dir>/pmc_summary.json, prints a table)

Per kernel instantiation: launches seen in the counter passes, mean FETCH_SIZE / WRITE_SIZE per launch (KiB, as rocprofv3
reports them), `traffic_bytes` = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM section: on gfx950
FETCH_SIZE tallies a 128-B read request at 64 B; WRITE_SIZE is exact) and the in-situ duration of the same kernel in the
graph-replayed trace of the same tree (`avg_ns`, `calls`).  bench.py's `ppo_kernel_rooflines` reads this file
(`profiles/rNN/ppo_traffic_*_pmc_summary.json`)."""
import csv
import glob
import json
import os
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


def main():
    root = sys.argv[1]
    ctr = {}
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), r["Counter_Name"], r["Dispatch_Id"])
            acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
        for (k, c, _d), v in acc.items():
            ctr.setdefault(k, {}).setdefault(c, []).append(v)
    stats = {}
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                       "total_ns": float(r["TotalDurationNs"]), "pct": float(r["Percentage"])}
        dst = os.path.join(root, "kernel_stats.csv")
        if not os.path.exists(dst):
            import shutil
            shutil.copy(f, dst)
    out = {}
    for k, st in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"]):
        row = dict(st)
        c = ctr.get(k, {})
        if "FETCH_SIZE" in c:
            row["FETCH_SIZE_KiB"] = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
            row["pmc_launches"] = len(c["FETCH_SIZE"])
        if "WRITE_SIZE" in c:
            row["WRITE_SIZE_KiB"] = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        if "FETCH_SIZE_KiB" in row and "WRITE_SIZE_KiB" in row:
            row["traffic_bytes"] = (2.0 * row["FETCH_SIZE_KiB"] + row["WRITE_SIZE_KiB"]) * 1024.0
            row["traffic_GBs_in_situ"] = row["traffic_bytes"] / row["avg_ns"]
            row["hbm_frac_of_8TBs"] = row["traffic_GBs_in_situ"] / 8000.0
        out[k] = row
    json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
    tot = sum(s["total_ns"] for s in stats.values())
    print("total kernel ms %.3f" % (tot / 1e6))
    print("%-60s %6s %9s %6s %11s %11s %10s %7s" % ("kernel", "calls", "avg us", "%", "fetch KiB", "write KiB", "MB (2F+W)", "TB/s"))
    for k, r in list(out.items())[:30]:
        print("%-60s %6d %9.1f %6.1f %11s %11s %10s %7s" % (
            k[:60], r["calls"], r["avg_ns"] / 1e3, r["pct"],
            "%.0f" % r["FETCH_SIZE_KiB"] if "FETCH_SIZE_KiB" in r else "-",
            "%.0f" % r["WRITE_SIZE_KiB"] if "WRITE_SIZE_KiB" in r else "-",
            "%.1f" % (r["traffic_bytes"] / 1e6) if "traffic_bytes" in r else "-",
            "%.2f" % (r["traffic_GBs_in_situ"] / 1e3) if "traffic_bytes" in r else "-"))


if __name__ == "__main__":
    main()
