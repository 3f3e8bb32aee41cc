#!/usr/bin/env python3
"""Tabulate hipcc's -Rpass-analysis=kernel-resource-usage remarks:  python scripts/kernel_resources.py remarks.txt [name-substring]
(name, VGPRs, AGPRs, SGPRs, VGPR spills, SGPR spills, scratch bytes per lane, occupancy, LDS bytes)."""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
needle = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for blk in text.split("Function Name: ")[1:]:
    name = blk.split()[0]
    def f(key):
        m = re.search(key + r": (\d+)", blk)
        return int(m.group(1)) if m else -1
    rows.append((name, f("VGPRs"), f("AGPRs"), f("TotalSGPRs"), f("VGPRs Spill"), f("SGPRs Spill"), f(r"ScratchSize \[bytes/lane\]"),
                 f(r"Occupancy \[waves/SIMD\]"), f(r"LDS Size \[bytes/block\]")))
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
except OSError:
    dem = [r[0] for r in rows]
print("%-90s %5s %5s %5s %6s %6s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "vspill", "sspill", "scratch", "occ", "LDS"))
for r, d in zip(rows, dem):
    d = d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if needle in d:
        print("%-90s %5d %5d %5d %6d %6d %7d %4d %7d" % ((d[:90],) + r[1:]))
