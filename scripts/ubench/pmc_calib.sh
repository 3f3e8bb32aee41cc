#!/bin/bash
# run on the GPU box: builds the calibration binary and collects FETCH_SIZE / WRITE_SIZE in separate passes
set -e
cd "$(dirname "$0")"
OUT=${1:-../../gpurun_out/pmc_calib}
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $OUT/pmc_calib pmc_calib.hip
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- $OUT/pmc_calib > $OUT/run_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -- $OUT/pmc_calib > $OUT/run_w.log 2>&1
python3 - <<PY
import csv, glob
for tag in ("f", "w"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Grid_Size"], r["Counter_Name"])
        acc.setdefault(k, []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-62s grid %-9s %-10s mean %12.1f KiB  (n=%d)" % (k[0], k[1], k[2], sum(v) / len(v), len(v)))
PY
rm -f $OUT/pmc_calib
