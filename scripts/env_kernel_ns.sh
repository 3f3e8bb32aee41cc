#!/bin/bash
# average duration of the env-step kernel from a rocprofv3 kernel trace of `bench.py --mode env` (the HIP-event timing of
# back-to-back launches is host-bound below ~13 us per launch):  scripts/env_kernel_ns.sh <tag> [bench args]
TAG=$1; shift
OUT=gpurun_out/envns_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --mode env --steps 200 --warmup 60 --no-cpu-baseline --no-saturated --no-other-configs "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/trace/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'vine_step' in r['Name']:
        print('$TAG', r['Name'][28:70], 'avg %.2f us  min %.2f us' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
rm -rf $OUT
