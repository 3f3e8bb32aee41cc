#!/bin/bash
# rocprofv3 kernel trace of the PRODUCTION path (rollout and optimiser steps replayed as hipGraphs).
# Usage: scripts/profile_ppo_graphed.sh <tag> [extra bench args]   -> gpurun_out/prof_ppo_<tag>/{kernel_stats.csv,summary.txt}
TAG=${1:-x}; shift
OUT=gpurun_out/prof_ppo_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --mode ppo --steps 8 --warmup 2 --no-cpu-baseline --no-saturated --no-secondary "$@" > $OUT/bench.log 2>&1
python3 - <<PY > $OUT/summary.txt
import csv, glob, shutil
f = glob.glob('$OUT/trace/*/*_kernel_stats.csv')[0]
shutil.copy(f, '$OUT/kernel_stats.csv')
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6, 'per iteration (10 iterations)', tot/1e6/10)
for r in rows[:36]:
    print('%8.2f ms %6s calls %8.1f us avg  %5.1f%%  %s' % (float(r['TotalDurationNs'])/1e6, r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage']), r['Name'][:110]))
PY
cat $OUT/summary.txt
tail -1 $OUT/bench.log | cut -c1-400
