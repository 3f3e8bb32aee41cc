#!/bin/bash
# rocprofv3 kernel stats of whole PPO iterations (eager rollout so that every kernel is visible).
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_ppo_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --mode ppo --steps 3 --warmup 2 --no-graph --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/trace/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:28]:
    print('%8.2f ms %6s calls %8.1f us avg  %5.1f%%  %s' % (float(r['TotalDurationNs'])/1e6, r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage']), r['Name'][:110]))
PY
tail -2 $OUT/bench.log | cut -c1-400
