OUT=gpurun_out/prof_ppo_j
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --mode ppo --steps 4 --warmup 3 --no-cpu-baseline --no-saturated --no-secondary > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/trace/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6, 'per iteration (7 iterations)', tot/1e6/7)
for r in rows[:32]:
    print('%8.2f ms %6s calls %8.1f us avg  %5.1f%%  %s' % (float(r['TotalDurationNs'])/1e6, r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage']), r['Name'][:100]))
PY
tail -1 $OUT/bench.log | cut -c1-300
