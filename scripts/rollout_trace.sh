#!/bin/bash
# kernel trace of a few graph-replayed PPO iterations, rollout kernels only:  scripts/rollout_trace.sh <tag> [bench args]
TAG=${1:-x}; shift
OUT=gpurun_out/rtrace_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --mode ppo --steps 8 --warmup 2 --no-cpu-baseline --no-saturated --no-secondary --no-other-configs "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/trace/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:70]
    if any(k in n for k in ('vine_step', 'policy_head', 'rollout_post', 'mlp3_elu_split', 'lstm_step', 'copy_batched', 'head_prep')):
        print('%-72s calls %5s avg %7.2f us' % (n, r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $OUT/trace
