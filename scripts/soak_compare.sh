#!/bin/bash
# Long-run stability of the default configuration: 1500 PPO iterations per (precision, seed); prints one line each.
# usage (GPU box): bash scripts/soak_compare.sh "True 7" "False 7" ...
cd "$(dirname "$0")/../gpurun_out" || exit 1
for cfg in "$@"; do
  set -- $cfg
  rm -rf runs
  python3 ../train.py task=Vine5LinkMovingBase num_envs=16384 max_iterations=1500 headless=True experiment=soak seed=$2 \
    train.params.config.env_stats_every=0 train.params.config.save_frequency=0 train.params.config.mixed_precision=$1 > soak_$1_$2.log 2>&1
  python3 - <<PY
import csv, math
rows=[r for r in csv.reader(open("runs/soak/summaries/scalars.csv"))]
rew=[(int(r[2]), float(r[1])) for r in rows if r[0]=="rewards/iter"]
low=[(i,round(v)) for i,v in rew if i>100 and v<900]
print("mixed=$1 seed=$2 last", round(rew[-1][1]), "min after 100:", round(min(v for i,v in rew if i>100)), "iters<900:", len(low), "mean after 100:", round(sum(v for i,v in rew if i>100)/sum(1 for i,v in rew if i>100),1), "non-finite", sum(1 for r in rows if not math.isfinite(float(r[1]))), flush=True)
PY
done
rm -rf runs
