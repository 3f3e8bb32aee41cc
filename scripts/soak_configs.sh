#!/bin/bash
# Long-run stability of the final tree in OTHER configurations than scripts/soak_compare.sh's default (16384 envs, pipe):
# 1500 PPO iterations each; prints one line per run.  usage (GPU box): bash scripts/soak_configs.sh
cd "$(dirname "$0")/../gpurun_out" || exit 1
run() {      # name seed extra-overrides...
  name=$1; seed=$2; shift 2
  rm -rf runs
  python3 ../train.py task=Vine5LinkMovingBase max_iterations=1500 headless=True experiment=soak seed=$seed \
    train.params.config.env_stats_every=0 train.params.config.save_frequency=0 "$@" > soak_cfg_${name}_$seed.log 2>&1
  python3 - <<PY
import csv, math
rows=[r for r in csv.reader(open("runs/soak/summaries/scalars.csv"))]
rew=[(int(r[2]), float(r[1])) for r in rows if r[0]=="rewards/iter"]
print("$name seed=$seed last", round(rew[-1][1]), "min after 100:", round(min(v for i,v in rew if i>100)), "iters<900:", sum(1 for i,v in rew if i>100 and v<900), "mean after 100:", round(sum(v for i,v in rew if i>100)/sum(1 for i,v in rew if i>100),1), "non-finite", sum(1 for r in rows if not math.isfinite(float(r[1]))), flush=True)
PY
}
for s in 7 11; do run free16384 $s num_envs=16384 task.env.CREATE_PIPE=False; done
for s in 7 11; do run shelf_delay1 $s num_envs=16384 task.env.CREATE_PIPE=False task.env.CREATE_SHELF=True task.env.ACTION_DELAY=1; done
for s in 7 11; do run envs4096 $s num_envs=4096; done
for s in 7; do run fp32update $s num_envs=16384 train.params.config.mixed_precision=False max_iterations=400; done
rm -rf runs
