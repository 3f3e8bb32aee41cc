#!/bin/bash
# End-to-end sanity of the training loop on the task YAML's defaults (pipe) and in free space: 150 PPO iterations each at 16384 envs;
# prints throughput / final return and copies the scalars.   usage (GPU box): bash scripts/train_sanity.sh <tag>
TAG=${1:-x}
cd "$(dirname "$0")/../gpurun_out" || exit 1
for name in default nopipe; do
  extra=""; [ $name = nopipe ] && extra="task.env.CREATE_PIPE=False"
  rm -rf runs/sanity_$name
  python3 ../train.py task=Vine5LinkMovingBase num_envs=16384 max_iterations=150 headless=True experiment=sanity_$name \
    train.params.config.save_frequency=0 $extra > train_${TAG}_$name.log 2>&1 || { echo "train.py failed ($name)"; tail -3 train_${TAG}_$name.log; exit 1; }
  cp runs/sanity_$name/summaries/scalars.csv train_${TAG}_${name}_16384envs_150iters.csv
  echo -n "$name: "; python3 ../scripts/train_fps.py train_${TAG}_${name}_16384envs_150iters.csv
  echo "   NaN lines: $(grep -ci nan train_${TAG}_${name}_16384envs_150iters.csv)"
done
