#!/bin/bash
# Free-space target reaching (reference README.md:63, BASELINE.json configs[1]): 4096 envs, 300 PPO iterations.
# usage: scripts/train_free_space.sh <experiment-name> [extra overrides...]
NAME=$1; shift
python3 "$(dirname "$0")/../train.py" task=Vine5LinkMovingBase num_envs=4096 max_iterations=300 headless=True experiment=$NAME \
  vine_randomize=False task.env.CREATE_PIPE=False task.env.OBSERVATION_TYPE=TIP_AND_CART_AND_OBJ_INFO \
  task.env.maxEpisodeLength=100 task.env.SUCCESS_DIST=0.04 task.env.MIN_TARGET_Y=-0.4 task.env.MAX_TARGET_Y=0.4 \
  task.env.MIN_TARGET_Z=0.55 task.env.MAX_TARGET_Z=0.7 RAIL_SOFT_LIMIT=0.25 RAIL_P_GAIN=30 RAIL_ACCELERATION=6 \
  train.params.config.env_stats_every=0 "$@"
