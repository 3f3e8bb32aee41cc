#!/usr/bin/env python3
"""Rollout time under a TRAINED policy with every seed fixed: two builds whose trajectories are bit-identical (exact savings in the
contact code) then see the same workload.   VINE_HIP_LIB=<lib> python scripts/ubench/obstacle_trained_ab.py [overrides]
Trains TRAIN_ITERS (100) iterations on the default task config (pipe), then reports the device time of the rollouts and updates of
the next 40 iterations and a checksum of the parameters (equal across builds = same trajectory)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd import load_config
from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
from vine_robot_isaacgymenvs_amd.utils.utils import set_seed

iters = int(os.environ.get("TRAIN_ITERS", "100"))
set_seed(42)
cfg = load_config(overrides=["num_envs=16384"] + sys.argv[1:])
cfg["task"]["seed"] = 42
env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0", graphics_device_id=0,
                                              headless=True)
params = cfg["train"]["params"]
params["config"].update(write_files=False, print_stats=False, use_graphs=True, sync_each_iteration=True)
torch.manual_seed(0)
agent = A2CAgent("t", params, vec_env=env)
agent.init_tensors()
agent.obs = agent.env_reset()["obs"]
for _ in range(iters):
    agent.train_epoch()
play = upd = 0.0
for _ in range(40):
    p, u, _s = agent.train_epoch()
    play += p
    upd += u
torch.cuda.synchronize()
chk = float(agent.optimizer.flat_params.double().sum())
rew = float(agent.game_rewards.get_mean()[0]) if float(agent.game_rewards.current_size) > 0 else float("nan")
print("rollout %.3f ms  update %.3f ms  (iterations %d-%d)  mean return %.1f  parameter checksum %.10e"
      % (play / 40 * 1e3, upd / 40 * 1e3, iters + 1, iters + 40, rew, chk))
