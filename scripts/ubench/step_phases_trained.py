"""Per-wave stamps of the four-lane step kernel under a TRAINED policy (default task config: pipe obstacle): trains the
default agent for TRAIN_ITERS iterations, then prints the stamp statistics of the last env-step launch (debug build:
scripts/ab_build.sh timing "-DVSQ_TIMING"; VINE_HIP_LIB=build/libvine_timing.so python scripts/ubench/step_phases_trained.py)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd import load_config, native
from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map

iters = int(os.environ.get("TRAIN_ITERS", "120"))
cfg = load_config(overrides=["num_envs=16384"] + sys.argv[1:])
env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0", graphics_device_id=0,
                                              headless=True)
params = cfg["train"]["params"]
params["config"].update(write_files=False, print_stats=False)
agent = A2CAgent("t", params, vec_env=env)
agent.init_tensors()
agent.obs = agent.env_reset()["obs"]
lib = native.load()
lib.vine_debug_timing.argtypes = [C.c_void_p]
names = ["entry", "loads issued", "(unused)", "RNG done", "loads back", "iterations done", "post + obs done", "end"]
for it in range(1, iters + 1):
    agent.train_epoch()
    if it in (1, iters // 2, iters):
        torch.cuda.synchronize()
        buf = (C.c_uint64 * (1024 * 8))()
        assert lib.vine_debug_timing(buf) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
        t0 = t[:, 0].min()
        d5, d7 = (t[:, 5] - t0) * 0.01, (t[:, 7] - t0) * 0.01
        print("iteration %3d  mean reward %8.2f  iterations done: mean %6.1f us  p50 %6.1f  p90 %6.1f  max %6.1f   end: mean %6.1f max %6.1f"
              % (it, float(agent.game_rewards.get_mean()[0]) if float(agent.game_rewards.current_size) > 0 else float("nan"),
                 d5.mean(), np.percentile(d5, 50), np.percentile(d5, 90), d5.max(), d7.mean(), d7.max()), flush=True)
