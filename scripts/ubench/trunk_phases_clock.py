#!/usr/bin/env python3
"""Per-wave phase stamps of trunk_phases_kernel (LSTM forward -> LayerNorm / heads / loss -> LSTM backward -> MLP backward in
one launch) inside real PPO iterations (debug build: scripts/ab_build.sh splitt "-DSPLIT_TIMING", then
VINE_HIP_LIB=build/libvine_splitt.so python scripts/ubench/trunk_phases_clock.py [overrides]).  Prints the stamps of the last
launch of the last iteration: time since the first wave's entry and shader cycles per phase."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd import load_config, native
from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map

cfg = load_config(overrides=["num_envs=16384", "task.env.CREATE_PIPE=False"] + sys.argv[1:])
env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0", graphics_device_id=0,
                                              headless=True)
params = cfg["train"]["params"]
params["config"].update(write_files=False, print_stats=False, use_graphs=True)
agent = A2CAgent("t", params, vec_env=env)
agent.init_tensors()
agent.obs = agent.env_reset()["obs"]
lib = native.load()
lib.vine_debug_mlp_split_timing.argtypes = [C.c_void_p]
for it in range(6):
    agent.train_epoch()
torch.cuda.synchronize()
buf = (C.c_uint64 * (4096 * 16))()
assert lib.vine_debug_mlp_split_timing(buf) == 0
t8 = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8, 2).astype(np.int64)[:2048]
t = t8[:, :5]
t8 = t8[t8[:, 0, 1] > 0]
t = t8[:, :5]
clk = (t[:, 4, 0] - t[:, 0, 0]) / np.maximum(t[:, 4, 1] - t[:, 0, 1], 1) * 100.0
rel = (t[:, :, 1] - t[:, :, 1].min()) / 100.0
print("%d waves stamped; in-kernel clock %.0f MHz (median)" % (len(t), np.median(clk)))
names = ("entry", "LSTM forward done", "LayerNorm + heads + loss done", "LSTM backward done", "MLP backward done (end)")
cyc = t[:, :, 0] - t[:, 0:1, 0]
for i, nm in enumerate(names):
    print("   %-30s at mean %7.2f us (min %7.2f max %7.2f)   phase: %6.2f us mean, %7.0f cycles"
          % (nm, rel[:, i].mean(), rel[:, i].min(), rel[:, i].max(), (rel[:, i] - rel[:, i - 1]).mean() if i else 0.0,
             (cyc[:, i] - cyc[:, i - 1]).mean() if i else 0.0))
# inside the loss phase: stamps 5 (LayerNorm statistics + heads done), 6 (loss terms done), 7 (backward + dx stores done)
sub = np.stack([t8[:, 1], t8[:, 5], t8[:, 6], t8[:, 7], t8[:, 2]], axis=1)
for i, nm in enumerate(("", "rows loaded, LayerNorm statistics + heads", "loss terms (16 lanes)", "backward + dx stores", "row folds, LDS reduction, partial rows")):
    if i:
        print("      loss phase / %-42s %6.2f us mean, %7.0f cycles" % (nm, ((sub[:, i, 1] - sub[:, i - 1, 1]) / 100.0).mean(),
                                                                     (sub[:, i, 0] - sub[:, i - 1, 0]).mean()))
