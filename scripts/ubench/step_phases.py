"""Per-phase time stamps of the four-lane env-step kernel (debug build: scripts/ab_build.sh timing "-DVSQ_TIMING", then
VINE_HIP_LIB=build/libvine_timing.so python scripts/ubench/step_phases.py [bench-style overrides]).  wall_clock64() runs at
100 MHz: 10 ns resolution.  Prints, over the 1024 waves of the last launch, the mean / max time at each stamp relative
to the earliest wave's start."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd import load_config, native
from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map

N = int(os.environ.get("STEP_N", "16384"))
cfg = load_config(overrides=["num_envs=%d" % N] + sys.argv[1:])
env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0", graphics_device_id=0,
                                              headless=True)
env.reset()
act = torch.rand(N, 2, device="cuda:0") * 2 - 1
for _ in range(200):
    env.step(act)
torch.cuda.synchronize()
lib = native.load()
buf = (C.c_uint64 * (1024 * 8))()
lib.vine_debug_timing.argtypes = [C.c_void_p]
assert lib.vine_debug_timing(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)[:N * 4 // 64]
t0 = t[:, 0].min()
names = ["entry", "loads issued", "constants", "RNG done", "loads back", "iterations done", "post + obs done", "end"]
for i, nm in enumerate(names):
    d = (t[:, i] - t0) * 10.0 / 1e3
    print("%-18s mean %6.2f us   max %6.2f us   (min %6.2f)" % (nm, d.mean(), d.max(), d.min()))
