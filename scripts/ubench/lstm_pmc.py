"""The two LSTM kernels of the mixed-precision update at its shapes, 30 launches each on cold-ish data (8 rotating
operand sets > L2 + MALL), for `rocprofv3 --kernel-trace --pmc ...` passes (HBM traffic per launch vs algorithmic bytes)."""
import torch
from vine_robot_isaacgymenvs_amd.learning import bench_support

res = bench_support.ppo_kernel_rooflines(torch.device("cuda:0"))
for r in res:
    print(r)
