"""The two persistent LSTM sequence kernels of the mixed-precision update at its shapes, on cold data (rotating operand
sets > L2 + MALL), for `rocprofv3 --kernel-trace --pmc ...` passes (HBM traffic per launch vs algorithmic bytes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.learning import bench_support

res = bench_support.ppo_kernel_rooflines(torch.device("cuda:0"))
for r in res:
    print(r)
