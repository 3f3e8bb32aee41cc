"""The grouped launch of the three MLP weight gradients at the update's shape (32768 samples), on operand sets in rotation
(cold), HIP events: python scripts/ubench/wgrad_group_bench.py  (VINE_HIP_LIB selects an A/B build)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.learning import fused

dev, bf, n, SETS = torch.device("cuda:0"), fused.lp_dtype(), 32768, 6
torch.manual_seed(0)
sets = []
for _ in range(SETS):
    xfull = (torch.randn(n, 96, device=dev) * 0.5).to(bf)
    a1, a2 = torch.randn(n, 256, device=dev).to(bf), torch.randn(n, 128, device=dev).to(bf)
    gz = [(torch.randn(n, c, device=dev) * 0.1).to(bf) for c in (256, 128, 64)]
    sets.append((gz, [xfull[:, 64:92], a1, a2]))
outs = [torch.empty(256, 28, device=dev), torch.empty(128, 256, device=dev), torch.empty(64, 128, device=dev)]


def run(i):
    gz, xs = sets[i % SETS]
    batch = fused.ColumnSumBatch()
    grp = fused.WeightGradGroup()
    for g, x, o in zip(gz, xs, outs):
        assert grp.add(g, x, o, batch)
    grp.flush()
    return batch


for i in range(SETS):
    run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
keep = []
e0.record()
for i in range(60):
    keep.append(run(i))
e1.record()
torch.cuda.synchronize()
print("grouped MLP weight gradients (launch only, no column sums): %.1f us" % (e0.elapsed_time(e1) / 60 * 1e3))
