"""Kernel-level comparison for rocprofv3: the matrix-core weight-gradient kernel vs split-K bmm + column sums, per
weight shape of the default network (run under `rocprofv3 --kernel-trace --stats`)."""
import sys
import torch
from vine_robot_isaacgymenvs_amd.learning import fused

fused.WGRAD_MAX_OUT = 1 << 30
if len(sys.argv) > 1:
    fused.WGRAD_WGS = int(sys.argv[1])
dev, bf, n = "cuda", torch.bfloat16, 32768
xfull = (torch.randn(n, 96, device=dev) * 0.5).to(bf)
cases = [("W1", 256, xfull[:, 64:90]), ("W2", 128, torch.randn(n, 256, device=dev).to(bf)),
         ("W3", 64, torch.randn(n, 128, device=dev).to(bf)), ("w_ih", 1024, xfull[:, :90]),
         ("w_hh", 1024, torch.randn(n, 256, device=dev).to(bf))]
for name, M, x in cases:
    dy = (torch.randn(n, M, device=dev) * 0.1).to(bf)
    print(name, fused._wgrad_plan(dy, x))
    for _ in range(20):
        fused.weight_grad(dy, x)
        fused.splitk_tn(dy, x)
    torch.cuda.synchronize()
