"""LSTM weight gradients at the update's shape: the two-copy operand (masked h_{t-1} tensor) against the "h once" form
(vine_weight_grad_cat_seq_mfma shifts / masks the one copy of the hidden states itself)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.learning import fused

dev = torch.device("cuda:0")
bf = fused.lp_dtype()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, H, width = 32768, 256, 92
B, M = n // T, 4 * H
torch.manual_seed(0)
xfull = (torch.randn(n, 96, device=dev) * 0.5).to(bf)
x1 = xfull[:, :width]
h_all = torch.randn(B * (T + 1), H, device=dev).to(bf)      # [B, T + 1, H]: slot 0 = h0, slot t + 1 = h_t
dones = (torch.rand(n, device=dev) < 0.25).to(torch.uint8)
hp = fused.masked_previous_hidden(h_all, dones, T)
dG = (torch.randn(n, M, device=dev) * 0.1).to(bf)
o1, o2 = torch.empty(M, width, device=dev), torch.empty(M, H, device=dev)
for name, f in (("masked tensor", lambda: fused.weight_grad_cat(dG, x1, hp, o1, o2)),
                ("h once       ", lambda: fused.weight_grad_cat(dG, x1, h_all, o1, o2, seq=(dones, T)))):
    for _ in range(5):
        assert f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    print("%s %.1f us (with its column sums)" % (name, (time.perf_counter() - t0) / 50 * 1e6))
