#!/usr/bin/env python3
"""Back-to-back launches of chosen vine_lstm_step_f32_split variants at the rollout's shapes, for rocprofv3 --pmc passes.
Usage: python scripts/ubench/lstm_split_variants.py [launches] [variant ...]   (variant = terms + 256 rt)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.learning import fused  # noqa: E402

lib = fused._lib()
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
variants = [int(v, 0) for v in sys.argv[2:]] or [9]
N, H, K = 16384, 256, 352
torch.manual_seed(0)
ZERO = bool(int(os.environ.get("SPLIT_ZERO", "0")))       # all-zero operands: the clock the chip holds without data toggling
xh = [torch.randn(N, K, device=dev) * (0.0 if ZERO else 1.0) for _ in range(2)]
wcat = torch.randn(4 * H, K, device=dev) / np.sqrt(K) * (0.0 if ZERO else 1.0)
bias = torch.randn(4 * H, device=dev) * 0.1
c = torch.randn(N, H, device=dev)
h = torch.empty(N, H, device=dev)
ws = torch.empty(3 * 4 * H * K, device=dev, dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
assert lib.vine_lstm_tile_weights_split(H, K, wcat.data_ptr(), K, ws.data_ptr(), st) == 0
for v in variants * int(os.environ.get("SPLIT_ROUNDS", "1")):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(iters + 3):
        if i == 3:
            e0.record()
        a, b = xh[i & 1], xh[(i & 1) ^ 1]
        assert lib.vine_lstm_step_f32_split(N, H, K, a.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c.data_ptr(), h.data_ptr(),
                                            H, c.data_ptr(), b.data_ptr() + 4 * 96, K, v, st) == 0
    e1.record()
    torch.cuda.synchronize()
    print("variant 0x%08x  %7.1f us" % (v, e0.elapsed_time(e1) / iters * 1e3))
