#!/usr/bin/env python3
"""Stand-alone timing of the two fp32 matrix-core rollout kernels (vine_mlp3_elu_f32, vine_lstm_step_f32) at the
rollout's shapes (16384 rows, K = 352, H = 256): HIP events over back-to-back launches; run it under
`rocprofv3 --kernel-trace --pmc ...` for the counters.  Usage: python scripts/ubench/rollout_f32.py [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.learning import fused  # noqa: E402

lib = fused._lib()
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
N, H, K, F = int(os.environ.get("ROLLOUT_N", "16384")), 256, 352, 28
torch.manual_seed(0)
xh = [torch.randn(N, K, device=dev) for _ in range(2)]
wcat = torch.randn(4 * H, K, device=dev) / np.sqrt(K)
bias = torch.randn(4 * H, device=dev) * 0.1
c = torch.randn(N, H, device=dev)
h = torch.empty(N, H, device=dev)
wt = torch.empty(4 * H * K, device=dev)
raw = torch.randn(N, F, device=dev)
mean, var = torch.zeros(F, device=dev, dtype=torch.float64), torch.ones(F, device=dev, dtype=torch.float64)
Ws = [torch.randn(o, i, device=dev) / np.sqrt(i) for o, i in ((256, F), (128, 256), (64, 128))]
bs = [torch.randn(o, device=dev) * 0.1 for o in (256, 128, 64)]
w1p = torch.zeros(256, 32, device=dev)
w1p[:, :F] = Ws[0]
st = torch.cuda.current_stream().cuda_stream
assert lib.vine_lstm_tile_weights_f32(H, K, wcat.data_ptr(), K, wt.data_ptr(), st) == 0


def lstm(i):
    a, b = xh[i & 1], xh[(i & 1) ^ 1]
    assert lib.vine_lstm_step_f32(N, H, K, a.data_ptr(), K, wt.data_ptr(), bias.data_ptr(), c.data_ptr(), h.data_ptr(), H,
                                  c.data_ptr(), b.data_ptr() + 4 * 96, K, st) == 0


def mlp(i):
    a = xh[i & 1]
    assert lib.vine_mlp3_elu_f32(N, a.data_ptr(), K, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                 w1p.data_ptr(), 32, bs[0].data_ptr(), 256, Ws[1].data_ptr(), 256, bs[1].data_ptr(), 128,
                                 Ws[2].data_ptr(), 128, bs[2].data_ptr(), 64, 1.0, st) == 0


wm = torch.empty(288 * 512, device=dev, dtype=torch.bfloat16)
assert lib.vine_mlp3_tile_weights_split(Ws[0].data_ptr(), F, F, Ws[1].data_ptr(), 256, Ws[2].data_ptr(), 128, wm.data_ptr(), st) == 0


def mlp_split(variant, rows):
    def run(i):
        a = xh[i & 1]
        assert lib.vine_mlp3_elu_f32_split(rows, a.data_ptr(), K, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                           wm.data_ptr(), bs[0].data_ptr(), bs[1].data_ptr(), bs[2].data_ptr(), 1.0, variant,
                                           None, 0.0, None, None, 0, st) == 0
    return run


def mlp_rows(rows):
    def run(i):
        a = xh[i & 1]
        assert lib.vine_mlp3_elu_f32(rows, a.data_ptr(), K, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                     w1p.data_ptr(), 32, bs[0].data_ptr(), 256, Ws[1].data_ptr(), 256, bs[1].data_ptr(), 128,
                                     Ws[2].data_ptr(), 128, bs[2].data_ptr(), 64, 1.0, st) == 0
    return run


MLP_FLOP = 2.0 * (32 * 256 + 256 * 128 + 128 * 64)
MLPS = tuple(("mlp_split t%d rt%d n%d" % (t, rt, rows), mlp_split(t + 256 * rt, rows), MLP_FLOP * rows)
             for rows in sorted({N, 4096}, reverse=True) for t in (9, 6) for rt in (4, 2, 1)) + \
    (("mlp3_elu_f32 n4096", mlp_rows(4096), MLP_FLOP * 4096),)
ws = torch.empty(3 * 4 * H * K, device=dev, dtype=torch.bfloat16)
assert lib.vine_lstm_tile_weights_split(H, K, wcat.data_ptr(), K, ws.data_ptr(), st) == 0


def lstm_split(variant):
    def run(i):
        a, b = xh[i & 1], xh[(i & 1) ^ 1]
        assert lib.vine_lstm_step_f32_split(N, H, K, a.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c.data_ptr(), h.data_ptr(),
                                            H, c.data_ptr(), b.data_ptr() + 4 * 96, K, variant, st) == 0
    return run


SPLITS = tuple(("lstm_split t%d rt%d" % (t, rt), lstm_split(t + 256 * rt), 2.0 * N * K * 4 * H) for t in (9, 6) for rt in (4, 2)) + \
    tuple(("lstm_nsplit t%d" % t, lstm_split(t + (1 << 16)), 2.0 * N * K * 4 * H) for t in (9, 6))
for name, fn, flop in MLPS + SPLITS + (("lstm_step_f32", lstm, 2.0 * N * K * 4 * H), ("mlp3_elu_f32", mlp, 2.0 * N * (32 * 256 + 256 * 128 + 128 * 64))):
    for i in range(5):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print("%-24s %7.1f us   %6.1f TFLOP/s  (%.0f %% of the 157.3 TFLOP/s fp32 matrix peak)" % (name, us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 100))
