#!/usr/bin/env python3
"""In-kernel clock and per-wave phases of vine_mlp3_elu_mfma (the update's MLP forward, 16-bit operands; debug build:
scripts/ab_build.sh splitt "-DSPLIT_TIMING", then VINE_HIP_LIB=build/libvine_splitt.so python scripts/ubench/mlp_mfma_clock.py
[rows]).  Back-to-back launches on random operands, then the stamps of the last launch."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.learning import fused  # noqa: E402

lib = fused._lib()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
K, F = 352, 28
lp = fused.lp_dtype()
torch.manual_seed(0)
xh = torch.zeros(N, K, device=dev, dtype=lp)
raw = torch.randn(N, F, device=dev)
mean, var = torch.zeros(F, device=dev, dtype=torch.float64), torch.ones(F, device=dev, dtype=torch.float64)
Ws = [(torch.randn(o, i, device=dev) / np.sqrt(i)).to(lp) for o, i in ((256, 32), (128, 256), (64, 128))]
Ws[0][:, F:] = 0
bs = [torch.randn(o, device=dev) * 0.1 for o in (256, 128, 64)]
a1, a2 = torch.empty(N, 256, device=dev, dtype=lp), torch.empty(N, 128, device=dev, dtype=lp)
st = torch.cuda.current_stream().cuda_stream
lib.vine_debug_mlp_split_timing.argtypes = [C.c_void_p]
iters = int(os.environ.get("SPLIT_ITERS", "3000"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(iters):
    assert lib.vine_mlp3_elu_mfma(N, xh.data_ptr() + 2 * 64, K, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                  Ws[0].data_ptr(), bs[0].data_ptr(), 256, Ws[1].data_ptr(), 256, bs[1].data_ptr(), 128,
                                  Ws[2].data_ptr(), 128, bs[2].data_ptr(), 64, 1.0, a1.data_ptr(), a2.data_ptr(), xh.data_ptr(),
                                  K, st) == 0
e1.record()
torch.cuda.synchronize()
buf = (C.c_uint64 * (4096 * 16))()
assert lib.vine_debug_mlp_split_timing(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8, 2).astype(np.int64)[:, :6]
t = t[t[:, 0, 1] > 0]
clk = (t[:, 5, 0] - t[:, 0, 0]) / np.maximum(t[:, 5, 1] - t[:, 0, 1], 1) * 100.0
r0 = t[:, :, 1].min()
rel = (t[:, :, 1] - r0) / 100.0
print("rows %d  %.1f us per launch; %d waves stamped; in-kernel clock %.0f MHz" % (N, e0.elapsed_time(e1) / iters * 1e3, len(t), np.median(clk)))
names = ("entry", "obs + W1 staged + barrier", "layer 1 (+ act1 stores)", "W2 / W3 staged + barrier", "layer 2 (+ act2 stores)", "layer 3, end")
cyc = t[:, :, 0] - t[:, 0:1, 0]
for i, nm in enumerate(names):
    print("   %-28s at mean %6.2f us (min %6.2f max %6.2f)   phase cycles mean %7.0f"
          % (nm, rel[:, i].mean(), rel[:, i].min(), rel[:, i].max(), (cyc[:, i] - cyc[:, i - 1]).mean() if i else 0.0))
