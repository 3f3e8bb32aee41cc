#!/usr/bin/env python3
"""In-kernel clock and per-wave phases of vine_mlp3_elu_f32_split (debug build: scripts/ab_build.sh splitt "-DSPLIT_TIMING",
then VINE_HIP_LIB=build/libvine_splitt.so python scripts/ubench/mlp_split_clock.py <variant> [rows]).  Back-to-back launches
on random operands, then the stamps of the last launch (ONE variant per process: the stamp buffer is not cleared)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.learning import fused  # noqa: E402

lib = fused._lib()
dev = torch.device("cuda:0")
v = int(sys.argv[1], 0) if len(sys.argv) > 1 else 9
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K, F = 352, 28
torch.manual_seed(0)
xh = [torch.randn(N, K, device=dev) for _ in range(2)]
raw = torch.randn(N, F, device=dev)
mean, var = torch.zeros(F, device=dev, dtype=torch.float64), torch.ones(F, device=dev, dtype=torch.float64)
Ws = [torch.randn(o, i, device=dev) / np.sqrt(i) for o, i in ((256, F), (128, 256), (64, 128))]
bs = [torch.randn(o, device=dev) * 0.1 for o in (256, 128, 64)]
st = torch.cuda.current_stream().cuda_stream
wm = torch.empty(288 * 512, device=dev, dtype=torch.bfloat16)
assert lib.vine_mlp3_tile_weights_split(Ws[0].data_ptr(), F, F, Ws[1].data_ptr(), 256, Ws[2].data_ptr(), 128, wm.data_ptr(), st) == 0
lib.vine_debug_mlp_split_timing.argtypes = [C.c_void_p]
iters = int(os.environ.get("SPLIT_ITERS", "5000"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(iters):
    a = xh[i & 1]
    assert lib.vine_mlp3_elu_f32_split(N, a.data_ptr(), K, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                       wm.data_ptr(), bs[0].data_ptr(), bs[1].data_ptr(), bs[2].data_ptr(), 1.0, v,
                                       None, 0.0, None, None, 0, st) == 0
e1.record()
torch.cuda.synchronize()
buf = (C.c_uint64 * (4096 * 16))()
assert lib.vine_debug_mlp_split_timing(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8, 2).astype(np.int64)
t = t[t[:, 0, 1] > 0]
clk = (t[:, 7, 0] - t[:, 0, 0]) / np.maximum(t[:, 7, 1] - t[:, 0, 1], 1) * 100.0          # MHz
r0 = t[:, :, 1].min()
rel = (t[:, :, 1] - r0) / 100.0                                                           # us since the first wave's entry
print("variant 0x%06x rows %d  %.1f us per launch; %d waves stamped; in-kernel clock %.0f MHz (median; %.0f .. %.0f)"
      % (v, N, e0.elapsed_time(e1) / iters * 1e3, len(t), np.median(clk), clk.min(), clk.max()))
names = ("entry", "obs + barrier", "layer-1 matrix", "layer-1 epilogue + barrier", "layer-2 matrix", "layer-2 epilogue + barrier",
         "layer-3 matrix", "end")
cyc = t[:, :, 0] - t[:, 0:1, 0]
for i, nm in enumerate(names):
    print("   %-28s at mean %6.2f us (min %6.2f max %6.2f)   phase cycles mean %7.0f"
          % (nm, rel[:, i].mean(), rel[:, i].min(), rel[:, i].max(), (cyc[:, i] - cyc[:, i - 1]).mean() if i else 0.0))
