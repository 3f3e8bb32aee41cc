#!/usr/bin/env python3
"""In-kernel clock and per-wave phases of vine_lstm_step_f32_split (debug build: scripts/ab_build.sh splitt "-DSPLIT_TIMING",
then VINE_HIP_LIB=build/libvine_splitt.so python scripts/ubench/lstm_split_clock.py [variant ...]).  ~2 s of back-to-back
launches on random operands, then the stamps of the last launch (ONE variant per process: the stamp buffer is not cleared): shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.learning import fused  # noqa: E402

lib = fused._lib()
dev = torch.device("cuda:0")
variants = [int(v, 0) for v in sys.argv[1:]] or [9]
N, H, K = int(os.environ.get("SPLIT_N", "16384")), 256, 352
torch.manual_seed(0)
xh = [torch.randn(N, K, device=dev) for _ in range(2)]
wcat = torch.randn(4 * H, K, device=dev) / np.sqrt(K)
bias = torch.randn(4 * H, device=dev) * 0.1
c = torch.randn(N, H, device=dev)
h = torch.empty(N, H, device=dev)
ws = torch.empty(3 * 4 * H * K, device=dev, dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
assert lib.vine_lstm_tile_weights_split(H, K, wcat.data_ptr(), K, ws.data_ptr(), st) == 0
lib.vine_debug_split_timing.argtypes = [C.c_void_p]
for v in variants:
    iters = int(os.environ.get("SPLIT_ITERS", "20000"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        a, b = xh[i & 1], xh[(i & 1) ^ 1]
        assert lib.vine_lstm_step_f32_split(N, H, K, a.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c.data_ptr(), h.data_ptr(),
                                            H, c.data_ptr(), b.data_ptr() + 4 * 96, K, v, st) == 0
    e1.record()
    torch.cuda.synchronize()
    buf = (C.c_uint64 * (8192 * 8))()
    assert lib.vine_debug_split_timing(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4, 2).astype(np.int64)
    t = t[t[:, 0, 1] > 0]
    clk = (t[:, 3, 0] - t[:, 0, 0]) / np.maximum(t[:, 3, 1] - t[:, 0, 1], 1) * 100.0          # MHz
    r0 = t[:, :, 1].min()
    rel = (t[:, :, 1] - r0) / 100.0                                                           # us since the first wave's entry
    print("variant 0x%06x  %.1f us per launch; %d waves stamped; in-kernel clock %.0f MHz (median; %.0f .. %.0f)"
          % (v, e0.elapsed_time(e1) / iters * 1e3, len(t), np.median(clk), clk.min(), clk.max()))
    for i, nm in enumerate(("entry", "prologue done", "matrix loops done", "end")):
        print("   %-18s mean %6.2f us   min %6.2f   max %6.2f" % (nm, rel[:, i].mean(), rel[:, i].min(), rel[:, i].max()))
    cyc = t[:, :, 0] - t[:, 0:1, 0]
    print("   shader cycles per wave: prologue %.0f, matrix loops %.0f, epilogue %.0f"
          % (cyc[:, 1].mean(), (cyc[:, 2] - cyc[:, 1]).mean(), (cyc[:, 3] - cyc[:, 2]).mean()))
    # waves in the order of their entry: lifetime and phases of the waves that entered first (the launch's first fill of the
    # chip) against the rest; when the last wave ended
    order = np.argsort(rel[:, 0])
    life = rel[:, 3] - rel[:, 0]
    q = max(1, len(order) // 8)
    for k in range(0, len(order), q):
        sel = order[k:k + q]
        print("   waves %5d..%5d by entry: entry %6.2f us, lifetime %6.2f us (prologue %5.0f, loops %6.0f, epilogue %5.0f cycles), end %6.2f"
              % (k, k + len(sel) - 1, rel[sel, 0].mean(), life[sel].mean(), cyc[sel, 1].mean(), (cyc[sel, 2] - cyc[sel, 1]).mean(),
                 (cyc[sel, 3] - cyc[sel, 2]).mean(), rel[sel, 3].mean()))
    print("   last wave ends %.2f us after the first entry" % rel[:, 3].max())
