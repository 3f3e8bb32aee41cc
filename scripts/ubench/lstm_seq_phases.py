"""Per-step time stamps of the persistent LSTM forward kernel (debug build: scripts/ab_build.sh seqt "-DSEQ_TIMING", then
VINE_HIP_LIB=build/libvine_seqt.so python scripts/ubench/lstm_seq_phases.py).  Runs the update's form of the kernel
(8192 sequences x 4 steps, 16-bit saved states, hidden states stored once) on cold operand sets and prints, per time step,
the mean over all waves of: step entered / x k-steps done (= arrival at the step's barrier) / matrix loop left / epilogue
(pointwise + stores issued) done, relative to the earliest stamp (wall_clock64: 10 ns).  With SEQ_IN_SITU=1 the stamps are
those of the last launch of a graphed PPO iteration (bench.py's agent) instead."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.learning import bench_support, fused

dev = torch.device("cuda:0")
if os.environ.get("SEQ_IN_SITU"):
    import subprocess
    # (a child process would not share the device symbol: run the bench in this process)
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-other-configs", "--no-saturated", "--no-secondary"]
    import runpy
    try:
        runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"), run_name="__main__")
    except SystemExit:
        pass
    lib = fused._lib()
    buf = (C.c_uint64 * (256 * 8 * 16))()
    lib.vine_debug_seq_timing.argtypes = [C.c_void_p]
    assert lib.vine_debug_seq_timing(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256 * 8, 4, 4).astype(np.int64)
    t0 = t.min()
    names = ["step entered", "x part done (at the barrier)", "loop left", "epilogue done"]
    for step in range(4):
        print("step %d: " % step + "   ".join("%s %6.2f (max %6.2f)" % (nm, (t[:, step, i] - t0).mean() / 100.0, (t[:, step, i] - t0).max() / 100.0)
                                              for i, nm in enumerate(names)))
    sys.exit(0)
res = bench_support.ppo_kernel_rooflines(dev)          # (its last launches are the forward... then the backward kernel)
print({r["kernel"]: round(r["us"], 1) for r in res})
lib = fused._lib()
# one more forward launch on cold data so that the stamps are the forward kernel's
B, T, H, width, wpad = 8192, 4, 256, 92, 96
bf = fused.lp_dtype()
w_ih = (torch.randn(4 * H, width, device=dev) / 10).to(bf)
w_hh = (torch.randn(4 * H, H, device=dev) / 16).to(bf)
wtile, whh = torch.empty(4 * H * (wpad + H), device=dev, dtype=bf), torch.empty(4 * H * H, device=dev, dtype=bf)
prep = fused.CopyBatch(); prep.add_lstm_tiles(w_ih, w_hh, wpad, wtile, whh); prep.flush(wtile)
x = torch.zeros(B * T, wpad, device=dev, dtype=bf); x[:, :width] = (torch.randn(B * T, width, device=dev) * 0.7).to(bf)
c0, h0 = torch.randn(B, H, device=dev) * 0.5, torch.randn(B, H, device=dev) * 0.5
dones = (torch.rand(B * T, device=dev) < 0.2).to(torch.uint8)
out = torch.empty(B * (T + 1), H, device=dev, dtype=bf)
c_all, c_last = torch.empty(T + 1, B, H, device=dev, dtype=bf), torch.empty(B, H, device=dev)
gates = torch.empty(T, B, 4 * H, device=dev, dtype=bf)
bias = torch.zeros(4 * H, device=dev)
flush = torch.empty(1 << 28, device=dev); flush.fill_(1.0)          # 1 GiB: caches cold
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
assert lib.vine_lstm_seq_forward_mfma(B, T, H, wpad, x.data_ptr(), wpad, None, T * H, wtile.data_ptr(), bias.data_ptr(),
                                      c0.data_ptr(), dones.data_ptr(), out.data_ptr(), c_all.data_ptr(), gates.data_ptr(), 3,
                                      c_last.data_ptr(), h0.data_ptr(), st) == 0
torch.cuda.synchronize()
buf = (C.c_uint64 * (256 * 8 * 16))()
lib.vine_debug_seq_timing.argtypes = [C.c_void_p]
assert lib.vine_debug_seq_timing(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256 * 8, 4, 4).astype(np.int64)
t0 = t.min()
names = ["step entered", "x part done (at the barrier)", "loop left", "epilogue done"]
for step in range(4):
    print("step %d: " % step + "   ".join("%s %6.2f (max %6.2f)" % (nm, (t[:, step, i] - t0).mean() / 100.0, (t[:, step, i] - t0).max() / 100.0)
                                          for i, nm in enumerate(names)))
