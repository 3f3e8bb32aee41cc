"""Stand-alone timing of vine_lstm_step_mfma against the unfused pair (bf16 GEMM + vine_lstm_cell_forward) at the
update's shape (B = 8192, K = H = 256, T-strided operands like the trunk's).  Run on the GPU box; also the target of
rocprofv3 --pmc passes when the kernel is being tuned."""
import sys
import torch
sys.path.insert(0, ".")
from vine_robot_isaacgymenvs_amd.learning import fused

dev = torch.device("cuda:0")
lib = fused._lib()
st = torch.cuda.current_stream().cuda_stream
B, H, T = 8192, 256, 4
bf = torch.bfloat16
hp = (torch.randn(B, T, H, device=dev) * 0.5).to(bf)
W = (torch.randn(4 * H, H, device=dev) / 16).to(bf)
ig = torch.randn(B, T, 4 * H, device=dev)
bias = torch.randn(4 * H, device=dev) * 0.1
c0, c1 = torch.randn(B, H, device=dev), torch.empty(B, H, device=dev)
out = torch.empty(B, T, H, device=dev)
gates = torch.empty(B, 4 * H, device=dev, dtype=bf)


xfull = (torch.randn(B * T, 96, device=dev) * 0.5).to(bf)
wcat = (torch.randn(4 * H, 96 + H, device=dev) / 16).to(bf)


def fused2():      # the update's two-operand form: [x_t (96) | h_{t-1} (256)], no input projection
    rc = lib.vine_lstm_step_mfma(B, H, 96 + H, xfull.data_ptr(), T * 96, hp.data_ptr(), T * H, 96, wcat.data_ptr(), 96 + H,
                                 None, 4 * H, bias.data_ptr(), c0.data_ptr(), None, 0, out.data_ptr(), T * H, c1.data_ptr(),
                                 gates.data_ptr(), hp.data_ptr() + 2 * H, None, 0, T * H, st)
    assert rc == 0


def fusedk():
    rc = lib.vine_lstm_step_mfma(B, H, H, hp.data_ptr(), T * H, None, 0, 0, W.data_ptr(), H, ig.data_ptr(), T * 4 * H, bias.data_ptr(),
                                 c0.data_ptr(), None, 0, out.data_ptr(), T * H, c1.data_ptr(), gates.data_ptr(),
                                 hp.data_ptr() + 2 * H, None, 0, T * H, st)
    assert rc == 0


def unfused():
    hg = torch.mm(hp[:, 0], W.t(), out_dtype=torch.float32)
    rc = lib.vine_lstm_cell_forward(B, H, ig.data_ptr(), T * 4 * H, hg.data_ptr(), bias.data_ptr(), c0.data_ptr(), None, 0,
                                    out.data_ptr(), T * H, c1.data_ptr(), gates.data_ptr(), hp.data_ptr() + 2 * H, None, 0,
                                    1, T * H, st)
    assert rc == 0


for name, f in (("fused mfma K=352", fused2), ("fused mfma K=256", fusedk), ("gemm + pointwise", unfused)):
    for _ in range(10):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100):
        f()
    e.record()
    torch.cuda.synchronize()
    print("%-18s %6.1f us" % (name, s.elapsed_time(e) * 10))
