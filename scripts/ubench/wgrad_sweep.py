"""Weight-gradient GEMMs of one optimiser step (dy^T x, n = 32768 rows, bf16 operands, fp32 result): slices of the
row reduction (bmm + column sums over the slices) swept per layer shape, next to a plain torch.mm over all rows."""
import torch
from vine_robot_isaacgymenvs_amd.learning import fused

dev = "cuda"
n = 32768


def bench(f, iters=40):
    for _ in range(5):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


shapes = {"W1 [256,32]": (256, 32), "W2 [128,256]": (128, 256), "W3 [64,128]": (64, 128), "w_ih [1024,96]": (1024, 96),
          "w_hh [1024,256]": (1024, 256), "w_cat [1024,352]": (1024, 352)}
for name, (M, N) in shapes.items():
    dy = torch.randn(n, M, device=dev).bfloat16()
    x = torch.randn(n, N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev)
    line = "%-18s" % name
    for s in (1, 2, 4, 8, 16, 32, 64, 128):
        if s == 1:
            t = bench(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
            line += "  plain %6.1f" % t
            continue
        a, b = dy.unflatten(0, (s, n // s)).transpose(1, 2), x.unflatten(0, (s, n // s))
        tb = bench(lambda: torch.bmm(a, b, out_dtype=torch.float32))
        part = torch.bmm(a, b, out_dtype=torch.float32)
        tc = bench(lambda: fused.column_sums(part, out))
        line += "  s%-3d %5.1f+%4.1f" % (s, tb, tc)
    print(line, flush=True)
