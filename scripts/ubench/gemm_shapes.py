"""GEMM shapes of one PPO optimiser step (minibatch 32768 = 8192 sequences x 4) in fp32 / bf16 / fp16 operands
(fp32 accumulate): what a reduced-precision operand path could buy.  Run on the GPU box."""
import torch

dev = "cuda"
n, B, H, F = 32768, 8192, 256, 92


def bench(f, iters=30):
    for _ in range(5):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


shapes = {
    "L1 fwd   [n,28]x[28,256]": (n, 28, 256, 1),
    "L2 fwd   [n,256]x[256,128]": (n, 256, 128, 1),
    "L3 fwd   [n,128]x[128,64]": (n, 128, 64, 1),
    "ig fwd   [n,92]x[92,1024]": (n, F, 4 * H, 1),
    "ig fwd   [n,96]x[96,1024] (padded K)": (n, 96, 4 * H, 1),
    "hg fwd   [B,256]x[256,1024]  x4": (B, H, 4 * H, 4),
    "g_rec    [B,1024]x[1024,256] x3": (B, 4 * H, H, 3),
    "gx       [n,1024]x[1024,92]": (n, 4 * H, F, 1),
    "gx mlp   [n,1024]x[1024,64]": (n, 4 * H, 64, 1),
    "L2 bwd dx [n,128]x[128,256]": (n, 128, 256, 1),
    "L3 bwd dx [n,64]x[64,128]": (n, 64, 128, 1),
}
print("%-42s %9s %9s %9s" % ("shape (us per optimiser step)", "fp32", "bf16", "fp16"))
tot = {}
for name, (m, k, nn, cnt) in shapes.items():
    row = []
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        a = torch.randn(m, k, device=dev, dtype=dt)
        b = torch.randn(k, nn, device=dev, dtype=dt)
        t = bench(lambda: torch.mm(a, b)) * cnt
        row.append(t)
        tot[dt] = tot.get(dt, 0) + t
    print("%-42s %9.1f %9.1f %9.1f" % (name, *row))
# weight gradients: dY^T X with the 32768-long reduction, split-K by bmm as fused.splitk_tn does
wg = {"g_ih  [1024,n]x[n,92]": (4 * H, F), "g_hh  [1024,n]x[n,256]": (4 * H, H), "L2 dW [128,n]x[n,256]": (128, 256),
      "L1 dW [256,n]x[n,28]": (256, 28), "L3 dW [64,n]x[n,128]": (64, 128)}
for name, (m, k) in wg.items():
    row = []
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        dy = torch.randn(n, m, device=dev, dtype=dt)
        x = torch.randn(n, k, device=dev, dtype=dt)
        s = 32

        def f():
            part = torch.bmm(dy.view(s, n // s, m).transpose(1, 2), x.view(s, n // s, k))
            return part.sum(0)
        t = bench(f)
        row.append(t)
        tot[dt] = tot.get(dt, 0) + t
    print("%-42s %9.1f %9.1f %9.1f" % (name, *row))
print("%-42s %9.1f %9.1f %9.1f" % ("total", tot[torch.float32], tot[torch.bfloat16], tot[torch.float16]))
