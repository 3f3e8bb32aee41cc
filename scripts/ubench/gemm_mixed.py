"""bf16 operands with fp32 output (torch.mm out_dtype) vs bf16 output vs fp32: the LSTM GEMM shapes."""
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


print("torch", torch.__version__)
a = torch.randn(B, H, device=dev, dtype=torch.bfloat16)
b = torch.randn(H, 4 * H, device=dev, dtype=torch.bfloat16)
try:
    c = torch.mm(a, b, out_dtype=torch.float32)
    print("out_dtype ok", c.dtype, float((c - a.float() @ b.float()).abs().max()))
    have = True
except Exception as e:
    print("out_dtype failed:", repr(e)[:300])
    have = False
shapes = {"hg [B,256]x[256,1024]": (B, H, 4 * H), "ig [n,96]x[96,1024]": (n, 96, 4 * H), "g_rec [B,1024]x[1024,256]": (B, 4 * H, H),
          "gx [n,1024]x[1024,64]": (n, 4 * H, 64), "L2 [n,256]x[256,128]": (n, 256, 128)}
for name, (m, k, nn) in shapes.items():
    a32, b32 = torch.randn(m, k, device=dev), torch.randn(k, nn, device=dev)
    a16, b16 = a32.bfloat16(), b32.bfloat16()
    t32 = bench(lambda: torch.mm(a32, b32))
    t16 = bench(lambda: torch.mm(a16, b16))
    tmix = bench(lambda: torch.mm(a16, b16, out_dtype=torch.float32)) if have else float("nan")
    tcast = bench(lambda: a32.bfloat16())
    print("%-28s fp32 %7.1f  bf16->bf16 %7.1f  bf16->fp32 %7.1f   cast A fp32->bf16 %6.1f us" % (name, t32, t16, tmix, tcast))
# split-K weight gradient with bf16 operands, fp32 partials
for name, (m, k) in {"g_hh [1024,n]x[n,256]": (4 * H, H), "g_ih [1024,n]x[n,96]": (4 * H, 96)}.items():
    dy32, x32 = torch.randn(n, m, device=dev), torch.randn(n, k, device=dev)
    dy16, x16 = dy32.bfloat16(), x32.bfloat16()
    s = 32
    f32 = lambda: torch.bmm(dy32.view(s, n // s, m).transpose(1, 2), x32.view(s, n // s, k)).sum(0)
    f16 = lambda: torch.bmm(dy16.view(s, n // s, m).transpose(1, 2), x16.view(s, n // s, k)).float().sum(0)
    res = [bench(f32), bench(f16)]
    if have:
        fm = lambda: torch.bmm(dy16.view(s, n // s, m).transpose(1, 2), x16.view(s, n // s, k), out_dtype=torch.float32).sum(0)
        try:
            res.append(bench(fm))
        except Exception as e:
            print("bmm out_dtype failed", repr(e)[:200])
            res.append(float("nan"))
    # one plain GEMM with the long K (no split): how do the bf16 heuristics do?
    res.append(bench(lambda: torch.mm(dy16.t(), x16)))
    print("%-28s fp32 splitK %7.1f  bf16 splitK(bf16 partials) %7.1f  bf16 splitK(fp32 partials) %7.1f  bf16 plain mm %7.1f"
          % (name, res[0], res[1], res[2] if len(res) > 3 else float("nan"), res[-1]))
