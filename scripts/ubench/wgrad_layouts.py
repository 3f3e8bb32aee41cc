"""Does the library weight-gradient product run faster when dy is handed over already transposed ([M, rows], the
reduction index contiguous) instead of as a transposed view of [rows, M]?  Run under rocprofv3 --kernel-trace."""
import torch

dev, bf, n, s = "cuda", torch.bfloat16, 32768, 32
for M, N in ((1024, 256), (1024, 96), (128, 256)):
    dy = (torch.randn(n, M, device=dev) * 0.1).to(bf)
    x = torch.randn(n, N, device=dev).to(bf)
    a_view = dy.unflatten(0, (s, n // s)).transpose(1, 2)                 # [s, M, n/s] view, M contiguous
    a_cont = a_view.contiguous()                                          # [s, M, n/s], reduction index contiguous
    b = x.unflatten(0, (s, n // s))
    for a in (a_view, a_cont):
        for _ in range(10):
            torch.bmm(a, b, out_dtype=torch.float32)
        torch.cuda.synchronize()
