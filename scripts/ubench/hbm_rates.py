"""What the memory system of the box sustains for plain streaming kernels (PyTorch elementwise kernels, 1 GiB tensors, far
beyond L2 + MALL): copy (read + write), read-only (sum), write-only (fill).  Total bytes moved / time; the LSTM sequence
kernels' mixed read/write traffic is priced against these beside the 8 TB/s spec figure."""
import torch

dev = torch.device("cuda:0")
n = 1 << 28                                     # 2^28 floats = 1 GiB
x = torch.randn(n, device=dev)
y = torch.empty_like(x)


def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t = timed(lambda: y.copy_(x))
print("copy  (1 GiB read + 1 GiB write): %.0f us  %.2f TB/s total" % (t * 1e6, 2 * 4 * n / t / 1e12))
t = timed(lambda: x.sum())
print("read  (1 GiB, sum):               %.0f us  %.2f TB/s" % (t * 1e6, 4 * n / t / 1e12))
t = timed(lambda: y.fill_(1.0))
print("write (1 GiB, fill):              %.0f us  %.2f TB/s" % (t * 1e6, 4 * n / t / 1e12))
t = timed(lambda: torch.add(x, 1.0, out=y))
print("x + 1 -> y (read + write):        %.0f us  %.2f TB/s total" % (t * 1e6, 2 * 4 * n / t / 1e12))
