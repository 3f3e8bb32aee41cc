"""Cost of the bias partial-sum epilogue of vine_linear_bwd_elu_mfma: the three backward shapes of the default MLP
(+ the LSTM input gradient) with and without the partial output (HIP events, 300 launches each)."""
import torch
from vine_robot_isaacgymenvs_amd.learning import fused

lib = fused._lib()
dev, bf, n = "cuda", torch.bfloat16, 32768
st = torch.cuda.current_stream().cuda_stream
for K, N in ((1024, 64), (64, 128), (128, 256)):
    G = (torch.randn(n, K, device=dev) * 0.1).to(bf)
    Wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(bf)
    a = torch.randn(n, N, device=dev).to(bf)
    gz = torch.empty(n, N, device=dev, dtype=bf)
    part = torch.empty(n // 64, N, device=dev)
    for label, pp in (("with partial", part.data_ptr()), ("no partial", None)):
        f = lambda: lib.vine_linear_bwd_elu_mfma(n, N, K, G.data_ptr(), K, Wt.data_ptr(), K, a.data_ptr(), N, 1.0,
                                                 gz.data_ptr(), N, pp, st)
        for _ in range(20):
            assert f() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            f()
        e1.record()
        torch.cuda.synchronize()
        print("K=%4d N=%3d %-13s %.1f us" % (K, N, label, e0.elapsed_time(e1) / 300 * 1e3))
