"""Stand-alone timing of vine_ln_heads_loss at the update's shape (32768 samples, H 256, 2 actions), 16-bit input in the
LSTM kernel's [B, T + 1, H] layout and fp32 input, rotating over input sets larger than the caches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd.abi import PPO_LOSS_SCRATCH_FLOATS
from vine_robot_isaacgymenvs_amd.learning import fused

dev = torch.device("cuda:0")
lib = fused._lib()
bf = fused.lp_dtype()
n, H, A, T, SETS = 32768, 256, 2, 4, 12
NH = A + 1
torch.manual_seed(0)
gamma, beta = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
w, wb = torch.randn(NH, H, device=dev) * 0.05, torch.randn(NH, device=dev) * 0.1
logstd = torch.tensor([-0.3, 0.2], device=dev)
actions = torch.randn(n, A, device=dev)
old_mu, old_sigma = 0.5 * actions + 0.1 * torch.randn(n, A, device=dev), torch.rand(n, A, device=dev) * 0.5 + 0.7
old_nlp = (0.5 * (((actions - old_mu) / old_sigma) ** 2).sum(-1) + 0.9189385 * A + old_sigma.log().sum(-1))
adv, old_values, returns = torch.randn(n, device=dev), torch.randn(n, device=dev), torch.randn(n, device=dev)
scal = (0.2, 1, 2.0, 0.01, 0.0001, 1.1)
R = lib.vine_ln_heads_loss_rows()
st = torch.cuda.current_stream().cuda_stream
heads, part = torch.empty(n, NH, device=dev), torch.empty(n // R, (2 + NH) * H, device=dev)
stats, gls, gmb, gvb = torch.empty(8, device=dev), torch.empty(A, device=dev), torch.zeros(A, device=dev), torch.zeros(1, device=dev)
kl, acc, mu, sg = torch.zeros(1, device=dev), torch.zeros(A, device=dev), torch.empty(n, A, device=dev), torch.empty(n, A, device=dev)
scratch = torch.empty(PPO_LOSS_SCRATCH_FLOATS, device=dev)
scale, found = torch.full((1,), 4.0, device=dev), torch.zeros(1, device=dev)
for name, flags, mk in (("16-bit [B, T+1, H]", 3 | (T << 8), lambda: (torch.randn(n // T * (T + 1), H, device=dev) * 0.6).to(bf)),
                        ("fp32 [n, H]", 1, lambda: torch.randn(n, H, device=dev) * 0.6)):
    xs = [mk() for _ in range(SETS)]
    dxs = [torch.empty(n, H, device=dev, dtype=bf) for _ in range(SETS)]

    def run(i):
        assert lib.vine_ln_heads_loss(n, H, NH, xs[i % SETS].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(),
                                      wb.data_ptr(), logstd.data_ptr(), actions.data_ptr(), old_nlp.data_ptr(), adv.data_ptr(),
                                      old_values.data_ptr(), returns.data_ptr(), old_mu.data_ptr(), old_sigma.data_ptr(), *scal,
                                      heads.data_ptr(), dxs[i % SETS].data_ptr(), flags, part.data_ptr(), stats.data_ptr(),
                                      gls.data_ptr(), gmb.data_ptr(), gvb.data_ptr(), scratch.data_ptr(), kl.data_ptr(),
                                      acc.data_ptr(), mu.data_ptr(), sg.data_ptr(), scale.data_ptr(), found.data_ptr(), st) == 0
    for i in range(SETS):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(60):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    print("%-20s %.1f us" % (name, e0.elapsed_time(e1) / 60 * 1e3), flush=True)
