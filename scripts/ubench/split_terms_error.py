#!/usr/bin/env python3
"""is the 6-pair form of the split-product rollout kernels narrower than the reference's fp32 GEMM?
Error against float64 of (a) the native fp32 matrix-core kernels, (b) the 9-pair split, (c) the 6-pair split, on the same
inputs -- the LSTM step (max / rms of |h - f64| and |c - f64|) and the three-layer MLP (|y - f64|) -- over several operand
distributions (binade-spanning as in the tests, and rollout-like: normalised observations / unit-scale states, weights at
initialisation scale) and seeds.  Prints one line per case and the worst ratios."""
import sys, os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd import native

lib = native.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
H, K = 256, 352


def lstm_case(seed, dist, N=4096):
    torch.manual_seed(seed)
    if dist == "binades":
        xh = torch.randn(N, K, device=dev) * torch.exp2(torch.randint(-20, 3, (N, K), device=dev).float())
        wcat = torch.randn(4 * H, K, device=dev) / np.sqrt(K) * torch.exp2(torch.randint(-12, 3, (4 * H, K), device=dev).float())
    elif dist == "rollout":      # ELU outputs / clamped normalised observations / tanh-bounded h; torch.nn.LSTM initial scale
        xh = torch.cat([torch.nn.functional.elu(torch.randn(N, 64, device=dev)), torch.randn(N, 32, device=dev).clamp(-5, 5),
                        torch.tanh(torch.randn(N, 256, device=dev)) * torch.rand(N, 256, device=dev)], dim=1)
        wcat = (torch.rand(4 * H, K, device=dev) * 2 - 1) / np.sqrt(H)
    else:                        # trained-like: heavier weights
        xh = torch.randn(N, K, device=dev)
        wcat = torch.randn(4 * H, K, device=dev) * 0.15
    xh[:, 92:96] = 0.0
    bias = torch.randn(4 * H, device=dev) * 0.1
    c_prev = torch.randn(N, H, device=dev)
    g = xh.double() @ wcat.double().t() + bias.double()
    i, f, gg, o = (g[:, k * H:(k + 1) * H] for k in range(4))
    c = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    out = {}
    wt = torch.empty(4 * H * K, device=dev)
    assert lib.vine_lstm_tile_weights_f32(H, K, wcat.data_ptr(), wcat.stride(0), wt.data_ptr(), st) == 0
    ho, co = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
    assert lib.vine_lstm_step_f32(N, H, K, xh.data_ptr(), K, wt.data_ptr(), bias.data_ptr(), c_prev.data_ptr(), ho.data_ptr(), H,
                                  co.data_ptr(), None, 0, st) == 0
    torch.cuda.synchronize()
    out["native"] = (ho.double() - h, co.double() - c)
    ws = torch.empty(3 * 4 * H * K, device=dev, dtype=torch.bfloat16)
    assert lib.vine_lstm_tile_weights_split(H, K, wcat.data_ptr(), wcat.stride(0), ws.data_ptr(), st) == 0
    for terms, name in ((9, "9"), (6, "6"), (9 | (3 << 16), "9d"), (6 | (3 << 16), "6d")):      # d: one gate per wave, two accumulators
        assert lib.vine_lstm_step_f32_split(N, H, K, xh.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c_prev.data_ptr(),
                                            ho.data_ptr(), H, co.data_ptr(), None, 0, terms, st) == 0
        torch.cuda.synchronize()
        out[name] = (ho.double() - h, co.double() - c)
    return {k: (float(v[0].abs().max()), float(v[0].pow(2).mean().sqrt()), float(v[1].abs().max()), float(v[1].pow(2).mean().sqrt()))
            for k, v in out.items()}


def mlp_case(seed, dist, n=4096, F=28):
    torch.manual_seed(seed)
    ldx = 352
    raw = torch.randn(n, F, device=dev) * 2.0 + 0.3
    mean = torch.randn(F, device=dev, dtype=torch.float64) * 0.2
    var = torch.rand(F, device=dev, dtype=torch.float64) + 0.3
    if dist == "binades":
        Ws = [torch.randn(o, i, device=dev) / np.sqrt(i) * torch.exp2(torch.randint(-6, 2, (o, i), device=dev).float())
              for o, i in ((256, F), (128, 256), (64, 128))]
    else:
        Ws = [(torch.rand(o, i, device=dev) * 2 - 1) / np.sqrt(i) * (1.0 if dist == "rollout" else 3.0) for o, i in ((256, F), (128, 256), (64, 128))]
    bs = [torch.randn(o, device=dev) * 0.1 for o in (256, 128, 64)]
    xn = torch.clamp((raw - mean.float()) / torch.sqrt(var.float() + 1e-5), -5.0, 5.0)
    a = xn.double()
    for W, b in zip(Ws, bs):
        a = torch.nn.functional.elu(a @ W.double().t() + b.double())
    out = {}
    x = torch.full((n, ldx), 7.0, device=dev)
    w1p = torch.zeros(256, 32, device=dev)
    w1p[:, :F] = Ws[0]
    assert lib.vine_mlp3_elu_f32(n, x.data_ptr(), ldx, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0, w1p.data_ptr(), 32,
                                 bs[0].data_ptr(), 256, Ws[1].data_ptr(), Ws[1].stride(0), bs[1].data_ptr(), 128, Ws[2].data_ptr(),
                                 Ws[2].stride(0), bs[2].data_ptr(), 64, 1.0, st) == 0
    torch.cuda.synchronize()
    out["native"] = x[:, :64].double() - a
    wt = torch.empty(288 * 512, device=dev, dtype=torch.bfloat16)
    assert lib.vine_mlp3_tile_weights_split(Ws[0].data_ptr(), Ws[0].stride(0), F, Ws[1].data_ptr(), Ws[1].stride(0), Ws[2].data_ptr(),
                                            Ws[2].stride(0), wt.data_ptr(), st) == 0
    for terms, name in ((9, "9"), (6, "6"), (9 | (1 << 16), "9d"), (6 | (1 << 16), "6d")):      # d: two accumulators per tile
        assert lib.vine_mlp3_elu_f32_split(n, x.data_ptr(), ldx, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                           wt.data_ptr(), bs[0].data_ptr(), bs[1].data_ptr(), bs[2].data_ptr(), 1.0, terms, None, 0.0,
                                           None, None, 0, st) == 0
        torch.cuda.synchronize()
        out[name] = x[:, :64].double() - a
    return {k: (float(v.abs().max()), float(v.pow(2).mean().sqrt())) for k, v in out.items()}


worst = {}
for kind, fn in (("lstm", lstm_case), ("mlp", mlp_case)):
    for dist in ("binades", "rollout", "heavy"):
        for seed in range(4):
            r = fn(seed, dist)
            line = "%-4s %-8s seed %d:" % (kind, dist, seed)
            for k in [k for k in ("native", "9", "6", "9d", "6d") if k in r]:
                line += "  %s " % k + " ".join("%.3e" % v for v in r[k])
            print(line)
            for k in [k for k in ("9", "6", "9d", "6d") if k in r]:
                for j, (a, b) in enumerate(zip(r[k], r["native"])):
                    key = (kind, k, j)
                    worst[key] = max(worst.get(key, 0.0), a / b)
print("worst ratio to the native fp32 MFMA kernel's error (columns: lstm = max|dh| rms|dh| max|dc| rms|dc|; mlp = max|dy| rms|dy|)")
for key in sorted(worst):
    print("   %s %s-pair column %d: %.3f" % (key[0], key[1], key[2], worst[key]))
