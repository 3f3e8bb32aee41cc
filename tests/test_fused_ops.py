"""Numerics of the hand-written PPO-update kernels (csrc/ppo_kernels.hip) against the plain PyTorch fp32 composition
they replace.  Tolerances: fp32 with different summation orders / exp-based tanh -> 2e-5 relative on activations,
1e-4 on gradients reduced over 32768 rows."""
import numpy as np
import pytest
import torch

from vine_robot_isaacgymenvs_amd.learning import fused


def test_lstm_reference_matches_time_major_loop():
    """CPU: the sequence-major reference used as fallback == the module's time-major loop == nn.LSTM."""
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.network import ModelA2CContinuousLogStd
    m = ModelA2CContinuousLogStd(load_config()["train"]["params"]["network"], 2, (28,), True, True)
    w = m.a2c_network.rnn
    B, T = 6, 4
    x = torch.randn(B * T, 92)
    h0, c0 = torch.randn(1, B, 256), torch.randn(1, B, 256)
    dones = (torch.rand(B * T) < 0.3).to(torch.uint8)
    out, (h, c) = w.forward_flat(x, (h0, c0), dones, T)
    xt = x.view(B, T, -1).transpose(0, 1)
    dt = dones.view(B, T).transpose(0, 1)
    out2, (h2, c2) = w(xt, (h0, c0), dt)
    assert torch.allclose(out.view(B, T, -1).transpose(0, 1), out2, atol=1e-6)
    assert torch.allclose(h, h2, atol=1e-6) and torch.allclose(c, c2, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("use_dones", [False, True])
def test_lstm_sequence_forward_backward(use_dones):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, T, Fin, H = 2048, 4, 92, 256
    k = 1.0 / np.sqrt(H)
    params = [torch.empty(s, device=dev).uniform_(-k, k).requires_grad_() for s in ((4 * H, Fin), (4 * H, H), (4 * H,), (4 * H,))]
    x = torch.randn(B * T, Fin, device=dev, requires_grad=True)
    h0, c0 = torch.randn(B, H, device=dev) * 0.5, torch.randn(B, H, device=dev) * 0.5
    dones = (torch.rand(B * T, device=dev) < 0.2).to(torch.uint8) if use_dones else None
    gout = torch.randn(B * T, H, device=dev)

    def run(fn):
        for p in params + [x]:
            p.grad = None
        out, h, c = fn(x, *params, h0, c0, dones, T)
        out.backward(gout)
        return [out.detach(), h.detach(), c.detach(), x.grad.clone()] + [p.grad.clone() for p in params]

    a = run(fused.lstm_sequence)
    b = run(fused._lstm_reference)
    names = ["out", "hT", "cT", "dx", "dW_ih", "dW_hh", "db_ih", "db_hh"]
    for n, u, v in zip(names, a, b):
        scale = float(v.abs().max()) + 1e-6
        assert float((u - v).abs().max()) / scale < 2e-4, n
    # rollout shape: T = 1, no grad
    with torch.no_grad():
        o1, h1, c1 = fused.lstm_sequence(x[:B].detach(), *params, h0, c0, None, 1)
        o2, h2, c2 = fused._lstm_reference(x[:B].detach(), *params, h0, c0, None, 1)
    assert torch.allclose(o1, o2, atol=2e-5) and torch.allclose(c1, c2, atol=2e-5) and torch.equal(o1, h1)


@pytest.mark.gpu
def test_pointwise_kernels_against_torch():
    """vine_column_sums (wide and tall variants, split / duplicate outputs, padded rows), vine_layernorm_forward/
    backward, vine_bias_elu and vine_elu_backward (fp32 and bfloat16 storage) against the torch ops they replace."""
    import ctypes as C
    from vine_robot_isaacgymenvs_amd.abi import PPO_PARTIAL_BLOCKS
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    lib = fused._lib()
    st = torch.cuda.current_stream().cuda_stream
    # column sums
    for R, Cc in ((32, 1024 * 92), (2048, 1024), (512, 64), (7, 130), (300, 5)):
        src = torch.randn(R, Cc + 6, device=dev)[:, :Cc]                      # padded rows
        ref = src.double().sum(0)
        out = fused.column_sums(src)
        assert float((out.double() - ref).abs().max()) < 1e-4 * (1 + float(ref.abs().max()))
        a, b = torch.empty(Cc // 2, device=dev), torch.empty(Cc - Cc // 2, device=dev)
        fused.column_sums(src, a, out1=b, n0=Cc // 2)
        assert torch.equal(torch.cat([a, b]), out)
        a2, b2 = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        fused.column_sums(src, a2, out1=b2, dup=True)
        assert torch.equal(a2, out) and torch.equal(b2, out)
    part = torch.randn(16, 3, 256, device=dev)
    assert torch.allclose(fused.column_sums(part[:, :2]), part[:, :2].sum(0), atol=1e-5)
    # the same jobs through the one-launch batched kernel (more than 16 jobs: two launches)
    batch, expect = fused.ColumnSumBatch(), []
    for rep in range(3):
        for R, Cc in ((32, 1024 * 92), (2048, 1024), (512, 64), (7, 130), (300, 5), (512, 1280)):
            # rep 1: unpadded rows -> the 16-B geometry of the batched kernel for the short, wide jobs
            src = torch.randn(R, Cc + 6, device=dev)[:, :Cc] if rep != 1 else torch.randn(R, Cc, device=dev)
            a, b = torch.empty(Cc // 2, device=dev), torch.empty(Cc - Cc // 2, device=dev)
            batch.add(src, a, out1=b, n0=Cc // 2)
            expect.append((src.double().sum(0), a, b))
    batch.flush(part)
    for ref, a, b in expect:
        got = torch.cat([a, b]).double()
        assert float((got - ref).abs().max()) < 1e-4 * (1 + float(ref.abs().max()))
    # LayerNorm
    n, H = 4099, 256
    x = torch.randn(n, H, device=dev) * 2 + 0.5
    g, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
    y, mean, rstd = torch.empty_like(x), torch.empty(n, device=dev), torch.empty(n, device=dev)
    assert lib.vine_layernorm_forward(n, H, x.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-5, y.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), st) == 0
    xr = x.double().requires_grad_()
    gr, br = g.double().requires_grad_(), b.double().requires_grad_()
    yr = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-5)
    assert float((y.double() - yr).abs().max()) < 2e-5
    dy = torch.randn(n, H, device=dev)
    yr.backward(dy.double())
    dx = torch.empty_like(x)
    part = torch.empty(PPO_PARTIAL_BLOCKS, 2 * H, device=dev)
    assert lib.vine_layernorm_backward(n, H, dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(),
                                       dx.data_ptr(), part.data_ptr(), st) == 0
    assert float((dx.double() - xr.grad).abs().max()) < 5e-5
    sums = fused.column_sums(part)
    assert float((sums[:H].double() - gr.grad).abs().max()) < 1e-3 and float((sums[H:].double() - br.grad).abs().max()) < 1e-3
    # bias + ELU forward, ELU backward from the output (fp32 and bf16 storage, strided rows)
    for Cc in (64, 128, 256):
        z, bias = torch.randn(n, Cc, device=dev), torch.randn(Cc, device=dev)
        ref = torch.nn.functional.elu(z + bias)
        wide = torch.zeros(n, Cc + 32, device=dev)
        assert lib.vine_bias_elu(n, Cc, z.data_ptr(), bias.data_ptr(), 1.0, wide.data_ptr(), Cc + 32, 0, st) == 0
        assert float((wide[:, :Cc] - ref).abs().max()) < 1e-6 and float(wide[:, Cc:].abs().max()) == 0.0
        wb = torch.zeros(n, Cc + 32, device=dev, dtype=fused.lp_dtype())
        assert lib.vine_bias_elu(n, Cc, z.data_ptr(), bias.data_ptr(), 1.0, wb.data_ptr(), Cc + 32, 1, st) == 0
        # round-to-nearest-even like torch; the fast exp may land an element on the other side of a rounding boundary
        assert float(((wb[:, :Cc].float() - ref).abs() - ref.abs() * 2.0 ** -8).max()) < 1e-6
        assert float((wb[:, :Cc] != ref.to(fused.lp_dtype())).float().mean()) < 1e-3
        gin = torch.randn(n, Cc, device=dev)
        ref_g = gin * torch.where(ref > 0, torch.ones_like(ref), ref + 1.0)
        for a_bf, o_bf in ((0, 0), (1, 1), (1, 0), (0, 1)):
            a_t = wb if a_bf else wide
            out = torch.empty(n, Cc, device=dev, dtype=fused.lp_dtype() if o_bf else torch.float32)
            part = torch.empty(PPO_PARTIAL_BLOCKS, Cc, device=dev)
            assert lib.vine_elu_backward(n, Cc, gin.data_ptr(), Cc, a_t.data_ptr(), Cc + 32, 1.0, out.data_ptr(), Cc,
                                         part.data_ptr(), a_bf, o_bf, st) == 0
            tol = 2e-2 if (a_bf or o_bf) else 1e-6
            assert float((out.float() - ref_g).abs().max()) < tol * (1 + float(ref_g.abs().max()))
            assert float((fused.column_sums(part) - ref_g.sum(0)).abs().max()) < (0.5 if a_bf else 1e-3)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,with_ig", [(8192, 256, True), (1024, 352, False), (64, 128, True)])
def test_lstm_step_mfma_matches_gemm_plus_pointwise(B, K, with_ig):
    """The matrix-core LSTM step (recurrent GEMM fused with the pointwise update, pre-activations never stored) against
    the unfused pair it replaces: torch GEMM on the same bf16 operands + vine_lstm_cell_forward."""
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    lib = fused._lib()
    st = torch.cuda.current_stream().cuda_stream
    H = 256
    bf = fused.lp_dtype()
    A = (torch.randn(B, K + 16, device=dev) * 0.5).to(bf)[:, :K]                  # padded rows
    W = (torch.randn(4 * H, K, device=dev) / K ** 0.5).to(bf)
    ig = torch.randn(B, 4 * H, device=dev) if with_ig else None
    bias = torch.randn(4 * H, device=dev) * 0.1
    c0 = torch.randn(B, H, device=dev)
    done = (torch.rand(B, device=dev) < 0.3).to(torch.uint8)
    done_n = (torch.rand(B, device=dev) < 0.3).to(torch.uint8)

    def outputs():
        return (torch.empty(B, H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, 4 * H, device=dev, dtype=bf),
                torch.empty(B, H, device=dev, dtype=bf))
    pre = torch.mm(A, W.t(), out_dtype=torch.float32)                 # the GEMM the fused kernel absorbs
    zero_ig = torch.zeros(B, 4 * H, device=dev)
    h2, c2, g2, hp2 = outputs()
    rc = lib.vine_lstm_step_mfma(B, H, K, A.data_ptr(), A.stride(0), None, 0, 0, W.data_ptr(), W.stride(0),
                                 ig.data_ptr() if with_ig else None, 4 * H, bias.data_ptr(), c0.data_ptr(), done.data_ptr(), 1,
                                 h2.data_ptr(), H, c2.data_ptr(), g2.data_ptr(), hp2.data_ptr(), done_n.data_ptr(), 1, H, st)
    assert rc == 0
    keep = (1.0 - done.float()).unsqueeze(1)
    # fused kernel: no keep-scaling of the GEMM term (its operand is masked upstream), c is masked by `done`
    c_masked = c0 * keep
    h3, c3, g3, hp3 = outputs()
    assert lib.vine_lstm_cell_forward(B, H, (ig if with_ig else zero_ig).data_ptr(), 4 * H, pre.data_ptr(), bias.data_ptr(),
                                      c_masked.data_ptr(), None, 0, h3.data_ptr(), H, c3.data_ptr(), g3.data_ptr(),
                                      hp3.data_ptr(), done_n.data_ptr(), 1, 1, H, st) == 0
    torch.cuda.synchronize()
    for name, a, b, tol in (("h", h2, h3, 2e-5), ("c", c2, c3, 2e-5), ("gates", g2.float(), g3.float(), 8e-3),
                            ("hp", hp2.float(), hp3.float(), 8e-3)):
        assert float((a - b).abs().max()) < tol, (name, float((a - b).abs().max()))
    if K == 352:      # two-source form: the same product with the operand split into [x (96) | h (256)] buffers
        A1, A2 = A[:, :96].contiguous(), A[:, 96:].contiguous()
        h4, c4, g4, hp4 = outputs()
        assert lib.vine_lstm_step_mfma(B, H, K, A1.data_ptr(), 96, A2.data_ptr(), 256, 96, W.data_ptr(), W.stride(0), None,
                                       4 * H, bias.data_ptr(), c0.data_ptr(), done.data_ptr(), 1, h4.data_ptr(), H,
                                       c4.data_ptr(), g4.data_ptr(), hp4.data_ptr(), done_n.data_ptr(), 1, H, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(h4, h2) and torch.equal(c4, c2) and torch.equal(g4, g2) and torch.equal(hp4, hp2)
    assert lib.vine_lstm_step_mfma(B + 1, H, K, A.data_ptr(), A.stride(0), None, 0, 0, W.data_ptr(), W.stride(0), None, 4 * H,
                                   bias.data_ptr(), c0.data_ptr(), None, 0, h2.data_ptr(), H, c2.data_ptr(), None, None,
                                   None, 0, H, st) == -2                          # unsupported shape: caller falls back


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,use_dones", [(8192, 256, True), (128, 256, False), (64, 128, True)])
def test_lstm_backward_mfma_matches_gemm_plus_pointwise(B, H, use_dones):
    """The backward LSTM step with the recurrent input gradient formed on the matrix cores inside the kernel against
    the two launches it replaces (bf16 GEMM dG_{t+1} w_hh + vine_lstm_cell_backward), over a whole 4-step sequence:
    gate gradients (bf16), the final cell gradient chain and the chained bias partial sums."""
    import time
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    lib = fused._lib()
    T = 4
    bf = fused.lp_dtype()
    w_hh = (torch.randn(4 * H, H, device=dev) / H ** 0.5).to(bf)
    g_out = torch.randn(B * T, H, device=dev) * 0.1
    c_all = torch.randn(T + 1, B, H, device=dev)
    gates = torch.rand(T, B, 4 * H, device=dev)
    gates[:, :, 2 * H:3 * H] = gates[:, :, 2 * H:3 * H] * 2 - 1                # the tanh gate
    gates = gates.to(bf)
    dones = (torch.rand(B * T, device=dev) < 0.2).to(torch.uint8) if use_dones else None
    dG_a, part_a = fused._lstm_backward_steps(lib, g_out, w_hh, c_all, gates, dones, T)
    dG_b, part_b = fused._lstm_backward_steps(lib, g_out, w_hh, c_all, gates, dones, T, w_hh_t=w_hh.t().contiguous())
    torch.cuda.synchronize()
    a, b = dG_a.float(), dG_b.float()
    # same products, different summation order inside the recurrent term, results rounded to bf16
    assert float((a - b).abs().max()) <= 1e-2 * float(a.abs().max()), float((a - b).abs().max())
    assert float((a - b).abs().mean()) <= 2e-4 * float(a.abs().mean()) + 1e-9
    sa, sb = part_a.sum(0), part_b.sum(0)
    assert float((sa - sb).abs().max()) <= 2e-3 * float(sa.abs().max()), float((sa - sb).abs().max())
    if B == 8192:
        wt = w_hh.t().contiguous()
        for name, kw in (("gemm + pointwise", {}), ("fused mfma", {"w_hh_t": wt})):
            for _ in range(3):
                fused._lstm_backward_steps(lib, g_out, w_hh, c_all, gates, dones, T, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fused._lstm_backward_steps(lib, g_out, w_hh, c_all, gates, dones, T, **kw)
            torch.cuda.synchronize()
            print("lstm backward, 4 steps, %s: %.1f us per step" % (name, (time.perf_counter() - t0) / 80 * 1e6))
    assert lib.vine_lstm_step_backward_mfma(B + 1, H, g_out.data_ptr(), T * H, None, 0, None, 0, None, None, 0,
                                            gates.data_ptr(), c_all[1].data_ptr(), c_all[0].data_ptr(), None, 0,
                                            dG_b.data_ptr(), T * 4 * H, c_all[2].data_ptr(), None, None,
                                            torch.cuda.current_stream().cuda_stream) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("low_precision", [False, True])
def test_lstm_sequence_kernels_match_float64_autograd(low_precision):
    """The persistent LSTM kernels against an INDEPENDENT reference: the dones-masked LSTM recurrence written out in
    float64 torch on the same bf16-rounded operands, its autograd giving the gradients w.r.t. the gate pre-activations
    (= dG) -- not against this repo's own step kernels.  Tolerances = bf16 rounding of the stored activations / dG."""
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    lib = fused._lib()
    B, T, H, width, wpad = 256, 4, 256, 92, 96
    bf = fused.lp_dtype()
    xfull = torch.zeros(B * T, wpad, device=dev, dtype=bf)
    xfull[:, :width] = (torch.randn(B * T, width, device=dev) * 0.7).to(bf)
    w_ih = (torch.randn(4 * H, width, device=dev) / 10).to(bf)
    w_hh = (torch.randn(4 * H, H, device=dev) / 16).to(bf)
    bias = torch.randn(4 * H, device=dev) * 0.1
    h0, c0 = torch.randn(B, H, device=dev) * 0.5, torch.randn(B, H, device=dev) * 0.5
    dones = (torch.rand(B * T, device=dev) < 0.25).to(torch.uint8)
    wtile, whh_tiled = torch.empty(4 * H * (wpad + H), device=dev, dtype=bf), torch.empty(4 * H * H, device=dev, dtype=bf)
    prep = fused.CopyBatch()
    prep.add_lstm_tiles(w_ih, w_hh, wpad, wtile, whh_tiled)
    prep.flush(xfull)
    cdt = bf if low_precision else torch.float32
    bufs = fused._lstm_state_buffers(xfull, w_hh, h0, c0, dones, T, True, c_dtype=cdt)
    c_last = torch.empty(B, H, device=dev) if low_precision else None
    out, c_all, gates, hp = fused._lstm_forward_steps(lib, xfull, None, w_hh, bias, h0, c0, dones, T, True, buffers=bufs,
                                                      c0_direct=c0, wtile=wtile, c_last=c_last)
    g_out = torch.randn(B * T, H, device=dev) * 0.1
    g_in = g_out.to(bf) if low_precision else g_out
    dG, part = fused._lstm_backward_steps(lib, g_in, w_hh, c_all, gates, dones, T, c0_direct=c0, w_hh_tiled=whh_tiled,
                                          c_last=c_last)
    torch.cuda.synchronize()
    # float64 reference (rl_games' LSTMWithDones: the state is zeroed where the PREVIOUS step was terminal)
    xd = xfull[:, :width].double().view(B, T, width)
    Wi, Wh, bd = w_ih.double(), w_hh.double(), bias.double()
    keep = 1.0 - dones.view(B, T).double()
    h, c = h0.double(), c0.double()
    zs, hs = [], []
    for t in range(T):
        h, c = h * keep[:, t:t + 1], c * keep[:, t:t + 1]
        # the kernels feed the recurrent product with the bf16-rounded masked state
        z = xd[:, t] @ Wi.t() + h.to(bf).double() @ Wh.t() + bd
        z.requires_grad_(True)
        z.retain_grad()
        i, f, g, o = z.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        zs.append(z); hs.append(h)
    ref_out = torch.stack(hs, 1).reshape(B * T, H)
    assert float((out.double() - ref_out.detach()).abs().max()) < 2e-5 * 50      # fp32 arithmetic on identical operands
    assert float(((c_last if low_precision else c_all[T]).double() - c.detach()).abs().max()) < 1e-3
    # backward: cut the graph at the pre-activations (h_{t-1} enters the next product detached-and-rounded, so the
    # recurrent path is propagated by hand exactly as the kernel does: dh_{t-1} += keep_t * (dG_t @ w_hh))
    dGs = [None] * T
    dh_next = torch.zeros(B, H, device=dev, dtype=torch.float64)
    dc_next = torch.zeros(B, H, device=dev, dtype=torch.float64)
    go = (g_in.double() if low_precision else g_out.double()).view(B, T, H)
    hcs = []
    h, c = h0.double(), c0.double()
    for t in range(T):          # recompute with leaves per step
        hm, cm = (h * keep[:, t:t + 1]).detach(), (c * keep[:, t:t + 1]).detach().requires_grad_(True)
        z = (xd[:, t] @ Wi.t() + hm.to(bf).double() @ Wh.t() + bd).detach().requires_grad_(True)
        i, f, g, o = z.chunk(4, 1)
        c = torch.sigmoid(f) * cm + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        hcs.append((z, cm, h, c))
    for t in reversed(range(T)):
        z, cm, h, c = hcs[t]
        gz, gc = torch.autograd.grad([h, c], [z, cm], [go[:, t] + dh_next, dc_next])
        dGs[t] = gz
        dc_next = gc * keep[:, t:t + 1]
        # the kernel multiplies the bf16-rounded dG by w_hh
        dh_next = (gz.to(bf).double() @ Wh) * keep[:, t:t + 1]
    ref_dG = torch.stack(dGs, 1).reshape(B * T, 4 * H)
    scale = float(ref_dG.abs().max())
    err = (dG.double() - ref_dG).abs()
    tol_max, tol_mean = (3e-2, 2e-3) if low_precision else (1.5e-2, 1e-3)
    assert float(err.max()) < tol_max * scale and float(err.mean()) < tol_mean * scale, (float(err.max()) / scale, float(err.mean()) / scale)
    sb = part.double().sum(0)
    assert float((sb - ref_dG.sum(0)).abs().max()) < 2e-2 * float(ref_dG.sum(0).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,use_dones", [(8192, True), (96, False), (32, True)])
def test_lstm_sequence_kernels_equal_step_kernels(B, use_dones):
    """The persistent sequence kernels (ONE launch for all T steps: h_t / dG_t in LDS, c_t / dc_t in registers, weights
    streamed from the fragment-ordered copy) against the per-step matrix-core kernels they replace.  Same products in
    the same accumulation order and the same pointwise formulas: every output must be BIT-identical (the bias-gradient
    partial sums are folded in another order: tolerance)."""
    import time
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    lib = fused._lib()
    st = torch.cuda.current_stream().cuda_stream
    T, H, width, wpad = 4, 256, 92, 96
    bf = fused.lp_dtype()
    xfull = torch.zeros(B * T, wpad, device=dev, dtype=bf)
    xfull[:, :width] = (torch.randn(B * T, width, device=dev) * 0.7).to(bf)
    w_ih = (torch.randn(4 * H, width, device=dev) / 10).to(bf)
    w_hh = (torch.randn(4 * H, H, device=dev) / 16).to(bf)
    bias = torch.randn(4 * H, device=dev) * 0.1
    h0, c0 = torch.randn(B, H, device=dev) * 0.5, torch.randn(B, H, device=dev) * 0.5
    dones = (torch.rand(B * T, device=dev) < 0.25).to(torch.uint8) if use_dones else None
    wcat = torch.zeros(4 * H, wpad + H, device=dev, dtype=bf)
    wcat[:, :width], wcat[:, wpad:] = w_ih, w_hh
    # the fragment-ordered copies: through the batched operand-preparation launch and through the stand-alone entry
    wtile, wtile2 = torch.empty(4 * H * (wpad + H), device=dev, dtype=bf), torch.empty(4 * H * (wpad + H), device=dev, dtype=bf)
    whh_tiled, whh_tiled2 = torch.empty(4 * H * H, device=dev, dtype=bf), torch.empty(4 * H * H, device=dev, dtype=bf)
    prep = fused.CopyBatch()
    prep.add_lstm_tiles(w_ih, w_hh, wpad, wtile, whh_tiled)
    prep.flush(xfull)
    assert lib.vine_lstm_tile_weights(H, wpad + H, wcat.data_ptr(), wcat.stride(0), 0, wtile2.data_ptr(), st) == 0
    assert lib.vine_lstm_tile_weights(H, 4 * H, w_hh.data_ptr(), w_hh.stride(0), 1, whh_tiled2.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(wtile, wtile2) and torch.equal(whh_tiled, whh_tiled2)
    assert torch.equal(wtile.view(torch.int16).sort().values, wcat.flatten().view(torch.int16).sort().values)     # a permutation

    def forward(**kw):
        bufs = fused._lstm_state_buffers(xfull, w_hh, h0, c0, dones, T, True)
        if B % 64 and "wcat" in kw:
            return None
        return fused._lstm_forward_steps(lib, xfull, None, w_hh, bias, h0, c0, dones, T, True, buffers=bufs, c0_direct=c0, **kw)
    seq = forward(wtile=wtile)
    step = forward(wcat=wcat)
    torch.cuda.synchronize()
    if step is None:        # the step kernels need B % 64 == 0: compare with the plain composition instead
        ref_out, _, ref_c = fused._lstm_reference(xfull[:, :width].float(), w_ih.float(), w_hh.float(), bias, torch.zeros_like(bias),
                                                  h0, c0, None if dones is None else dones, T)
        assert float((seq[0] - ref_out).abs().max()) < 3e-2 and float((seq[1][T] - ref_c).abs().max()) < 3e-2
    else:
        for name, a, b in zip(("out", "c_all", "gates", "hp"), seq, step):
            a, b = (a[1:], b[1:]) if name == "c_all" else (a, b)
            assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))
    out, c_all, gates, hp = seq
    g_out = torch.randn(B * T, H, device=dev) * 0.1
    dG_s, part_s = fused._lstm_backward_steps(lib, g_out, w_hh, c_all, gates, dones, T, c0_direct=c0, w_hh_tiled=whh_tiled)
    if B % 64 == 0:
        dG_p, part_p = fused._lstm_backward_steps(lib, g_out, w_hh, c_all, gates, dones, T, c0_direct=c0,
                                                  w_hh_t=w_hh.t().contiguous())
    else:
        dG_p, part_p = fused._lstm_backward_steps(lib, g_out, w_hh, c_all, gates, dones, T, c0_direct=c0)
    torch.cuda.synchronize()
    assert part_s.shape == (B // 32, 4 * H)
    if B % 64 == 0:
        assert torch.equal(dG_s, dG_p), float((dG_s.float() - dG_p.float()).abs().max())
    else:
        assert float((dG_s.float() - dG_p.float()).abs().max()) <= 1e-2 * float(dG_p.float().abs().max())
    sa, sb = part_s.sum(0), part_p.view(-1, 4 * H).sum(0)
    assert float((sa - sb).abs().max()) <= 1e-4 * float(sb.abs().max()) + 1e-5, float((sa - sb).abs().max())
    # the bias partial sums ARE the column sums of the fp32 gate gradients: against the bf16 result within its rounding
    assert float((sa - dG_s.float().sum(0)).abs().max()) <= 2e-2 * float(sa.abs().max())
    if B == 8192:
        for name, f in (("forward, one launch per step", lambda: forward(wcat=wcat)), ("forward, persistent", lambda: forward(wtile=wtile)),
                        ("backward, one launch per step", lambda: fused._lstm_backward_steps(
                            lib, g_out, w_hh, c_all, gates, dones, T, c0_direct=c0, w_hh_t=w_hh.t().contiguous())),
                        ("backward, persistent", lambda: fused._lstm_backward_steps(
                            lib, g_out, w_hh, c_all, gates, dones, T, c0_direct=c0, w_hh_tiled=whh_tiled))):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            print("lstm %s: %.1f us per 4-step sequence" % (name, (time.perf_counter() - t0) / 20 * 1e6))
    # low-precision store of the backward-only data (the update's default): cell states c_1 .. c_{T-1} as bfloat16 + c_T in
    # fp32, hidden-state gradient as bfloat16 -- the forward outputs are untouched, dG moves within bf16 rounding
    bufs_lp = fused._lstm_state_buffers(xfull, w_hh, h0, c0, dones, T, True, c_dtype=bf)
    c_last = torch.empty(B, H, device=dev)
    out_lp, c_lp, gates_lp, hp_lp = fused._lstm_forward_steps(lib, xfull, None, w_hh, bias, h0, c0, dones, T, True, buffers=bufs_lp,
                                                              c0_direct=c0, wtile=wtile, c_last=c_last)
    torch.cuda.synchronize()
    assert torch.equal(out_lp, out) and torch.equal(gates_lp, gates) and torch.equal(hp_lp, hp)
    assert torch.equal(c_last, c_all[T]) and torch.equal(c_lp[1:T], c_all[1:T].to(bf))
    # ... and with the masked initial state formed by the kernel itself from the fp32 state (hp slot 0 is then an OUTPUT)
    bufs_h = fused._lstm_state_buffers(xfull, w_hh, h0, c0, dones, T, True, c_dtype=bf)
    bufs_h[3][:, 0] = float("nan")
    out_h, c_h, gates_h, hp_h = fused._lstm_forward_steps(lib, xfull, None, w_hh, bias, h0, c0, dones, T, True, buffers=bufs_h,
                                                          c0_direct=c0, wtile=wtile, c_last=c_last, h0_direct=h0)
    torch.cuda.synchronize()
    assert torch.equal(out_h, out) and torch.equal(gates_h, gates) and torch.equal(hp_h, hp)
    dG_lp, part_lp = fused._lstm_backward_steps(lib, g_out.to(bf), w_hh, c_lp, gates, dones, T, c0_direct=c0,
                                                w_hh_tiled=whh_tiled, c_last=c_last)
    torch.cuda.synchronize()
    scale = float(dG_s.float().abs().max())
    assert float((dG_lp.float() - dG_s.float()).abs().max()) <= 2e-2 * scale
    assert float((dG_lp.float() - dG_s.float()).abs().mean()) <= 2e-3 * scale
    # unsupported shapes are refused, not mis-run
    assert lib.vine_lstm_seq_forward_mfma(B + 1, T, H, wpad, xfull.data_ptr(), wpad, hp.data_ptr(), T * H, wtile.data_ptr(),
                                          bias.data_ptr(), c0.data_ptr(), None, out.data_ptr(), c_all.data_ptr(), None, 0, None,
                                          None, st) == -2
    assert lib.vine_lstm_seq_backward_mfma(B, 9, H, g_out.data_ptr(), whh_tiled.data_ptr(), gates.data_ptr(), c_all.data_ptr(),
                                           c0.data_ptr(), None, dG_s.data_ptr(), None, 0, None, 0, st) == -2


@pytest.mark.gpu
def test_weight_grad_mfma_matches_float64_product(monkeypatch):
    """dy^T x on the matrix cores (transposed LDS reads, row slices + deterministic column sums) for every weight of the
    default network, against the float64 product of the same bf16 operands; operands that are column blocks of a
    wider padded buffer (the LSTM operand buffer) included."""
    import time
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    bf = fused.lp_dtype()
    n = 32768
    monkeypatch.setattr(fused, "WGRAD_MAX_OUT", 1 << 30)          # the kernel is opt-in (see fused.WGRAD_MAX_OUT)
    xfull = (torch.randn(n, 96, device=dev) * 0.5).to(bf)
    xfull[:, 90:] = float("nan")                       # pad columns are read but must never reach an output
    cases = [("W1 [256,26]", 256, xfull[:, 64:90]), ("W2 [128,256]", 128, (torch.randn(n, 256, device=dev)).to(bf)),
             ("W3 [64,128]", 64, torch.randn(n, 128, device=dev).to(bf)), ("w_ih [1024,90]", 1024, xfull[:, :90]),
             ("w_hh [1024,256]", 1024, torch.randn(n, 256, device=dev).to(bf))]
    for name, M, x in cases:
        dy = (torch.randn(n, M, device=dev) * 0.1).to(bf)
        assert fused._wgrad_plan(dy, x) is not None, name
        out = fused.weight_grad(dy, x)
        ref = dy.double().t() @ x.double()
        err = float((out.double() - ref).abs().max()) / float(ref.abs().max())
        assert out.shape == ref.shape and err < 2e-5, (name, err)
        again = fused.weight_grad(dy, x)
        assert torch.equal(out, again)                 # fixed summation order
        res = []
        for f in (lambda: fused.weight_grad(dy, x), lambda: fused.splitk_tn(dy, x)):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                f()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 30 * 1e6)
        print("weight gradient %-16s mfma %6.1f us   split-K bmm + column sums %6.1f us" % (name, res[0], res[1]))
    # small / direct (one slice) and the fallback for unsupported operands
    dy, x = torch.randn(256, 64, device=dev).to(bf), torch.randn(256, 128, device=dev).to(bf)
    assert fused._wgrad_plan(dy, x) == (128, 1)
    assert float((fused.weight_grad(dy, x).double() - dy.double().t() @ x.double()).abs().max()) < 1e-3
    x26 = torch.randn(256, 26, device=dev).to(bf)                                   # unpadded 26-column rows
    assert fused._wgrad_plan(dy, x26) is None
    monkeypatch.setattr(fused, "WGRAD_MAX_OUT", 0)
    assert fused._wgrad_plan(dy, x) is None                                         # default: off
    assert float((fused.weight_grad(dy, x26).double() - dy.double().t() @ x26.double()).abs().max()) < 1e-3
    lib = fused._lib()
    assert lib.vine_weight_grad_mfma(256, 48, 128, 128, dy.data_ptr(), 64, x.data_ptr(), 128, 1, x.data_ptr(), None) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [True, False])
@pytest.mark.parametrize("n,width", [(32768, 92), (4096, 82)])
def test_weight_grad_cat_matches_float64_products(n, width, wide, monkeypatch):
    """The LSTM's two weight gradients from ONE pass over dG (vine_weight_grad_cat_mfma: one 128 x 352 output tile per
    workgroup, or 64 x 176 tiles, over the virtual [x | h]; transposed LDS reads, row slices + deterministic column sums)
    against the float64 products of the same bf16 operands; the x block is a column block of the 96-column operand
    buffer whose pad columns hold NaN (read, never stored)."""
    import time
    monkeypatch.setattr(fused, "WGRAD_CAT_WIDE", wide)
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    bf = fused.lp_dtype()
    H, M = 256, 1024
    xfull = (torch.randn(n, 96, device=dev) * 0.5).to(bf)
    xfull[:, width:] = float("nan")
    x1 = xfull[:, :width]
    hp = torch.randn(n, H, device=dev).to(bf)
    dG = (torch.randn(n, M, device=dev) * 0.1).to(bf)
    o1, o2 = torch.empty(M, width, device=dev), torch.empty(M, H, device=dev)
    assert fused.weight_grad_cat(dG, x1, hp, o1, o2)
    r1, r2 = dG.double().t() @ x1.double(), dG.double().t() @ hp.double()
    for name, o, r in (("dW_ih", o1, r1), ("dW_hh", o2, r2)):
        err = float((o.double() - r).abs().max()) / float(r.abs().max())
        assert torch.isfinite(o).all() and err < 2e-5, (name, err)
    a1, a2 = torch.empty_like(o1), torch.empty_like(o2)
    assert fused.weight_grad_cat(dG, x1, hp, a1, a2) and torch.equal(a1, o1) and torch.equal(a2, o2)     # fixed summation order
    if n == 32768:
        for name, f in (("one pass over dG, %s tiles" % ("128 x 352" if wide else "64 x 176"), lambda: fused.weight_grad_cat(dG, x1, hp, o1, o2)),
                        ("two split-K products (hipBLASLt)", lambda: (fused.splitk_tn(dG, x1, out=o1), fused.splitk_tn(dG, hp, out=o2)))):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                f()
            torch.cuda.synchronize()
            print("LSTM weight gradients, %s: %.1f us" % (name, (time.perf_counter() - t0) / 30 * 1e6))
    # shapes outside the kernel's family are refused: callers fall back to library products
    assert not fused.weight_grad_cat(dG[:, :992], x1, hp, torch.empty(992, width, device=dev), torch.empty(992, H, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(8192, 4), (512, 8), (4096, 1), (64, 2)])
def test_hidden_states_stored_once(B, T):
    """"h once": the persistent forward kernel writes ONE 16-bit, unmasked copy of the hidden states (+ the 16-bit h0)
    ([B, T + 1, H], slot 0 = h0) instead of fp32 for the LayerNorm and masked 16-bit for the recurrent weight gradient,
    and vine_weight_grad_cat_seq_mfma masks that copy itself.  Against the two-copy form, bit for bit: the 16-bit
    states are the rounded fp32 ones, every other forward output is unchanged, and the weight gradients' partial sums are
    those of vine_weight_grad_cat_mfma on the masked tensor."""
    dev = torch.device("cuda:0")
    torch.manual_seed(23)
    lib = fused._lib()
    H, width, wpad = 256, 92, 96
    bf = fused.lp_dtype()
    xfull = torch.zeros(B * T, wpad, device=dev, dtype=bf)
    xfull[:, :width] = (torch.randn(B * T, width, device=dev) * 0.7).to(bf)
    w_ih = (torch.randn(4 * H, width, device=dev) / 10).to(bf)
    w_hh = (torch.randn(4 * H, H, device=dev) / 16).to(bf)
    bias = torch.randn(4 * H, device=dev) * 0.1
    h0, c0 = torch.randn(B, H, device=dev) * 0.5, torch.randn(B, H, device=dev) * 0.5
    dones = (torch.rand(B * T, device=dev) < 0.25).to(torch.uint8)
    wtile, whh_tiled = torch.empty(4 * H * (wpad + H), device=dev, dtype=bf), torch.empty(4 * H * H, device=dev, dtype=bf)
    prep = fused.CopyBatch()
    prep.add_lstm_tiles(w_ih, w_hh, wpad, wtile, whh_tiled)
    prep.flush(xfull)

    def forward(h_once):
        bufs = fused._lstm_state_buffers(xfull, w_hh, h0, c0, dones, T, True, prep=fused.CopyBatch(), copy_c0=False, c_dtype=bf,
                                         mask_h0=False, h_once=h_once)
        for b in bufs:
            if b is not None:
                b.view(torch.int16 if b.dtype == bf else torch.int32).fill_(-1)      # NaN patterns: every element must be written
        c_last = torch.empty(B, H, device=dev)
        out, c_all, gates, hp = fused._lstm_forward_steps(lib, xfull, None, w_hh, bias, h0, c0, dones, T, True, buffers=bufs,
                                                          c0_direct=c0, wtile=wtile, c_last=c_last, h0_direct=h0)
        return out, c_all, gates, hp, c_last
    out2, c2, g2, hp2, cl2 = forward(False)
    out1, c1, g1, hp1, cl1 = forward(True)
    torch.cuda.synchronize()
    assert out1.dtype == bf and out1.shape == (B * (T + 1), H) and hp1 is None
    slots = out1.view(B, T + 1, H)
    assert torch.equal(slots[:, 1:], out2.to(bf).view(B, T, H)) and torch.equal(slots[:, 0], h0.to(bf))
    assert torch.equal(g1, g2) and torch.equal(c1[1:T], c2[1:T]) and torch.equal(cl1, cl2)
    # the operand the recurrent weight gradient needs, rebuilt from the one copy, is the masked tensor of the two-copy form
    assert torch.equal(fused.masked_previous_hidden(out1, dones, T), hp2.view(B * T, H))
    n, M = B * T, 4 * H
    dG = (torch.randn(n, M, device=dev) * 0.1).to(bf)
    o1, o2, a1, a2 = (torch.empty(M, w_, device=dev) for w_ in (width, H, width, H))
    plan = fused._wgrad_cat_plan(dG, xfull[:, :width], hp2.view(n, H), o1, o2)
    if plan is None or plan[6] not in (21, 22):      # (too few rows for the wide tile: the forward half was the test)
        assert not fused.weight_grad_cat(dG, xfull[:, :width], out1, a1, a2, seq=(dones, T))
        return
    assert fused.weight_grad_cat(dG, xfull[:, :width], hp2.view(n, H), o1, o2)
    assert fused.weight_grad_cat(dG, xfull[:, :width], out1, a1, a2, seq=(dones, T))
    torch.cuda.synchronize()
    assert torch.equal(a1, o1) and torch.equal(a2, o2)
    # a sequence length the kernel's stage layout does not cover is refused (callers build the masked tensor in torch)
    assert not fused.weight_grad_cat(dG, xfull[:, :width], out1, a1, a2, seq=(dones, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,stride,off", [(128, 256, 256, 0), (64, 128, 128, 0), (256, 28, 96, 64), (256, 18, 32, 0)])
def test_weight_grad_cat_single_operand(M, N, stride, off):
    """The MLP's weight gradients through the same kernel (no first operand): [128, 256] and [64, 128] with 128-wide
    output tiles, the first layer's [256, num_obs] with a 32-wide tile over a column block of the padded operand buffer."""
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    bf = fused.lp_dtype()
    n = 32768
    full = torch.full((n, stride), float("nan"), device=dev, dtype=bf)
    full[:, off:off + N] = (torch.randn(n, N, device=dev) * 0.5).to(bf)
    x = full[:, off:off + N]
    dy = (torch.randn(n, M, device=dev) * 0.1).to(bf)
    out = torch.empty(M, N, device=dev)
    assert fused.weight_grad_cat(dy, None, x, None, out)
    ref = dy.double().t() @ x.double()
    err = float((out.double() - ref).abs().max()) / float(ref.abs().max())
    assert torch.isfinite(out).all() and err < 2e-5, err
    assert torch.equal(fused.weight_grad(dy, x), out)                  # the default route of bf16 operands


@pytest.mark.gpu
@pytest.mark.parametrize("n,raw", [(32768, True), (4096, True), (4096, False)])
def test_mlp3_kernels_match_float64_autograd(n, raw):
    """The one-launch MLP (vine_mlp3_elu_mfma: observation normalisation + three Linear+ELU layers, activations carried
    in registers) and its one-launch backward (vine_mlp3_bwd_elu_mfma) against float64 torch on the same bf16-rounded
    operands: activations, the normalised observation block, the three pre-activation gradients and the per-workgroup
    bias partial sums."""
    dev = torch.device("cuda:0")
    torch.manual_seed(21)
    lib = fused._lib()
    st = torch.cuda.current_stream().cuda_stream
    bf = fused.lp_dtype()
    F_in, U, K0 = 28, 64, 1024
    obs = torch.randn(n, F_in, device=dev) * 3.0 + 0.5
    mean, var = torch.randn(F_in, device=dev, dtype=torch.float64) * 0.3, torch.rand(F_in, device=dev, dtype=torch.float64) + 0.5
    xn = ((obs - mean.float()) / torch.sqrt(var.float() + 1e-5)).clamp(-5, 5)
    xfull = torch.full((n, 96), float("nan"), device=dev, dtype=bf)
    if not raw:
        xfull[:, U:U + F_in] = xn.to(bf)
        xfull[:, U + F_in:] = 0
    W1, W2, W3 = ((torch.randn(o, i, device=dev) / i ** 0.5).to(bf) for o, i in ((256, F_in), (128, 256), (64, 128)))
    b1, b2, b3 = (torch.randn(o, device=dev) * 0.1 for o in (256, 128, 64))
    w1p = torch.zeros(256, 32, device=dev, dtype=bf)
    w1p[:, :F_in] = W1
    act1, act2 = torch.empty(n, 256, device=dev, dtype=bf), torch.empty(n, 128, device=dev, dtype=bf)
    assert lib.vine_mlp3_elu_mfma(n, xfull.data_ptr() + 2 * U, 96, obs.data_ptr() if raw else None, F_in,
                                  mean.data_ptr() if raw else None, var.data_ptr() if raw else None, 1e-5, 5.0, w1p.data_ptr(),
                                  b1.data_ptr(), 256, W2.data_ptr(), 256, b2.data_ptr(), 128, W3.data_ptr(), 128, b3.data_ptr(), 64,
                                  1.0, act1.data_ptr(), act2.data_ptr(), xfull.data_ptr(), 96, st) == 0
    torch.cuda.synchronize()
    x0 = xfull[:, U:U + F_in]
    assert torch.equal(x0, xn.to(bf)) and (xfull[:, U + F_in:].float() == 0).all()
    elu = torch.nn.functional.elu
    a1 = elu(x0.double() @ W1.double().t() + b1.double())
    a2 = elu(act1.double() @ W2.double().t() + b2.double())            # each layer from the kernel's own (rounded) input
    a3 = elu(act2.double() @ W3.double().t() + b3.double())
    for name, got, ref in (("act1", act1, a1), ("act2", act2, a2), ("act3", xfull[:, :U], a3)):
        err = (got.double() - ref).abs() / (ref.abs() + 1.0)
        assert float(err.max()) < 6e-3, (name, float(err.max()))            # one bf16 rounding of the output
    # ---- backward
    dG = (torch.randn(n, K0, device=dev) * 0.05).to(bf)
    Wt0 = (torch.randn(U, K0, device=dev) / K0 ** 0.5).to(bf)               # the MLP block of w_ih, transposed
    Wt1, Wt2 = W3.t().contiguous(), W2.t().contiguous()                      # [128, 64], [256, 128]
    gz3, gz2, gz1 = (torch.empty(n, c, device=dev, dtype=bf) for c in (64, 128, 256))
    R = 128 if (n % 128 == 0 and n >= 32768) else 64
    p3, p2, p1 = (torch.empty(n // R, c, device=dev) for c in (64, 128, 256))
    assert lib.vine_mlp3_bwd_elu_mfma(n, dG.data_ptr(), K0, K0, Wt0.data_ptr(), K0, Wt1.data_ptr(), 64, Wt2.data_ptr(), 128,
                                      xfull.data_ptr(), 96, act2.data_ptr(), act1.data_ptr(), 64, 128, 256, 1.0,
                                      gz3.data_ptr(), gz2.data_ptr(), gz1.data_ptr(), p3.data_ptr(), p2.data_ptr(),
                                      p1.data_ptr(), st) == 0
    torch.cuda.synchronize()
    dact = lambda a: torch.where(a.double() > 0, torch.ones_like(a, dtype=torch.float64), a.double() + 1.0)      # ELU' from the output
    r3 = (dG.double() @ Wt0.double().t()) * dact(xfull[:, :U])
    r2 = (gz3.double() @ Wt1.double().t()) * dact(act2)
    r1 = (gz2.double() @ Wt2.double().t()) * dact(act1)
    for name, got, ref, part in (("gz3", gz3, r3, p3), ("gz2", gz2, r2, p2), ("gz1", gz1, r1, p1)):
        scale = float(ref.abs().max())
        assert float((got.double() - ref).abs().max()) < 6e-3 * scale, name
        sums, rs = part.double().sum(0), ref.sum(0)
        assert float((sums - rs).abs().max()) < 1e-3 * float(rs.abs().max()) + 1e-4 * scale, name
    assert lib.vine_mlp3_elu_mfma(n + 8, xfull.data_ptr() + 2 * U, 96, None, 0, None, None, 0.0, 0.0, w1p.data_ptr(), b1.data_ptr(),
                                  256, W2.data_ptr(), 256, b2.data_ptr(), 128, W3.data_ptr(), 128, b3.data_ptr(), 64, 1.0, None,
                                  None, xfull.data_ptr(), 96, st) == -2


@pytest.mark.gpu
def test_weight_grad_group_equals_single_launches(monkeypatch):
    """The three MLP weight gradients of the update in ONE launch (vine_weight_grad_group) are bit-identical to three
    single launches of the same kernel family, and match the float64 products."""
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    bf = fused.lp_dtype()
    n = 32768
    xfull = torch.full((n, 96), float("nan"), device=dev, dtype=bf)
    xfull[:, 64:92] = (torch.randn(n, 28, device=dev) * 0.5).to(bf)
    cases = [((torch.randn(n, 64, device=dev) * 0.1).to(bf), (torch.randn(n, 128, device=dev) * 0.5).to(bf)),
             ((torch.randn(n, 128, device=dev) * 0.1).to(bf), (torch.randn(n, 256, device=dev) * 0.5).to(bf)),
             ((torch.randn(n, 256, device=dev) * 0.1).to(bf), xfull[:, 64:92])]
    monkeypatch.setattr(fused, "WGRAD_GROUP_WGS", fused.WGRAD_CAT_WGS)      # same row slices as a launch of its own
    batch, group = fused.ColumnSumBatch(), fused.WeightGradGroup()
    outs = [torch.empty(dy.shape[1], x.shape[1], device=dev) for dy, x in cases]
    for (dy, x), o in zip(cases, outs):
        assert group.add(dy, x, o, batch)
    group.flush()
    batch.flush(outs[0])
    for (dy, x), o in zip(cases, outs):
        single, b1 = torch.empty_like(o), fused.ColumnSumBatch()
        assert fused.weight_grad_cat(dy, None, x, None, single, batch=b1)
        b1.flush(single)
        assert torch.equal(single, o)
        ref = dy.double().t() @ x.double()
        assert float((o.double() - ref).abs().max()) / float(ref.abs().max()) < 2e-5
    assert not fused.WeightGradGroup().add(cases[0][0], cases[0][1], outs[0], None)      # needs the column-sum batch


@pytest.mark.gpu
@pytest.mark.parametrize("F_", [28, 18, 1])
def test_running_mean_std_kernels_match_torch_composition(F_):
    """vine_rms_update + vine_normalize_obs (float64 statistics, two-stage sums) against the module's torch path."""
    from vine_robot_isaacgymenvs_amd.learning.running_mean_std import RunningMeanStd
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    a, b = RunningMeanStd((F_,)).to(dev), RunningMeanStd((F_,)).to(dev)
    b._use_kernels = lambda x: False                       # reference: the torch composition
    a.train(); b.train()
    for n in (32768, 4099, 2):
        x = torch.randn(n, F_, device=dev) * torch.linspace(0.1, 9.0, F_, device=dev) + 3.0
        ya, yb = a(x), b(x)
        assert float((ya - yb).abs().max()) < 2e-5       # fp32 batch mean of the torch path / column std 0.1
    for name in ("running_mean", "running_var", "count"):
        u, v = getattr(a, name), getattr(b, name)
        # the torch path reduces the batch in float32 before the float64 merge; the kernels accumulate in float64
        assert float(((u - v).abs() / (v.abs() + 1e-30)).max()) < 2e-6, name
    a.eval(); b.eval()
    x = torch.randn(1000, F_, device=dev) * 20
    ya, yb = a(x), b(x)
    assert float((ya - yb).abs().max()) < 2e-5 and float(ya.abs().max()) == 5.0
    assert torch.equal(a.count, b.count)                   # eval mode leaves the statistics alone


@pytest.mark.gpu
@pytest.mark.parametrize("F_,n,k", [(28, 32768, 8), (18, 2048, 4), (28, 130, 3), (1, 2, 2)])
def test_running_mean_std_updates_of_a_mini_epoch_at_once_are_bit_identical(F_, n, k):
    """vine_rms_update_multi (the k training-mode updates of PPO's first mini-epoch in three launches at its head) against k
    calls of vine_rms_update on the slices: every intermediate state (what step i normalises with) and the final module
    state are bit-identical."""
    from vine_robot_isaacgymenvs_amd.learning.running_mean_std import RunningMeanStd
    dev = torch.device("cuda:0")
    torch.manual_seed(F_ + n)
    a, b = RunningMeanStd((F_,)).to(dev), RunningMeanStd((F_,)).to(dev)
    a.train(); b.train()
    warm = torch.randn(777, F_, device=dev) * 3.0 - 1.0
    a.update_kernels(warm); b.update_kernels(warm)            # (not the initial state)
    x = torch.randn(k * n, F_, device=dev) * torch.linspace(0.1, 9.0, F_, device=dev) + 3.0
    mean, var = a.update_kernels_multi(x, k)
    for i in range(k):
        b.update_kernels(x[i * n:(i + 1) * n])
        assert torch.equal(mean[i], b.running_mean) and torch.equal(var[i], b.running_var), i
    for name in ("running_mean", "running_var", "count"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    a.eval()
    assert a.update_kernels_multi(x, k) is None and torch.equal(a.count, b.count)      # eval mode: nothing happens
    from vine_robot_isaacgymenvs_amd import native
    assert native.load().vine_rms_update_multi(0, n, F_, x.data_ptr(), None, None, None, None, None, None, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("n,K,N", [(32768, 256, 128), (16384, 128, 64), (64, 32, 64)])
def test_linear_elu_mfma_matches_gemm_plus_bias_elu(n, K, N):
    """Matrix-core Linear + bias + ELU against torch GEMM (same bf16 operands, fp32 out) + vine_bias_elu; output
    into a column block of a wider buffer; timing of both printed for the record."""
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    lib = fused._lib()
    st = torch.cuda.current_stream().cuda_stream
    bf = fused.lp_dtype()
    A = (torch.randn(n, K + 8, device=dev) * 0.7).to(bf)[:, :K]
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(bf)
    bias = torch.randn(N, device=dev) * 0.2
    wide = torch.zeros(n, N + 32, device=dev, dtype=bf)
    ref = torch.zeros(n, N + 32, device=dev, dtype=bf)

    def fusedk():
        assert lib.vine_linear_elu_mfma(n, N, K, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), bias.data_ptr(), 1.0,
                                        wide.data_ptr(), N + 32, st) == 0

    def unfused():
        z = torch.mm(A, W.t(), out_dtype=torch.float32)
        assert lib.vine_bias_elu(n, N, z.data_ptr(), bias.data_ptr(), 1.0, ref.data_ptr(), N + 32, 1, st) == 0

    fusedk(); unfused()
    torch.cuda.synchronize()
    d = (wide.float() - ref.float()).abs()
    assert float(d.max()) < 2e-2 and float((d > 0).float().mean()) < 0.02      # bf16 rounding boundaries only
    assert float(wide[:, N:].float().abs().max()) == 0.0
    times = []
    for f in (fusedk, unfused):
        for _ in range(5):
            f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            f()
        e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e) * 20)
    print("linear_elu n=%d K=%d N=%d: fused %.1f us, gemm+bias_elu %.1f us" % (n, K, N, times[0], times[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("n,K,N", [(32768, 128, 256), (16384, 64, 128), (64, 256, 64), (32768, 1024, 64), (128, 512, 128)])
def test_linear_bwd_elu_mfma_matches_gemm_plus_elu_backward(n, K, N):
    """gz = (G W) * elu'(a) and its per-workgroup column sums on the matrix cores against GEMM + vine_elu_backward."""
    from vine_robot_isaacgymenvs_amd.abi import PPO_PARTIAL_BLOCKS
    dev = torch.device("cuda:0")
    torch.manual_seed(12)
    lib = fused._lib()
    st = torch.cuda.current_stream().cuda_stream
    bf = fused.lp_dtype()
    G = (torch.randn(n, K, device=dev) / n).to(bf)
    W = (torch.randn(K, N, device=dev) / K ** 0.5).to(bf)                  # the layer's weight [out, in]
    a = torch.nn.functional.elu(torch.randn(n, N + 16, device=dev)).to(bf)[:, :N]
    gz, part = torch.empty(n, N, device=dev, dtype=bf), torch.empty(n // 64, N, device=dev)
    wt = W.t().contiguous()
    assert lib.vine_linear_bwd_elu_mfma(n, N, K, G.data_ptr(), K, wt.data_ptr(), K, a.data_ptr(), a.stride(0), 1.0,
                                        gz.data_ptr(), N, part.data_ptr(), st) == 0
    g = torch.mm(G, W, out_dtype=torch.float32)
    ref, rpart = torch.empty(n, N, device=dev, dtype=bf), torch.empty(PPO_PARTIAL_BLOCKS, N, device=dev)
    assert lib.vine_elu_backward(n, N, g.data_ptr(), N, a.data_ptr(), a.stride(0), 1.0, ref.data_ptr(), N, rpart.data_ptr(),
                                 1, 1, st) == 0
    torch.cuda.synchronize()
    scale = float(ref.float().abs().max())
    assert float((gz.float() - ref.float()).abs().max()) < 1e-2 * scale
    s1, s2 = part.sum(0), rpart.sum(0)
    assert float((s1 - s2).abs().max()) < 1e-3 * (float(s2.abs().max()) + 1e-12) + 1e-7
    if n >= 16384:
        def fusedk():
            lib.vine_linear_bwd_elu_mfma(n, N, K, G.data_ptr(), K, wt.data_ptr(), K, a.data_ptr(), a.stride(0), 1.0,
                                         gz.data_ptr(), N, part.data_ptr(), st)

        def unfused():
            g2 = torch.mm(G, W, out_dtype=torch.float32)
            lib.vine_elu_backward(n, N, g2.data_ptr(), N, a.data_ptr(), a.stride(0), 1.0, ref.data_ptr(), N, rpart.data_ptr(),
                                  1, 1, st)
        times = []
        for f in (fusedk, unfused):
            for _ in range(5):
                f()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                f()
            e.record()
            torch.cuda.synchronize()
            times.append(s.elapsed_time(e) * 20)
        print("linear_bwd_elu n=%d K=%d N=%d: fused %.1f us, gemm+elu_bwd %.1f us" % (n, K, N, times[0], times[1]))


@pytest.mark.gpu
def test_copy_batch_ops():
    """vine_copy_batched: copy / zero / transpose / fp32->bf16 cast / add / masked, strided views, 20 jobs (two launches)."""
    dev = torch.device("cuda:0")
    torch.manual_seed(13)
    bf = fused.lp_dtype()
    cb, checks = fused.CopyBatch(), []
    for rep in range(3):
        src = torch.randn(1024, 92, device=dev).to(bf)
        wide = torch.full((1024, 352), 7.0, device=dev, dtype=bf)
        cb.add(cb.COPY, wide[:, :92], src)
        cb.add(cb.ZERO, wide[:, 92:96])
        checks.append((wide[:, :92], src)); checks.append((wide[:, 92:96], torch.zeros(1024, 4, device=dev, dtype=bf)))
        w = torch.randn(128, 256, device=dev).to(bf)
        wt = torch.empty(256, 128, device=dev, dtype=bf)
        cb.add(cb.TRANSPOSE, wt, w)
        checks.append((wt, w.t()))
        x = torch.randn(4099, 28, device=dev)
        xb = torch.empty(4099, 96, device=dev, dtype=bf)[:, 64:92]
        cb.add(cb.CAST_BF16, xb, x)
        checks.append((xb, x.to(bf)))
        a, b = torch.randn(1024, device=dev), torch.randn(1024, device=dev)
        o = torch.empty(1024, device=dev)
        cb.add(cb.ADD, o, a, b)
        checks.append((o, a + b))
        h0 = torch.randn(512, 256, device=dev)
        dones = (torch.rand(512 * 4, device=dev) < 0.3).to(torch.uint8)
        hp = torch.empty(512, 4, 256, device=dev, dtype=bf)
        cb.add(cb.MASKED, hp[:, 0], h0, dones, aux=4)
        checks.append((hp[:, 0], (h0 * (1.0 - dones.view(512, 4)[:, 0:1].float())).to(bf)))
        odd = torch.randn(100, 27, device=dev)                      # 27 columns: the element-wise path
        odd_dst = torch.empty(100, 31, device=dev)[:, 2:29]
        cb.add(cb.COPY, odd_dst, odd)
        checks.append((odd_dst, odd))
        odd_b = torch.empty(100, 27, device=dev, dtype=bf)
        cb.add(cb.CAST_BF16, odd_b, odd)
        checks.append((odd_b, odd.to(bf)))
        c0, c1 = torch.randn(512, 256, device=dev), torch.empty(512, 256, device=dev)
        cb.add(cb.MASKED, c1, c0, None)
        checks.append((c1, c0))
    cb.flush(src)
    torch.cuda.synchronize()
    for got, want in checks:
        assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("width,wpad", [(92, 96), (64, 64), (82, 96)])
def test_copy_scatter_forms_write_the_same_bytes(width, wpad, monkeypatch):
    """Round 4: the LSTM weight tiles and the 16-bit transposes in their coalesced-load / scattered-store forms (copy ops
    8 - 10) against the gather forms (ops 6, 7, 2): identical destination bytes, zero pad columns included.  width 82 is not
    a multiple of 4: the batch falls back to the gather forms by itself."""
    dev = torch.device("cuda:0")
    lp = fused.lp_dtype()
    torch.manual_seed(width)
    H = 256
    w_ih = torch.randn(4 * H, width, device=dev).to(lp)
    w_hh = torch.randn(4 * H, H, device=dev).to(lp)
    w2 = torch.randn(128, 256, device=dev).to(lp)
    w3 = torch.randn(64, 128, device=dev).to(lp)
    outs = []
    for scatter in (False, True):
        monkeypatch.setattr(fused, "COPY_SCATTER", scatter)
        fwd = torch.full((4 * H * (wpad + H),), 7.0, device=dev, dtype=lp)
        bwd = torch.full((4 * H * H,), 7.0, device=dev, dtype=lp)
        t2, t3 = torch.empty(256, 128, device=dev, dtype=lp), torch.empty(128, 64, device=dev, dtype=lp)
        tih = torch.empty(64, 4 * H, device=dev, dtype=lp)
        cb = fused.CopyBatch()
        cb.add_lstm_tiles(w_ih, w_hh, wpad, fwd, bwd)
        cb.add(fused.CopyBatch.TRANSPOSE, t2, w2)
        cb.add(fused.CopyBatch.TRANSPOSE, t3, w3)
        cb.add(fused.CopyBatch.TRANSPOSE, tih, w_ih[:, :64])
        ops = [j[0] for j in cb.jobs]
        assert (8 in ops and 9 in ops) == (scatter and width % 4 == 0) and (10 in ops) == scatter
        cb.flush(fwd)
        torch.cuda.synchronize()
        outs.append((fwd, bwd, t2, t3, tih))
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert torch.equal(outs[1][2], w2.t()) and torch.equal(outs[1][4], w_ih[:, :64].t())


@pytest.mark.gpu
def test_gae_kernel_matches_reference_loop():
    """vine_gae against the Python loop (rl_games discount_values, next-nonterminal form)."""
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import discount_values
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    T, N = 16, 1000
    rew, val = torch.randn(T, N, 1, device=dev), torch.randn(T, N, 1, device=dev)
    dones = (torch.rand(T, N, device=dev) < 0.1).to(torch.uint8)
    last_d = (torch.rand(N, device=dev) < 0.1).to(torch.uint8)
    last_v = torch.randn(N, 1, device=dev)
    ref = discount_values(0.99, 0.95, last_d.float(), last_v, dones.float(), val, rew)
    adv, ret = torch.empty_like(rew), torch.empty_like(rew)
    assert fused._lib().vine_gae(T, N, rew.data_ptr(), val.data_ptr(), dones.data_ptr(), last_v.data_ptr(), last_d.data_ptr(),
                                 0.99, 0.95, adv.data_ptr(), ret.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    assert float((adv - ref).abs().max()) < 1e-5 and float((ret - (ref + val)).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("T,N,F", [(16, 1024, 28), (16, 192, 18), (6, 64, 28), (32, 320, 28)])
@pytest.mark.parametrize("norm_value,norm_adv", [(True, True), (False, True), (True, False)])
def test_dataset_assemble_matches_the_stock_composition(T, N, F, norm_value, norm_adv):
    """vine_dataset_assemble (three launches) against what the stock path does between the rollout and the first optimiser
    step: discount_values, returns = A + V, swap_and_flatten01 of every buffer, value_mean_std(values) then
    value_mean_std(returns) in training mode, advantages = returns - values normalised by mean / unbiased std.
    Transposed buffers bit-exact; normalised series to fp32 round-off; the float64 running statistics to 1e-12."""
    import ctypes as C
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import discount_values, swap_and_flatten01
    from vine_robot_isaacgymenvs_amd.learning.running_mean_std import RunningMeanStd
    dev = torch.device("cuda:0")
    torch.manual_seed(T * N + F)
    A = 2
    rew, val = torch.randn(T, N, 1, device=dev) * 0.3, torch.randn(T, N, 1, device=dev) * 2 + 0.5
    dones = (torch.rand(T, N, device=dev) < 0.1).to(torch.uint8)
    last_d = (torch.rand(N, device=dev) < 0.1).to(torch.uint8)
    last_v = torch.randn(N, 1, device=dev)
    bufs = {"obs": torch.randn(T, N, F, device=dev), "actions": torch.randn(T, N, A, device=dev),
            "neglogp": torch.randn(T, N, device=dev), "mu": torch.randn(T, N, A, device=dev),
            "sigma": torch.rand(T, N, A, device=dev)}
    vms = RunningMeanStd((1,)).to(dev)
    vms.running_mean.fill_(0.3); vms.running_var.fill_(1.7); vms.count.fill_(5000.0)     # mid-training statistics
    ref_vms = RunningMeanStd((1,)).to(dev)
    ref_vms.load_state_dict(vms.state_dict())
    # ---- the stock composition (A2CAgent.play_steps_rnn + prepare_dataset)
    advs = discount_values(0.99, 0.95, last_d.float(), last_v, dones.float(), val, rew)
    returns, values = swap_and_flatten01(advs + val), swap_and_flatten01(val)
    adv_ref = (returns - values).sum(dim=1)
    if norm_value:
        ref_vms.train()
        with torch.no_grad():       # (the torch composition, not the module's kernels: the reference of this test)
            ref_vms.update(values); v_ref = ref_vms.eval()(values)
            ref_vms.train(); ref_vms.update(returns); r_ref = ref_vms.eval()(returns)
    else:
        v_ref, r_ref = values, returns
    if norm_adv:
        adv_ref = (adv_ref - adv_ref.mean()) / (adv_ref.std() + 1e-8)
    # ---- the kernels
    n = N * T
    dv, dr, da = (torch.empty(n, 1, device=dev), torch.empty(n, 1, device=dev), torch.empty(n, device=dev))
    out = {"obs": torch.empty(n, F, device=dev), "actions": torch.empty(n, A, device=dev), "neglogp": torch.empty(n, device=dev),
           "mu": torch.empty(n, A, device=dev), "sigma": torch.empty(n, A, device=dev)}
    d_out = torch.empty(n, device=dev, dtype=torch.uint8)
    jobs = [(bufs[k], out[k], (bufs[k].shape[2] if bufs[k].dim() == 3 else 1), 4) for k in out] + [(dones, d_out, 1, 1)]
    k = len(jobs)
    scratch = torch.empty((N + 255) // 256 * 6 + 4, device=dev, dtype=torch.float64)
    pending = torch.zeros(3, device=dev, dtype=torch.float64)
    rc = fused._lib().vine_dataset_assemble(
        T, N, rew.data_ptr(), val.data_ptr(), dones.data_ptr(), last_v.data_ptr(), last_d.data_ptr(), 0.99, 0.95,
        vms.running_mean.data_ptr() if norm_value else None, vms.running_var.data_ptr() if norm_value else None,
        vms.count.data_ptr() if norm_value else None, float(vms.epsilon), int(norm_value), int(norm_adv), dv.data_ptr(),
        dr.data_ptr(), da.data_ptr(), k, (C.c_void_p * k)(*[j[0].data_ptr() for j in jobs]),
        (C.c_void_p * k)(*[j[1].data_ptr() for j in jobs]), (C.c_int32 * k)(*[int(j[2]) for j in jobs]),
        (C.c_int32 * k)(*[j[3] for j in jobs]), scratch.data_ptr(), pending.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    if norm_value:      # the module's statistics are read only; the update arrives in `pending` for the caller to commit
        assert float(vms.running_mean[0]) == 0.3 and float(vms.running_var[0]) == 1.7 and float(vms.count) == 5000.0
        vms.running_mean[0], vms.running_var[0] = pending[0], pending[1]
        vms.count.copy_(pending[2])
    for name in out:
        assert torch.equal(out[name], swap_and_flatten01(bufs[name])), name
    assert torch.equal(d_out, swap_and_flatten01(dones))
    torch.testing.assert_close(dv, v_ref, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(dr, r_ref, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(da, adv_ref, rtol=2e-5, atol=2e-5)
    if norm_value:
        for a, b in ((vms.running_mean, ref_vms.running_mean), (vms.running_var, ref_vms.running_var), (vms.count, ref_vms.count)):
            # (the reference forms its batch moments in float32: x.mean(0), x.var(0))
            assert abs(float(a.reshape(-1)[0]) - float(b.reshape(-1)[0])) <= 2e-6 * max(1.0, abs(float(b.reshape(-1)[0])))
    # shapes the kernel refuses are reported, not mangled
    assert fused._lib().vine_dataset_assemble(
        T, N + 1, rew.data_ptr(), val.data_ptr(), dones.data_ptr(), last_v.data_ptr(), last_d.data_ptr(), 0.99, 0.95, None, None,
        None, 0.0, 0, 0, dv.data_ptr(), dr.data_ptr(), da.data_ptr(), 0, None, None, None, None, scratch.data_ptr(), None,
        torch.cuda.current_stream().cuda_stream) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0.0, 0.1, 1234.567, -3.3e-3])
def test_dataset_assemble_constant_series_keeps_the_variance_non_negative(level):
    """ds_finalize forms the batch variance as (sum x^2 - n mean^2) / (n - 1); for a constant series (values
    early in training, all-zero rewards) that can round below zero, and the Chan merge would carry a negative running_var
    into the module and the checkpoint.  x.var() (the stock RunningMeanStd) is never negative: neither is the clamped form."""
    import ctypes as C
    dev = torch.device("cuda:0")
    T, N = 16, 1024
    rew = torch.zeros(T, N, 1, device=dev)
    val = torch.full((T, N, 1), level, device=dev)
    dones = torch.zeros(T, N, device=dev, dtype=torch.uint8)
    last_d = torch.zeros(N, device=dev, dtype=torch.uint8)
    last_v = torch.full((N, 1), level, device=dev)
    n = N * T
    dv, dr, da = torch.empty(n, 1, device=dev), torch.empty(n, 1, device=dev), torch.empty(n, device=dev)
    mean = torch.tensor([level], device=dev, dtype=torch.float64)
    var = torch.zeros(1, device=dev, dtype=torch.float64)              # a module that has only ever seen this constant
    cnt = torch.tensor(5000.0, device=dev, dtype=torch.float64)
    scratch = torch.empty((N + 255) // 256 * 6 + 4, device=dev, dtype=torch.float64)
    pending = torch.zeros(3, device=dev, dtype=torch.float64)
    rc = fused._lib().vine_dataset_assemble(
        T, N, rew.data_ptr(), val.data_ptr(), dones.data_ptr(), last_v.data_ptr(), last_d.data_ptr(), 1.0, 1.0,
        mean.data_ptr(), var.data_ptr(), cnt.data_ptr(), 1e-5, 1, 1, dv.data_ptr(), dr.data_ptr(), da.data_ptr(), 0, None, None,
        None, None, scratch.data_ptr(), pending.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert float(pending[1]) >= 0.0 and float(pending[2]) == 5000.0 + 2 * n
    assert abs(float(pending[0]) - level) <= 1e-6 * max(1.0, abs(level))
    assert torch.isfinite(dv).all() and torch.isfinite(dr).all() and torch.isfinite(da).all()


@pytest.mark.gpu
def test_splitk_linear_gradients():
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    x = torch.randn(32768, 92, device=dev, requires_grad=True)
    w = torch.randn(64, 92, device=dev, requires_grad=True)
    b = torch.randn(64, device=dev, requires_grad=True)
    g = torch.randn(32768, 64, device=dev)
    y = fused.linear(x, w, b)
    y.backward(g)
    got = [y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    for p in (x, w, b):
        p.grad = None
    y2 = torch.nn.functional.linear(x, w, b)
    y2.backward(g)
    for u, v in zip(got, [y2.detach(), x.grad, w.grad, b.grad]):
        assert float((u - v).abs().max()) / (float(v.abs().max()) + 1e-6) < 1e-4
    assert fused.splitk_tn(g, x.detach()).shape == (64, 92)


@pytest.mark.gpu
@pytest.mark.parametrize("clip_value", [True, False])
def test_ppo_loss_kernel_matches_autograd(clip_value):
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    n, A = 32768, 2
    mu = (torch.randn(n, A, device=dev) * 0.8).requires_grad_()      # some |mu| > 1.1: bound loss active
    logstd = torch.tensor([0.1, -0.3], device=dev, requires_grad=True)
    value = torch.randn(n, 1, device=dev, requires_grad=True)
    actions = mu.detach() + torch.randn(n, A, device=dev) * 0.9
    old_neglogp = torch.randn(n, device=dev) * 0.3 + 2.0
    adv = torch.randn(n, device=dev)
    old_values = value.detach() + torch.randn(n, 1, device=dev) * 0.3
    returns = torch.randn(n, 1, device=dev)
    old_mu = mu.detach() + torch.randn(n, A, device=dev) * 0.05
    old_sigma = torch.full((n, A), 1.05, device=dev)
    cfg = dict(e_clip=0.2, clip_value=clip_value, critic_coef=2.0, entropy_coef=0.01, bounds_coef=1e-4)
    loss, st = fused.ppo_loss_reference(mu, logstd, value, actions, old_neglogp, adv, old_values, returns, old_mu,
                                        old_sigma, **cfg)
    loss.backward()
    g_mu, g_val, g_ls, stats = fused.ppo_loss_fused(mu, logstd, value, actions, old_neglogp, adv, old_values, returns,
                                                    old_mu, old_sigma, **cfg)
    torch.cuda.synchronize()
    assert abs(float(stats[5]) - float(loss)) < 1e-4 * (1 + abs(float(loss)))
    for idx, key in ((0, "a_loss"), (1, "c_loss"), (2, "b_loss"), (3, "entropy"), (4, "kl")):
        assert abs(float(stats[idx]) - float(st[key])) < 1e-4 * (1 + abs(float(st[key]))), key
    for u, v, name in ((g_mu, mu.grad, "mu"), (g_val, value.grad, "value"), (g_ls, logstd.grad, "logstd")):
        assert float((u - v).abs().max()) / (float(v.abs().max()) + 1e-12) < 2e-4, name


@pytest.mark.gpu
@pytest.mark.parametrize("obs_dim,use_slots,mixed", [(28, False, False), (28, True, False), (18, True, False),
                                                    (28, True, True), (18, False, True)])
def test_fused_trunk_matches_float64_composition(obs_dim, use_slots, mixed):
    """The whole network as one hand-written autograd node (fused._Trunk: LayerNorm / ELU-backward / LSTM kernels,
    merged heads, partial-sum bias gradients) against the stock module composition evaluated in float64 on the CPU:
    head outputs, final LSTM state and the gradient of every parameter."""
    import copy
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.network import ModelA2CContinuousLogStd
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    model = ModelA2CContinuousLogStd(load_config()["train"]["params"]["network"], 2, (obs_dim,), True, True)
    with torch.no_grad():                      # make biases / LayerNorm parameters non-trivial
        for name, p in model.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.1)
    ref = copy.deepcopy(model).double()
    net = model.a2c_network.to(dev)
    B, T = 1024, 4
    n = B * T
    obs = torch.randn(n, obs_dim).clamp(-5, 5)
    h0, c0 = torch.randn(1, B, 256) * 0.5, torch.randn(1, B, 256) * 0.5
    dones = (torch.rand(n) < 0.2).to(torch.uint8)
    g = torch.randn(n, 3) / n
    # reference
    mu, _, value, (hT, cT) = ref.a2c_network(obs.double(), (h0.double(), c0.double()), T, dones)
    torch.autograd.backward([mu, value], [g[:, :2].double(), g[:, 2:].double()])
    # fused
    params = list(net.parameters())
    if use_slots:                              # the optimiser's flat gradient block: gradients are written in place
        for p in params:
            p.grad = torch.zeros_like(p)
    obs_d = obs.to(dev)
    if mixed:
        net.op_weight_lookup = lambda p: p.detach().to(fused.lp_dtype())
    out_tol, grad_tol = (2e-2, 4e-2) if mixed else (2e-5, 2e-4)
    assert net.trunk_supported(obs_d, T)
    heads, (h2, c2) = net.forward_heads(obs_d, (h0.to(dev), c0.to(dev)), T, dones.to(dev))
    heads.backward(g.to(dev))

    def close(a, b, tol, name):
        b = b.to(torch.float64)
        err = float((a.detach().cpu().double() - b).abs().max()) / (float(b.abs().max()) + 1e-12)
        assert err < tol, (name, err)

    close(heads[:, :2], mu, out_tol, "mu")
    close(heads[:, 2:], value, out_tol, "value")
    close(h2, hT, out_tol, "hT")
    close(c2, cT, out_tol, "cT")
    ref_params = dict(ref.a2c_network.named_parameters())
    for name, p in net.named_parameters():
        if name == "sigma":
            continue
        assert p.grad is not None, name
        close(p.grad, ref_params[name].grad, grad_tol, name)


@pytest.mark.gpu
@pytest.mark.parametrize("clip_value", [True, False])
def test_ln_heads_loss_kernel_matches_float64_autograd(clip_value):
    """LayerNorm + heads + PPO loss + their backward in ONE launch (vine_ln_heads_loss) against the float64 autograd
    of the stock composition: heads, d loss / d x, the gradients of gamma, beta, the head weights and biases, log sigma,
    the loss statistics, the KL slot and the refreshed dataset mu / sigma."""
    import ctypes
    from vine_robot_isaacgymenvs_amd.abi import PPO_LOSS_SCRATCH_FLOATS
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    n, H, A = 4096, 256, 2
    NH = A + 1
    lib = fused._lib()
    x = torch.randn(n, H, device=dev) * 1.5 + 0.3
    gamma, beta = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
    w, wb = torch.randn(NH, H, device=dev) * 0.05, torch.randn(NH, device=dev) * 0.1
    logstd = torch.tensor([-0.3, 0.2], device=dev)
    actions = torch.randn(n, A, device=dev)
    old_mu, old_sigma = torch.randn(n, A, device=dev) * 0.5, torch.rand(n, A, device=dev) * 0.5 + 0.5
    old_nlp = (0.5 * (((actions - old_mu) / old_sigma) ** 2).sum(-1) + 0.9189385 * A + old_sigma.log().sum(-1))
    adv, old_values, returns = torch.randn(n, device=dev), torch.randn(n, device=dev), torch.randn(n, device=dev)
    scal = (0.2, int(clip_value), 2.0, 0.01, 0.0001, 1.1)
    R = lib.vine_ln_heads_loss_rows()
    heads, dx = torch.empty(n, NH, device=dev), torch.empty(n, H, device=dev)
    part = torch.empty(n // R, (2 + NH) * H, device=dev)
    stats, gls = torch.empty(8, device=dev), torch.empty(A, device=dev)
    gmb, gvb = torch.zeros(A, device=dev), torch.zeros(1, device=dev)
    scratch = torch.empty(PPO_LOSS_SCRATCH_FLOATS, device=dev)
    kl_out, gls_acc = torch.zeros(1, device=dev), torch.zeros(A, device=dev)
    mu_st, sg_st = torch.empty(n, A, device=dev), torch.empty(n, A, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):          # twice: the workgroup ticket must be back at zero after a launch
        gmb.zero_(); gvb.zero_(); gls_acc.zero_()
        rc = lib.vine_ln_heads_loss(n, H, NH, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(), wb.data_ptr(),
                                    logstd.data_ptr(), actions.data_ptr(), old_nlp.data_ptr(), adv.data_ptr(),
                                    old_values.data_ptr(), returns.data_ptr(), old_mu.data_ptr(), old_sigma.data_ptr(), *scal,
                                    heads.data_ptr(), dx.data_ptr(), 0, part.data_ptr(), stats.data_ptr(), gls.data_ptr(),
                                    gmb.data_ptr(), gvb.data_ptr(), scratch.data_ptr(), kl_out.data_ptr(), gls_acc.data_ptr(),
                                    mu_st.data_ptr(), sg_st.data_ptr(), None, None, st)
        assert rc == 0
    torch.cuda.synchronize()
    # float64 reference
    xd = x.double().requires_grad_(True)
    gd, bd, wd, wbd, lsd = (t.double().requires_grad_(True) for t in (gamma, beta, w, wb, logstd))
    y = torch.nn.functional.layer_norm(xd, (H,), gd, bd, 1e-5)
    hd = y @ wd.t() + wbd
    loss, ref = fused.ppo_loss_reference(hd[:, :A], lsd, hd[:, A:], actions.double(), old_nlp.double(), adv.double(),
                                         old_values.double(), returns.double(), old_mu.double(), old_sigma.double(),
                                         scal[0], bool(scal[1]), scal[2], scal[3], scal[4], scal[5])
    loss.backward()
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
    assert rel(heads, hd.detach()) < 1e-5
    assert rel(dx, xd.grad) < 2e-4
    sums = part.double().sum(0)
    assert rel(sums[:H], gd.grad) < 2e-4 and rel(sums[H:2 * H], bd.grad) < 2e-4
    assert rel(sums[2 * H:].view(NH, H), wd.grad) < 2e-4
    assert rel(torch.cat([gmb, gvb]), wbd.grad) < 2e-4
    assert rel(gls, lsd.grad) < 2e-4 and rel(gls_acc, lsd.grad) < 2e-4
    for k_, name in enumerate(("a_loss", "c_loss", "b_loss", "entropy", "kl", "loss")):
        assert abs(float(stats[k_]) - float(ref[name])) < 2e-5 * max(1.0, abs(float(ref[name]))), name
    assert abs(float(kl_out) - float(ref["kl"])) < 2e-5
    assert rel(mu_st, hd.detach()[:, :A]) < 1e-5 and rel(sg_st, lsd.detach().exp().expand(n, A)) < 1e-6
    # the gradient handed to the LSTM backward as bfloat16 (dx_bf16): the fp32 result rounded, everything else unchanged
    dx16, heads2 = torch.empty(n, H, device=dev, dtype=fused.lp_dtype()), torch.empty_like(heads)
    gmb.zero_(); gvb.zero_(); gls_acc.zero_()
    assert lib.vine_ln_heads_loss(n, H, NH, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(), wb.data_ptr(),
                                  logstd.data_ptr(), actions.data_ptr(), old_nlp.data_ptr(), adv.data_ptr(),
                                  old_values.data_ptr(), returns.data_ptr(), old_mu.data_ptr(), old_sigma.data_ptr(), *scal,
                                  heads2.data_ptr(), dx16.data_ptr(), 1, part.data_ptr(), stats.data_ptr(), gls.data_ptr(),
                                  gmb.data_ptr(), gvb.data_ptr(), scratch.data_ptr(), kl_out.data_ptr(), gls_acc.data_ptr(),
                                  mu_st.data_ptr(), sg_st.data_ptr(), None, None, st) == 0
    torch.cuda.synchronize()
    # (the two instantiations may contract their fp32 arithmetic differently: one 16-bit ulp, incl. fp16 subnormals)
    assert torch.allclose(dx16.float(), dx.to(fused.lp_dtype()).float(), rtol=2e-3, atol=1.2e-7) and torch.equal(heads2, heads)
    # shapes outside the family are refused
    assert lib.vine_ln_heads_loss(n + 8, H, NH, *([x.data_ptr()] * 3), 1e-5, *([x.data_ptr()] * 10), *scal,
                                  x.data_ptr(), x.data_ptr(), 0, *([x.data_ptr()] * 10), None, None, st) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096, 32768])
def test_ln_heads_loss_kernel_16bit_input(n):
    """vine_ln_heads_loss reading the LSTM output in the 16-bit format (dx_bf16 bit 1: other column ownership per lane,
    all rows requested up front and kept in registers) against the same kernel fed the SAME values as fp32: every output
    agrees to summation-order accuracy, dx to one 16-bit ulp."""
    from vine_robot_isaacgymenvs_amd.abi import PPO_LOSS_SCRATCH_FLOATS
    dev = torch.device("cuda:0")
    torch.manual_seed(12)
    H, A = 256, 2
    NH = A + 1
    lib = fused._lib()
    bf = fused.lp_dtype()
    x16 = (torch.randn(n, H, device=dev) * 0.6 + 0.1).to(bf)
    x32 = x16.float()
    gamma, beta = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
    w, wb = torch.randn(NH, H, device=dev) * 0.05, torch.randn(NH, device=dev) * 0.1
    logstd = torch.tensor([-0.3, 0.2], device=dev)
    actions = torch.randn(n, A, device=dev)
    old_mu, old_sigma = 0.5 * actions + 0.1 * torch.randn(n, A, device=dev), torch.rand(n, A, device=dev) * 0.5 + 0.7      # (moderate probability ratios: no 16-bit overflow)
    old_nlp = (0.5 * (((actions - old_mu) / old_sigma) ** 2).sum(-1) + 0.9189385 * A + old_sigma.log().sum(-1))
    adv, old_values, returns = torch.randn(n, device=dev), torch.randn(n, device=dev), torch.randn(n, device=dev)
    scal = (0.2, 1, 2.0, 0.01, 0.0001, 1.1)
    R = lib.vine_ln_heads_loss_rows()
    st = torch.cuda.current_stream().cuda_stream
    scale, found = torch.full((1,), 4.0, device=dev), torch.zeros(1, device=dev)

    def run(x, flags):
        o = dict(heads=torch.empty(n, NH, device=dev), dx=torch.empty(n, H, device=dev, dtype=bf),
                 part=torch.empty(n // R, (2 + NH) * H, device=dev), stats=torch.empty(8, device=dev),
                 gls=torch.empty(A, device=dev), gmb=torch.zeros(A, device=dev), gvb=torch.zeros(1, device=dev),
                 kl=torch.zeros(1, device=dev), acc=torch.zeros(A, device=dev), mu=torch.empty(n, A, device=dev),
                 sg=torch.empty(n, A, device=dev))
        scratch = torch.empty(PPO_LOSS_SCRATCH_FLOATS, device=dev)
        for _ in range(2):      # twice: the ticket is back at zero, the accumulating outputs are re-zeroed
            o["gmb"].zero_(); o["gvb"].zero_(); o["acc"].zero_()
            assert lib.vine_ln_heads_loss(n, H, NH, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(),
                                          wb.data_ptr(), logstd.data_ptr(), actions.data_ptr(), old_nlp.data_ptr(),
                                          adv.data_ptr(), old_values.data_ptr(), returns.data_ptr(), old_mu.data_ptr(),
                                          old_sigma.data_ptr(), *scal, o["heads"].data_ptr(), o["dx"].data_ptr(), flags,
                                          o["part"].data_ptr(), o["stats"].data_ptr(), o["gls"].data_ptr(),
                                          o["gmb"].data_ptr(), o["gvb"].data_ptr(), scratch.data_ptr(), o["kl"].data_ptr(),
                                          o["acc"].data_ptr(), o["mu"].data_ptr(), o["sg"].data_ptr(), scale.data_ptr(),
                                          found.data_ptr(), st) == 0
        torch.cuda.synchronize()
        return o
    a, b = run(x16, 3), run(x32, 1)
    assert float(found) == 0.0
    # ... and reading the samples from slots 1 .. T of a [n / T, T + 1, H] tensor (the LSTM kernel's "h once" layout)
    for T in (4, 1, 8):
        xs = torch.full((n // T, T + 1, H), float("nan"), device=dev, dtype=bf)
        xs[:, 1:] = x16.view(n // T, T, H)
        c = run(xs, 3 | (T << 8))
        for k in a:
            assert torch.equal(a[k], c[k]), (T, k)
    rel = lambda u, v: float((u.double() - v.double()).abs().max() / (v.double().abs().max() + 1e-12))
    assert rel(a["heads"], b["heads"]) < 2e-6 and rel(a["mu"], b["mu"]) < 2e-6 and torch.equal(a["sg"], b["sg"])
    assert torch.allclose(a["dx"].float(), b["dx"].float(), rtol=4e-3, atol=2.4e-7)
    assert rel(a["part"].sum(0), b["part"].sum(0)) < 1e-5
    for k in ("stats", "gls", "gmb", "gvb", "kl", "acc"):
        assert rel(a[k], b[k]) < 1e-5, k
    # the serial last step deferred (flag bit 2) into the batched column-sum launch: same results, bit for bit
    import ctypes as C
    from vine_robot_isaacgymenvs_amd.abi import LossFinalize
    d = dict(heads=torch.empty(n, NH, device=dev), dx=torch.empty(n, H, device=dev, dtype=bf),
             part=torch.empty(n // R, (2 + NH) * H, device=dev), stats=torch.full((8,), float("nan"), device=dev),
             gls=torch.full((A,), float("nan"), device=dev), gmb=torch.zeros(A, device=dev), gvb=torch.zeros(1, device=dev),
             kl=torch.zeros(1, device=dev), acc=torch.zeros(A, device=dev), mu=torch.empty(n, A, device=dev),
             sg=torch.empty(n, A, device=dev))
    scratch = torch.empty(PPO_LOSS_SCRATCH_FLOATS, device=dev)
    assert lib.vine_ln_heads_loss(n, H, NH, x16.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(),
                                  wb.data_ptr(), logstd.data_ptr(), actions.data_ptr(), old_nlp.data_ptr(), adv.data_ptr(),
                                  old_values.data_ptr(), returns.data_ptr(), old_mu.data_ptr(), old_sigma.data_ptr(), *scal,
                                  d["heads"].data_ptr(), d["dx"].data_ptr(), 3 | 4, d["part"].data_ptr(), d["stats"].data_ptr(),
                                  d["gls"].data_ptr(), d["gmb"].data_ptr(), d["gvb"].data_ptr(), scratch.data_ptr(),
                                  d["kl"].data_ptr(), d["acc"].data_ptr(), d["mu"].data_ptr(), d["sg"].data_ptr(),
                                  scale.data_ptr(), found.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.isnan(d["stats"]).all() and float(d["kl"]) == 0.0          # nothing folded yet
    fin = LossFinalize(scratch.data_ptr(), n // R, A, n, logstd.data_ptr(), scal[2], scal[3], scal[4], d["stats"].data_ptr(),
                       d["gls"].data_ptr(), d["gmb"].data_ptr(), d["gvb"].data_ptr(), d["kl"].data_ptr(), d["acc"].data_ptr(),
                       scale.data_ptr())
    colsum = torch.empty((2 + NH) * H, device=dev)
    one = lambda v, t=C.c_int64: (t * 1)(v)
    assert lib.vine_column_sums_batched_fin(1, one(n // R), one((2 + NH) * H), (C.c_void_p * 1)(d["part"].data_ptr()),
                                            one((2 + NH) * H), (C.c_void_p * 1)(colsum.data_ptr()), one(0),
                                            (C.c_void_p * 1)(None), one(0, C.c_int32), None, C.byref(fin), st) == 0
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], d[k]), k
    assert rel(colsum, a["part"].sum(0)) < 1e-5
    # 16-bit input with an fp32 gradient is not a supported combination
    assert lib.vine_ln_heads_loss(n, H, NH, x16.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(),
                                  wb.data_ptr(), logstd.data_ptr(), actions.data_ptr(), old_nlp.data_ptr(), adv.data_ptr(),
                                  old_values.data_ptr(), returns.data_ptr(), old_mu.data_ptr(), old_sigma.data_ptr(), *scal,
                                  a["heads"].data_ptr(), a["dx"].data_ptr(), 2, a["part"].data_ptr(), a["stats"].data_ptr(),
                                  a["gls"].data_ptr(), a["gmb"].data_ptr(), a["gvb"].data_ptr(), a["part"].data_ptr(),
                                  a["kl"].data_ptr(), a["acc"].data_ptr(), a["mu"].data_ptr(), a["sg"].data_ptr(), None, None,
                                  st) == -2


@pytest.mark.gpu
def test_ppo_loss_kernel_against_reference_text_golden():
    """Golden F8 through the HIP loss kernel: the means of the actor, clipped-critic and bound (soft bound 1.0) terms the
    reference's in-tree text computes per sample (isaacgymenvs/learning/common_agent.py:482-516, 427-435)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f8_ppo_loss_terms.npz"))
    dev = torch.device("cuda:0")
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    n, A = g["mu"].shape
    # the kernel derives neglogp from (mu, sigma, action): sigma = 1, action = mu + sqrt(2 (nlp - const)) on one axis;
    # only old_neglogp - neglogp enters the loss, so both are shifted to make every target reachable
    const = 0.5 * np.log(2 * np.pi) * A
    shift = float(const - g["neglogp"].min() + 0.01)
    mu = t("mu")
    actions = mu.clone()
    actions[:, 0] += torch.sqrt(2.0 * (t("neglogp") + shift - const))
    logstd = torch.zeros(A, device=dev)
    _, _, _, stats = fused.ppo_loss_fused(mu, logstd, t("values"), actions, t("old_neglogp") + shift, t("advantage"), t("old_values"),
                                          t("returns"), mu, torch.ones_like(mu), float(g["e_clip"]), True, 2.0, 0.0, 1e-4,
                                          soft_bound=1.0)
    stats = stats.cpu().numpy()
    assert abs(stats[0] - g["a_loss"].mean()) < 2e-5 * (1 + abs(g["a_loss"].mean()))
    assert abs(stats[1] - g["c_loss"].mean()) < 2e-5 * (1 + abs(g["c_loss"].mean()))
    assert abs(stats[2] - g["b_loss_soft_bound_1"].mean()) < 2e-5


@pytest.mark.gpu
def test_default_loss_kernel_against_reference_text_golden_per_sample():
    """Golden F8 through the kernel the default update actually runs (`ln_heads_loss_kernel`, vine_ln_heads_loss), PER
    SAMPLE: the kernel reports means over its rows, so every one of the 512 fixture samples is launched as its own batch
    of identical rows (one workgroup's worth) -- its three statistics are then that sample's actor, clipped-critic and
    bound terms as the reference's text computes them (common_agent.py:482-516, 427-435).  The heads are injected through
    the head biases (zero head weights), the sample's neglogp through the action (sigma = 1)."""
    import os
    from vine_robot_isaacgymenvs_amd import native
    from vine_robot_isaacgymenvs_amd.abi import PPO_LOSS_SCRATCH_FLOATS, PPO_PARTIAL_BLOCKS
    lib = native.load()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f8_ppo_loss_terms.npz"))
    dev = torch.device("cuda:0")
    S, A = g["mu"].shape
    H, NH = 256, A + 1
    n = lib.vine_ln_heads_loss_rows()
    const = 0.5 * np.log(2 * np.pi) * A
    shift = float(const - g["neglogp"].min() + 0.01)          # only old_neglogp - neglogp enters the loss
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(n, H, device=dev, generator=gen)
    gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    w = torch.zeros(NH, H, device=dev)
    logstd = torch.zeros(A, device=dev)
    heads, dx = torch.empty(n, NH, device=dev), torch.empty(n, H, device=dev)
    part = torch.empty(PPO_PARTIAL_BLOCKS, 2 * H + NH * H, device=dev)
    stats, gls = torch.zeros(8, device=dev), torch.zeros(A, device=dev)
    gmb, gvb = torch.zeros(A, device=dev), torch.zeros(1, device=dev)
    scratch = torch.empty(PPO_LOSS_SCRATCH_FLOATS, device=dev)
    kl_out, gls_acc = torch.zeros(1, device=dev), torch.zeros(A, device=dev)
    mu_st, sg_st = torch.empty(n, A, device=dev), torch.empty(n, A, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    full = lambda v, cols=None: (torch.full((n,), float(v), device=dev) if cols is None
                                 else torch.tensor(np.asarray(v, np.float32), device=dev).expand(n, cols).contiguous())
    got = np.zeros((S, 3))
    for i in range(S):
        mu_i = g["mu"][i]
        wb = torch.tensor(np.concatenate([mu_i, g["values"][i]]).astype(np.float32), device=dev)
        act = mu_i.astype(np.float64).copy()
        act[0] += np.sqrt(2.0 * (float(g["neglogp"][i]) + shift - const))
        actions = full(act, A)
        old_mu, old_sigma = full(mu_i, A), torch.ones(n, A, device=dev)
        # (named tensors: a temporary's storage would be recycled by the next allocation while the kernel still reads it)
        onlp, adv = full(g["old_neglogp"][i] + shift), full(g["advantage"][i])
        oval, ret = full(g["old_values"][i, 0]), full(g["returns"][i, 0])
        rc = lib.vine_ln_heads_loss(n, H, NH, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(), wb.data_ptr(),
                                    logstd.data_ptr(), actions.data_ptr(), onlp.data_ptr(),
                                    adv.data_ptr(), oval.data_ptr(),
                                    ret.data_ptr(), old_mu.data_ptr(), old_sigma.data_ptr(),
                                    float(g["e_clip"]), 1, 2.0, 0.0, 1e-4, 1.0,
                                    heads.data_ptr(), dx.data_ptr(), 0, part.data_ptr(), stats.data_ptr(), gls.data_ptr(),
                                    gmb.data_ptr(), gvb.data_ptr(), scratch.data_ptr(), kl_out.data_ptr(), gls_acc.data_ptr(),
                                    mu_st.data_ptr(), sg_st.data_ptr(), None, None, st)
        assert rc == 0
        got[i] = stats[:3].cpu().numpy()
    tol = lambda ref: 3e-5 * (1.0 + np.abs(ref))
    assert (np.abs(got[:, 0] - g["a_loss"]) < tol(g["a_loss"])).all()
    assert (np.abs(got[:, 1] - g["c_loss"][:, 0]) < tol(g["c_loss"][:, 0])).all()
    assert (np.abs(got[:, 2] - g["b_loss_soft_bound_1"]) < tol(g["b_loss_soft_bound_1"])).all()
    assert (g["c_loss"] != g["c_loss_noclip"]).any()          # the clipped branch of the critic term is exercised


@pytest.mark.gpu
def test_fused_update_equals_stock_update():
    """One optimiser step of the agent through the fused path and through the stock composition, from the same
    weights and minibatch: same loss statistics, same updated parameters (to fp32 reduction-order noise)."""
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    outs = []
    for use_fused in (True, False):
        cfg = load_config(overrides=["num_envs=512", "minibatch_size=2048"])
        cfg["task"]["seed"] = 42
        env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
        params = cfg["train"]["params"]
        params["config"].update(write_files=False, print_stats=False, use_graphs=False, use_fused_ops=use_fused,
                                mini_epochs=1, mixed_precision=False)
        torch.manual_seed(0)
        agent = A2CAgent("t", params, vec_env=env)
        agent.init_tensors()
        agent.fused_rollout = False          # same (stock, torch.randn) rollout for both; only the update differs
        agent.game_rewards.mean, agent.game_rewards.current_size = torch.zeros(1, device="cuda:0"), torch.zeros((), device="cuda:0")
        agent.game_lengths.mean, agent.game_lengths.current_size = torch.zeros(1, device="cuda:0"), torch.zeros((), device="cuda:0")
        agent.obs = agent.env_reset()["obs"]
        torch.manual_seed(5)
        play, upd, stats = agent.train_epoch()
        torch.cuda.synchronize()
        outs.append((torch.cat([p.detach().flatten() for p in agent.model.parameters()]),
                     {k: float(v) for k, v in stats.items()}))
        env.close()
    (p1, s1), (p2, s2) = outs
    for k in s1:
        assert abs(s1[k] - s2[k]) < 2e-3 * (1 + abs(s2[k])), (k, s1[k], s2[k])
    assert float((p1 - p2).abs().max()) < 2e-3      # 4 Adam steps of 3e-4.. lr: identical sign pattern of the updates


@pytest.mark.gpu
@pytest.mark.parametrize("mixed,scope", [(False, "epoch"), (True, "epoch"), (True, "step"), (True, "all"), (False, "all")])
def test_graphed_update_equals_eager_update(mixed, scope, monkeypatch):
    """From the second iteration on the update is replayed from hipGraphs: on one rank the whole update as ONE graph (scope
    "all", the default), a whole mini-epoch per graph
    (scope "epoch": one graph with and one without the running-statistics update), or one graph per optimiser step
    (scope "step", the form used with several ranks: [Adam + schedule of the previous step | forward / backward], the
    all-reduce issued between two such graphs, one trailing Adam graph).  Five iterations with and without graphs must leave the same parameters, learning rate and loss
    statistics (fp32 and mixed-precision update)."""
    monkeypatch.setenv("VINE_UPD_GRAPH", scope)
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    outs = []
    for use_graphs in (True, False):
        cfg = load_config(overrides=["num_envs=512", "minibatch_size=2048"])
        cfg["task"]["seed"] = 42
        env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
        params = cfg["train"]["params"]
        params["config"].update(write_files=False, print_stats=False, use_graphs=use_graphs, mixed_precision=mixed)
        torch.manual_seed(0)
        agent = A2CAgent("t", params, vec_env=env)
        assert agent.fused_mixed == mixed
        agent.init_tensors()
        agent.obs = agent.env_reset()["obs"]
        stats = None
        for _ in range(5):                   # 1 eager + capture/first replay + 3 further replays
            _, _, stats = agent.train_epoch()
        torch.cuda.synchronize()
        if use_graphs and scope in ("epoch", "all") and mixed:
            # the first mini-epoch's graph holds the normaliser updates of all its steps at its head (vine_rms_update_multi)
            nb = agent.num_minibatches
            assert agent._upd_graphs[(scope, True)]["keep"][nb] is not None
            assert agent.graph_status["update"] == "graph (1 per %s)" % ("iteration" if scope == "all" else "mini-epoch")
        if use_graphs:
            # per-step form: (step, with / without the RMS update) graphs, each led by the previous step's Adam, + the
            # un-led first step of an update + the trailing Adam graph
            assert len(agent._upd_graphs) == (1 if scope == "all" else 2 if scope == "epoch" else 2 * agent.num_minibatches + 1)
            assert not getattr(agent, "_update_graphs_failed", False)
        outs.append((torch.cat([p.detach().flatten() for p in agent.model.parameters()]).clone(), float(agent.lr),
                     {k: float(v) for k, v in stats.items()}, agent.model.running_mean_std.running_mean.clone()))
        env.close()
    (p1, lr1, s1, m1), (p2, lr2, s2, m2) = outs
    # no float atomics anywhere on the training path and the same kernels either way: replayed and eagerly launched
    # training are bit-identical
    assert lr1 == lr2
    assert torch.equal(m1, m2)
    assert s1 == s2, (s1, s2)
    assert torch.equal(p1, p2)


@pytest.mark.gpu
def test_default_update_stores_the_hidden_states_once(monkeypatch):
    """The default (mixed-precision) update takes the "h once" route -- one 16-bit copy of the LSTM's hidden states read
    by the loss kernel and by the weight-gradient kernel -- and lands where the two-copy route (fp32 states for the
    LayerNorm) lands, to the rounding of the LayerNorm's input."""
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    calls = {"seq": 0, "plain": 0}
    real = fused.weight_grad_cat

    def counting(dy, x1, x2, out1, out2, batch=None, seq=None):
        ok = real(dy, x1, x2, out1, out2, batch=batch, seq=seq)
        if ok and x1 is not None:
            calls["seq" if seq is not None else "plain"] += 1
        return ok
    monkeypatch.setattr(fused, "weight_grad_cat", counting)
    outs = []
    for h_once in (True, False):
        monkeypatch.setattr(fused, "H_ONCE", h_once)
        cfg = load_config(overrides=["num_envs=512", "minibatch_size=2048"])
        cfg["task"]["seed"] = 42
        env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
        params = cfg["train"]["params"]
        params["config"].update(write_files=False, print_stats=False, use_graphs=False)
        torch.manual_seed(0)
        agent = A2CAgent("t", params, vec_env=env)
        assert agent.fused_mixed
        agent.init_tensors()
        agent.obs = agent.env_reset()["obs"]
        before = dict(calls)
        for _ in range(2):
            _, _, stats = agent.train_epoch()
        torch.cuda.synchronize()
        steps = 2 * agent.mini_epochs_num * agent.num_minibatches
        assert calls["seq" if h_once else "plain"] - before["seq" if h_once else "plain"] == steps
        assert calls["plain" if h_once else "seq"] == before["plain" if h_once else "seq"]
        outs.append((torch.cat([p.detach().flatten() for p in agent.model.parameters()]).clone(),
                     {k: float(v) for k, v in stats.items()}))
        env.close()
    (p1, s1), (p2, s2) = outs
    assert torch.isfinite(p1).all() and float((p1 - p2).abs().max()) < 3e-3, float((p1 - p2).abs().max())
    for k in s1:
        assert abs(s1[k] - s2[k]) < 2e-2 * (1 + abs(s2[k])), (k, s1[k], s2[k])


@pytest.mark.gpu
def test_checkpoint_restore_continues_bit_identically(tmp_path):
    """Save after two iterations, restore into a fresh agent (weights, Adam moments, normaliser statistics, learning
    rate, bf16 operand shadows) and train one more update on the SAME experience: the restored agent must land on the
    same parameters bit for bit (default configuration: mixed precision, hipGraph rollout and update)."""
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map

    def build():
        cfg = load_config(overrides=["num_envs=512", "minibatch_size=2048"])
        cfg["task"]["seed"] = 42
        env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
        params = cfg["train"]["params"]
        params["config"].update(write_files=False, print_stats=False)
        torch.manual_seed(0)
        agent = A2CAgent("t", params, vec_env=env)
        agent.init_tensors()
        agent.obs = agent.env_reset()["obs"]
        return agent, env

    a, env_a = build()
    assert a.fused_mixed
    for _ in range(2):
        a.train_epoch()
    path = a.save(str(tmp_path / "ck"))
    # one more iteration on agent a; capture the experience it used
    a.set_eval()
    with torch.no_grad():
        batch = a.play_steps_rnn()
    batch_copy = {k: ([t.clone() for t in v] if isinstance(v, list) else (v.clone() if torch.is_tensor(v) else v))
                  for k, v in batch.items()}

    def update_on(agent, b):
        agent.set_train()
        if agent.fused_mixed:
            agent.optimizer.refresh_shadow()
        agent.curr_frames = b.pop("played_frames")
        agent.prepare_dataset(b)
        for _ep in range(agent.mini_epochs_num):
            for i in range(agent.num_minibatches):
                mb = agent.get_minibatch(i)
                _, _, _, kl, _, cmu, csig = agent.calc_gradients(mb)
                s, e = mb["range"]
                agent.dataset["mu"][s:e] = cmu
                agent.dataset["sigma"][s:e] = csig
                agent.update_lr_from_kl(kl)
            agent.model.running_mean_std.eval()
        torch.cuda.synchronize()
        return torch.cat([p.detach().flatten() for p in agent.model.parameters()]).clone(), float(agent.lr)

    pa, lra = update_on(a, dict(batch_copy))
    b, env_b = build()
    b.restore(path)
    batch_b = {k: ([t.clone() for t in v] if isinstance(v, list) else (v.clone() if torch.is_tensor(v) else v))
               for k, v in batch_copy.items()}
    pb, lrb = update_on(b, batch_b)
    assert lra == lrb
    assert torch.equal(pa, pb)
    env_a.close(); env_b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("amp", ["fp16", "bf16"])
def test_torch_autocast_path_still_trains(amp):
    """The reference's literal mechanism -- torch autocast (+ GradScaler for fp16) over the stock composition -- stays
    available: ``use_fused_ops: False``, or a ``mixed_precision_dtype`` other than the library's 16-bit format with the
    fused ops on (the hand-written fp32 kernels must step aside under autocast: they once received half-precision
    tensors there)."""
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    cfg = load_config(overrides=["num_envs=512", "minibatch_size=4096"])
    cfg["task"]["seed"] = 42
    env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                  graphics_device_id=0, headless=True)
    params = cfg["train"]["params"]
    lib_fmt = {torch.float16: "fp16", torch.bfloat16: "bf16"}[fused.lp_dtype()]
    params["config"].update(write_files=False, print_stats=False, mixed_precision=True, mixed_precision_dtype=amp,
                            use_fused_ops=(amp != lib_fmt))
    torch.manual_seed(0)
    agent = A2CAgent("t", params, vec_env=env)
    assert agent.mixed_precision and not agent.fused_mixed
    agent.init_tensors()
    agent.obs = agent.env_reset()["obs"]
    for _ in range(2):
        _, _, stats = agent.train_epoch()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in agent.model.parameters())
    assert all(np.isfinite(float(v)) for v in stats.values())
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_env,mb,obs_type,mixed", [(100, 400, "POS_AND_FD_VEL_AND_OBJ_INFO", True),
                                                     (72, 288, "TIP_AND_CART_AND_OBJ_INFO", True),
                                                     (100, 1600, "POS_AND_FD_VEL_AND_OBJ_INFO", False),
                                                     (64, 1024, "POS_AND_FD_VEL_AND_OBJ_INFO", True)])
def test_ragged_sizes_through_the_whole_path(n_env, mb, obs_type, mixed):
    """Env counts that are not multiples of the kernels' tile sizes (64-row MFMA tiles, 256-thread workgroups, 4-wide
    vectors), both observation widths: three full PPO iterations incl. graph capture, every shape guard falling back
    where it must."""
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    cfg = load_config(overrides=["num_envs=%d" % n_env, "minibatch_size=%d" % mb, "OBSERVATION_TYPE=" + obs_type])
    env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                  graphics_device_id=0, headless=True)
    params = cfg["train"]["params"]
    params["config"].update(write_files=False, print_stats=False, mixed_precision=mixed)
    agent = A2CAgent("t", params, vec_env=env)
    agent.init_tensors()
    agent.obs = agent.env_reset()["obs"]
    for _ in range(3):
        _, _, stats = agent.train_epoch()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in agent.model.parameters())
    assert all(np.isfinite(float(v)) for v in stats.values())
    assert agent._fast is not None and list(agent._upd_graphs) == [("all", True)]      # the whole update: one graph
    env.close()


def _adam_pair(device):
    from vine_robot_isaacgymenvs_amd.learning.flat_adam import FlatAdam
    torch.manual_seed(3)
    shapes = [(64, 28), (64,), (1024, 92), (3,), (2,)]
    a = [torch.randn(s, device=device).requires_grad_() for s in shapes]
    b = [t.detach().clone().requires_grad_() for t in a]
    lr = torch.tensor(3e-4, device=device)
    flat = FlatAdam(a, lr, eps=1e-8)
    ref = torch.optim.Adam(b, lr=3e-4, eps=1e-8)
    for it in range(5):
        grads = [torch.randn(s, device=device) for s in shapes]
        for p, q, g in zip(a, b, grads):
            p.grad.copy_(2.0 * g)          # as if summed over 2 ranks
            q.grad = g.clone()
        flat.step(grad_scale=0.5)
        ref.step()
        if it == 2:                        # the schedule changes the device scalar between steps
            lr.mul_(1.5)
            ref.param_groups[0]["lr"] *= 1.5
    return a, b, flat, ref


def test_flat_adam_cpu_matches_torch_adam():
    a, b, flat, ref = _adam_pair(torch.device("cpu"))
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)
    assert float(flat.flat_grads.abs().max()) == 0.0
    sd = flat.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 5


@pytest.mark.gpu
def test_flat_adam_kernel_matches_torch_adam():
    a, b, flat, ref = _adam_pair(torch.device("cuda:0"))
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-7)
    assert float(flat.flat_grads.abs().max()) == 0.0 and float(flat.step_t) == 5.0
    for i, q in enumerate(b):
        assert torch.allclose(flat.state_dict()["state"][i]["exp_avg_sq"], ref.state[q]["exp_avg_sq"], rtol=1e-4, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [700, 5000, 9 * 1024 + 3, 408261])
def test_adam_election_advances_step_once_per_launch(n):
    """The Adam kernel's "last workgroup" election (two levels: 8 group tickets, then the common one) at grids of 1, 5, 10 and
    399 workgroups: every launch advances the device step counter by exactly one, applies the KL schedule once and leaves
    the tickets reset -- 20 launches in a row -- and the parameters follow torch.optim.Adam."""
    from vine_robot_isaacgymenvs_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(n)
    p = torch.randn(n, device=dev)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, eps=1e-8)
    g, m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    lr, step = torch.full((1,), 1e-3, device=dev), torch.zeros(1, device=dev)
    kl = torch.full((1,), 1.0, device=dev)                      # far above the threshold: lr / 1.5 per launch, once
    st = torch.cuda.current_stream().cuda_stream
    lr_expect = 1e-3
    for it in range(20):
        grad = torch.randn(n, device=dev)
        g.copy_(grad)
        ref.grad = grad.clone()
        for group in opt.param_groups:
            group["lr"] = lr_expect
        opt.step()
        assert lib.vine_adam_step_sched(n, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), lr.data_ptr(), step.data_ptr(),
                                        0.9, 0.999, 1e-8, 0.0, 1.0, None, kl.data_ptr(), 1.0, 0.008, 1e-6, 1e-2, st) == 0
        torch.cuda.synchronize()
        lr_expect = max(lr_expect / 1.5, 1e-6)
        assert float(step) == it + 1
        assert abs(float(lr) - lr_expect) <= 1e-6 * lr_expect + 1e-12
    assert torch.allclose(p, ref.detach(), rtol=2e-5, atol=2e-7)
    assert float(g.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("mixed", [False, True, "lp16"])
def test_fused_rollout_matches_stock_and_graph_replay(mixed):
    """Fused rollout (hand-written inference trunk, policy head, post-step kernels) vs the stock PyTorch model on the
    stored inputs: deterministic quantities equal (to bf16 operand tolerance in the mixed-precision mode); eager and
    hipGraph-replayed fused rollouts are bit-identical (device-side Philox counter)."""
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.learning.network import ModelA2CContinuousLogStd
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map

    def build(use_graphs):
        cfg = load_config(overrides=["num_envs=512", "minibatch_size=2048", "seed=11"])
        cfg["task"]["seed"] = 42
        env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
        params = cfg["train"]["params"]
        params["config"].update(write_files=False, print_stats=False, use_graphs=use_graphs, mixed_precision=bool(mixed),
                                rollout_precision="lp16" if mixed == "lp16" else "fp32")
        torch.manual_seed(0)
        agent = A2CAgent("t", params, vec_env=env)
        agent.init_tensors()
        agent.obs = agent.env_reset()["obs"]
        # rollout inference is fp32 (the reference's) whatever the update's precision, unless rollout_precision says lp16
        assert agent._fast is not None and agent._fast["op"] == (fused.lp_dtype() if mixed == "lp16" else torch.float32)
        return agent, env

    tol = 3e-2 if mixed == "lp16" else 2e-4

    outs = []
    for use_graphs in (False, True):
        agent, env = build(use_graphs)
        assert agent._can_fuse_rollout()
        agent.set_eval()
        with torch.no_grad():
            for _ in range(3):
                batch = agent.play_steps_rnn()
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in agent.buf.items()} | {"meter": agent.meter.clone(),
                                                                    "last": agent.last_values.clone()})
        if not use_graphs:
            # deterministic parts of step n against the stock model evaluated on the stored inputs
            buf = agent.buf
            nlp = ModelA2CContinuousLogStd.neglogp(buf["actions"], buf["mus"], buf["sigmas"], torch.log(buf["sigmas"]))
            assert torch.allclose(nlp, buf["neglogpacs"], atol=2e-4)
            states = [agent.mb_rnn_states[0][:, :, 2].contiguous(), agent.mb_rnn_states[1][:, :, 2].contiguous()]   # LSTM state stored before step 8
            res = agent.model({"is_train": False, "obs": buf["obses"][8], "rnn_states": states})
            assert torch.allclose(res["mus"], buf["mus"][8], atol=tol)
            assert torch.allclose(res["values"], buf["values"][8], atol=tol * 5)       # un-normalised value scale
            eps = (buf["actions"] - buf["mus"]) / buf["sigmas"]
            assert abs(float(eps.mean())) < 0.02 and abs(float(eps.std()) - 1.0) < 0.02   # N(0,1) sampling
            assert float(agent.meter[1]) > 0 and float(agent.current_lengths.max()) <= 48
            assert int(agent.roll_counter) == 48
        env.close()
    for k in outs[0]:
        if k == "meter":      # float atomics: the sum order over finished episodes is not fixed
            assert torch.allclose(outs[0][k], outs[1][k], rtol=1e-5)
        else:
            assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.gpu
def test_rollout_plumbing_switches_are_bit_identical(monkeypatch):
    """the default rollout plumbing of round 4 -- the deferred meter / counter fold riding in the next MLP launch
    (VINE_ROLLOUT_FIN_RIDE: vine_rollout_post_defer + _pending_fin), the single-launch start copies (VINE_ROLLOUT_COPYBATCH:
    LSTM-state snapshots, slot-0 observations, done flags moved as float32 words) and the head reading the value
    normaliser's float64 statistics itself (VINE_POLICY_HEAD_RMS: vine_policy_head_rms) -- against the same rollout with
    each of them off: meter, roll_counter, every rollout buffer, the stored LSTM states, the episode accumulators and
    last_values must not change.  Also: a pending fold is never dropped when no fp32 MLP launch follows (explicit
    vine_rollout_finalize instead of the former assert)."""
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map

    def run(fin, copybatch, head_rms, step_fused=0, rollouts=3):
        monkeypatch.setenv("VINE_ROLLOUT_STEP_FUSED", str(step_fused))      # (round 5's one-launch step: compared below)
        monkeypatch.setenv("VINE_ROLLOUT_FIN_RIDE", str(fin))
        monkeypatch.setenv("VINE_ROLLOUT_COPYBATCH", str(copybatch))
        monkeypatch.setenv("VINE_POLICY_HEAD_RMS", str(head_rms))
        cfg = load_config(overrides=["num_envs=512", "minibatch_size=2048", "seed=11", "task.env.maxEpisodeLength=20"])
        cfg["task"]["seed"] = 42
        env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
        params = cfg["train"]["params"]
        params["config"].update(write_files=False, print_stats=False, use_graphs=False, mixed_precision=True)
        torch.manual_seed(0)
        agent = A2CAgent("t", params, vec_env=env)
        agent.init_tensors()
        agent.obs = agent.env_reset()["obs"]
        vms = agent.model.value_mean_std
        vms.running_mean.fill_(0.37); vms.running_var.fill_(2.3)          # a head that really un-normalises
        agent.set_eval()
        with torch.no_grad():
            for _ in range(rollouts):
                agent.play_steps_rnn()
        torch.cuda.synchronize()
        out = {k: v.clone() for k, v in agent.buf.items()}
        out.update(meter=agent.meter.clone(), counter=agent.roll_counter.clone(), last=agent.last_values.clone(),
                   cur_r=agent.current_rewards.clone(), cur_l=agent.current_lengths.clone(), dones=agent.dones.clone(),
                   h=agent.mb_rnn_states[0].clone(), c=agent.mb_rnn_states[1].clone(), h_live=agent.rnn_states[0].clone())
        pend = agent._pending_fin
        launches = agent.rollout_step_launches
        env.close()
        return out, (pend, launches)

    base, (pend, launches) = run(1, 1, 1)
    assert pend is None and launches == 5                             # the last-values forward carried the last fold
    assert int(base["counter"]) == 48 and float(base["meter"][1]) > 0 and float(base["meter"][3]) > 0     # episodes finished (20-step limit)
    for combo in ((0, 1, 1), (1, 0, 1), (0, 0, 1)):
        other, _ = run(*combo)
        for k in base:
            assert torch.equal(base[k], other[k]), (combo, k)
    # the two-float head (mean.float(), sqrt(var.float() + eps) formed by torch) against the kernel reading the float64 statistics
    other, _ = run(1, 1, 0)
    for k in base:
        if base[k].dtype.is_floating_point:
            torch.testing.assert_close(other[k], base[k], rtol=2e-6, atol=2e-6, msg=k)
        else:
            assert torch.equal(base[k], other[k]), k
    # round 5: head + env step + bookkeeping as ONE launch per step (vine_step_rollout; the default): same keys and formulas,
    # LayerNorm sums in another order -> the first step agrees to round-off, flags and counters of the whole run exactly
    # (the trajectories drift apart at the step kernel's own sensitivity to 1e-7 action differences: compared step 0 only)
    base1, _ = run(1, 1, 1, rollouts=1)
    one, (pend1, launches1) = run(1, 1, 1, step_fused=1, rollouts=1)
    assert pend1 is None and launches1 == 3
    assert torch.equal(one["counter"], base1["counter"]) and int(one["counter"]) == 16
    for k in ("mus", "values", "actions", "neglogpacs", "rewards"):
        torch.testing.assert_close(one[k][0], base1[k][0], rtol=1e-4, atol=1e-4, msg=k)
    assert torch.equal(one["dones"][:2], base1["dones"][:2]) and torch.equal(one["obses"][0], base1["obses"][0])
    torch.testing.assert_close(one["obses"][1], base1["obses"][1], rtol=1e-3, atol=1e-3)
    assert torch.equal(one["cur_l"], base1["cur_l"])              # episode lengths: the same envs finished at the same steps


@pytest.mark.gpu
def test_pending_rollout_fold_runs_as_its_own_launch_when_nothing_can_carry_it():
    """vine_rollout_finalize == the fold vine_rollout_post runs behind its per-env pass (same kernel body)."""
    from vine_robot_isaacgymenvs_amd.abi import ROLLOUT_POST_SCRATCH_FLOATS
    lib = fused._lib()
    dev = torch.device("cuda:0")
    N, H = 1024, 256
    g = torch.Generator(device=dev).manual_seed(3)
    rew = torch.randn(N, device=dev, generator=g)
    reset = (torch.rand(N, device=dev, generator=g) < 0.3).to(torch.int64)
    tmo = (torch.rand(N, device=dev, generator=g) < 0.1).to(torch.uint8)
    values = torch.randn(N, 1, device=dev, generator=g)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for split in (False, True):
        shaped, dones = torch.empty(N, 1, device=dev), torch.empty(N, device=dev, dtype=torch.uint8)
        cur_r, cur_l = torch.ones(N, 1, device=dev) * 2.0, torch.ones(N, device=dev) * 7.0
        h, c = torch.randn(1, N, H, device=dev, generator=torch.Generator(device=dev).manual_seed(9)), torch.ones(1, N, H, device=dev)
        meter = torch.tensor([1.5, 10.0, 30.0, 10.0, 0, 0, 0, 0], device=dev)
        counter = torch.tensor([5], device=dev, dtype=torch.int64)
        scratch = torch.zeros(ROLLOUT_POST_SCRATCH_FLOATS, device=dev)
        common = (N, H, rew.data_ptr(), reset.data_ptr(), tmo.data_ptr(), values.data_ptr(), 0.0, 0.01, 0.99, shaped.data_ptr(),
                  dones.data_ptr(), cur_r.data_ptr(), cur_l.data_ptr(), h.data_ptr(), c.data_ptr())
        if split:
            assert lib.vine_rollout_post_defer(*common, None, 0, 0, scratch.data_ptr(), st) == 0
            assert lib.vine_rollout_finalize(meter.data_ptr(), 100.0, counter.data_ptr(), scratch.data_ptr(),
                                             lib.vine_rollout_post_blocks(N), st) == 0
        else:
            assert lib.vine_rollout_post(*common, meter.data_ptr(), 100.0, counter.data_ptr(), None, 0, 0, scratch.data_ptr(), st) == 0
        torch.cuda.synchronize()
        res.append((shaped, dones, cur_r, cur_l, h, c, meter, counter))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert int(res[0][7]) == 6
    assert lib.vine_rollout_finalize(None, 100.0, None, None, 1, st) == -1      # VINE_ERR_INVALID_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("overrides", [[], ["task.env.CREATE_PIPE=True"], ["OBSERVATION_TYPE=TIP_AND_CART_AND_OBJ_INFO", "vine_randomize=False"]])
def test_rollout_step_in_one_launch_matches_the_three_launches(overrides):
    """Round 5: vine_step_rollout (policy head + env step + rollout bookkeeping inside the four-lane step kernel) against
    vine_policy_head_rms -> vine_step -> vine_rollout_post on a twin env: same Philox keys and formulas, so the sampled
    actions agree to fp32 round-off (the LayerNorm sums run over 4 lanes x 64 units instead of 16 x 16), everything the step
    derives from them to the step kernel's own sensitivity, flags / counters exactly; finished envs get their LSTM-state rows
    cleared and their episode totals into the per-workgroup rows."""
    import ctypes as C
    from vine_robot_isaacgymenvs_amd import abi, load_config
    from vine_robot_isaacgymenvs_amd.abi import ROLLOUT_POST_SCRATCH_FLOATS
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    lib = fused._lib()
    dev = torch.device("cuda:0")
    N, H, A = 1024, 256, 2
    st = torch.cuda.current_stream().cuda_stream

    def make():
        cfg = load_config(overrides=["num_envs=%d" % N, "task.env.CREATE_PIPE=False", "task.env.maxEpisodeLength=6"] + overrides)
        cfg["task"]["seed"] = 42
        return isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
    ea, eb = make(), make()
    assert eb.rollout_step_blocks() == N * 4 // 256
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    gamma, beta = 1.0 + 0.1 * rnd(H), 0.1 * rnd(H)
    w_mu, b_mu, w_v, b_v = 0.1 * rnd(A, H), 0.1 * rnd(A), 0.1 * rnd(1, H), 0.1 * rnd(1)
    logstd = torch.tensor([-0.3, 0.2], device=dev)
    vmean = torch.tensor([0.37], device=dev, dtype=torch.float64)
    vvar = torch.tensor([2.3], device=dev, dtype=torch.float64)
    counter = torch.tensor([3], device=dev, dtype=torch.int64)
    hw, hc = torch.empty(3 * H, device=dev), torch.empty(3, device=dev)
    assert lib.vine_rollout_head_prep(gamma.data_ptr(), beta.data_ptr(), w_mu.data_ptr(), b_mu.data_ptr(), w_v.data_ptr(),
                                      b_v.data_ptr(), hw.data_ptr(), hc.data_ptr(), st) == 0
    torch.cuda.synchronize()
    torch.testing.assert_close(hw.view(3, H), gamma * torch.cat([w_mu, w_v]), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(hc, torch.cat([(beta * w_mu).sum(1) + b_mu, (beta * w_v).sum(1) + b_v]), rtol=1e-5, atol=1e-6)

    def state():
        return dict(mu=torch.empty(N, A, device=dev), sigma=torch.empty(N, A, device=dev), value=torch.empty(N, 1, device=dev),
                    action=torch.empty(N, A, device=dev), nlp=torch.empty(N, device=dev), shaped=torch.empty(N, 1, device=dev),
                    dones=torch.empty(N, device=dev, dtype=torch.uint8), cur_r=torch.zeros(N, 1, device=dev),
                    cur_l=torch.zeros(N, device=dev), h=torch.ones(1, N, H, device=dev), c=torch.ones(1, N, H, device=dev),
                    hop=torch.ones(N, 352, device=dev), scratch=torch.zeros(ROLLOUT_POST_SCRATCH_FLOATS, device=dev),
                    obs=torch.empty(N, ea.num_obs, device=dev), meter=torch.zeros(8, device=dev))
    sa, sb = state(), state()
    n_done = 0
    for step in range(8):                       # 6-step episodes: every env finishes (time-out bootstrap) inside the run
        y = rnd(N, H)
        for s_ in (sa, sb):
            s_["h"].fill_(1.0); s_["c"].fill_(1.0); s_["hop"].fill_(1.0)
        # ---- A: three launches
        assert lib.vine_policy_head_rms(N, A, H, y.data_ptr(), w_mu.data_ptr(), b_mu.data_ptr(), w_v.data_ptr(), b_v.data_ptr(),
                                        logstd.data_ptr(), vmean.data_ptr(), vvar.data_ptr(), 1e-5, 12345, counter.data_ptr(),
                                        sa["mu"].data_ptr(), sa["sigma"].data_ptr(), sa["value"].data_ptr(), sa["action"].data_ptr(),
                                        sa["nlp"].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, st) == 0
        ea.step_into(sa["action"], sa["obs"])
        assert lib.vine_rollout_post_defer(N, H, ea.rew_buf.data_ptr(), ea.reset_buf.data_ptr(), ea.timeout_buf.data_ptr(),
                                           sa["value"].data_ptr(), 0.0, 0.01, 0.99, sa["shaped"].data_ptr(), sa["dones"].data_ptr(),
                                           sa["cur_r"].data_ptr(), sa["cur_l"].data_ptr(), sa["h"].data_ptr(), sa["c"].data_ptr(),
                                           sa["hop"].data_ptr() + 4 * 96, 352, 0, sa["scratch"].data_ptr(), st) == 0
        # ---- B: one launch
        ra = abi.RolloutArgs()
        ra.y, ra.hw, ra.hc, ra.logstd = y.data_ptr(), hw.data_ptr(), hc.data_ptr(), logstd.data_ptr()
        ra.value_mean, ra.value_var, ra.ln_eps, ra.value_eps = vmean.data_ptr(), vvar.data_ptr(), 1e-5, 1e-5
        ra.seed, ra.counter = 12345, counter.data_ptr()
        ra.mu_out, ra.sigma_out, ra.value_out = sb["mu"].data_ptr(), sb["sigma"].data_ptr(), sb["value"].data_ptr()
        ra.action_out, ra.neglogp_out = sb["action"].data_ptr(), sb["nlp"].data_ptr()
        ra.reward_shift, ra.reward_scale, ra.gamma_bootstrap = 0.0, 0.01, 0.99
        ra.shaped_out, ra.dones_out = sb["shaped"].data_ptr(), sb["dones"].data_ptr()
        ra.cur_rewards, ra.cur_lengths = sb["cur_r"].data_ptr(), sb["cur_l"].data_ptr()
        ra.h_state, ra.c_state, ra.h_op, ra.h_op_stride = sb["h"].data_ptr(), sb["c"].data_ptr(), sb["hop"].data_ptr() + 4 * 96, 352
        ra.partial = sb["scratch"].data_ptr()
        eb.step_rollout_into(ra, sb["obs"])
        torch.cuda.synchronize()
        for k in ("mu", "value", "action", "nlp"):
            torch.testing.assert_close(sb[k], sa[k], rtol=2e-6, atol=2e-6, msg="step %d %s" % (step, k))
        assert torch.equal(sb["sigma"], sa["sigma"])
        assert torch.equal(eb.reset_buf, ea.reset_buf) and torch.equal(eb.timeout_buf, ea.timeout_buf), step
        assert torch.equal(eb.progress_buf, ea.progress_buf) and torch.equal(sb["dones"], sa["dones"])
        torch.testing.assert_close(sb["obs"], sa["obs"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(eb.rew_buf, ea.rew_buf, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(sb["shaped"], sa["shaped"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sb["cur_r"], sa["cur_r"], rtol=1e-4, atol=1e-4)
        assert torch.equal(sb["cur_l"], sa["cur_l"])
        for k in ("h", "c", "hop"):
            assert torch.equal(sb[k], sa[k]), (step, k)          # rows of finished envs cleared, nothing else touched
        done = ea.reset_buf != 0
        assert float(sb["h"][0][done].abs().max() if done.any() else 0.0) == 0.0
        pa = sa["scratch"][:lib.vine_rollout_post_blocks(N) * 3].view(-1, 3).sum(0)
        pb = sb["scratch"][:eb.rollout_step_blocks() * 3].view(-1, 3).sum(0)
        torch.testing.assert_close(pb, pa, rtol=1e-4, atol=1e-4)
        assert float(pb[2]) == float(done.sum())
        n_done += int(done.sum())
        counter += 1
    assert int(ea.progress_buf.max()) <= 6 and n_done >= N          # every env finished an episode (time-outs) inside the run
    # the fused entry refuses what it does not cover
    bad = abi.RolloutArgs()
    assert lib.vine_step_rollout(eb._handle, C.addressof(bad), sb["obs"].data_ptr(), eb.rew_buf.data_ptr(), eb.reset_buf.data_ptr(),
                                 eb.progress_buf.data_ptr(), eb.timeout_buf.data_ptr(), st) == -1
    ea.close(); eb.close()


# --------------------------------------------------------------------------- GradScaler semantics on the device
@pytest.mark.gpu
def test_adam_amp_is_gradscaler_step_and_update():
    """vine_adam_step_amp = torch.amp.GradScaler's step + update around torch.optim.Adam (rl_games wraps the reference's
    ``mixed_precision: True`` update in one): gradients unscaled by 1 / scale, an overflowed step skipped entirely
    (parameters, moments, step counter, 16-bit copies untouched; gradients cleared) with the scale halved, the scale
    doubled after `growth_interval` consecutive good steps, the flag cleared by the launch."""
    from vine_robot_isaacgymenvs_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n = 4099                                    # not a multiple of 4: the scalar tail too
    p0 = torch.randn(n, device=dev)
    p = p0.clone()
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    lr, step = torch.tensor(3e-4, device=dev), torch.zeros((), device=dev)
    shadow = p.to(fused.lp_dtype())
    amp = torch.tensor([1024.0, 0.0, 3.0, 0.0], device=dev)          # scale, tracker, growth interval
    found = torch.zeros(1, device=dev)
    ref_p = torch.nn.Parameter(p0.clone())
    ref = torch.optim.Adam([ref_p], lr=3e-4, eps=1e-8)
    st = torch.cuda.current_stream().cuda_stream

    def launch(g_unscaled, overflow):
        g = (g_unscaled * float(amp[0])).contiguous()
        if overflow:
            g[7] = float("inf")
            found.fill_(1.0)
        rc = lib.vine_adam_step_amp(n, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), lr.data_ptr(), step.data_ptr(),
                                    0.9, 0.999, 1e-8, 0.0, 1.0, shadow.data_ptr(), None, 0.0, 0.0, 0.0, 0.0, amp.data_ptr(),
                                    found.data_ptr(), st)
        assert rc == 0
        torch.cuda.synchronize()
        assert float(g.abs().max()) == 0.0 and float(found) == 0.0       # gradient block and flag cleared either way
        return g

    scales = []
    for k, overflow in enumerate([False, True, False, False, False, False]):
        g = torch.randn(n, device=dev) * 1e-3
        before = (p.clone(), m.clone(), v.clone(), float(step), shadow.clone())
        launch(g, overflow)
        if overflow:
            assert torch.equal(p, before[0]) and torch.equal(m, before[1]) and torch.equal(v, before[2])
            assert float(step) == before[3] and torch.equal(shadow, before[4])
        else:
            ref_p.grad = g.clone()
            ref.step()
            assert float((p - ref_p.detach()).abs().max()) < 2e-6
            assert torch.equal(shadow, p.to(fused.lp_dtype()))
        scales.append((float(amp[0]), float(amp[1])))
    # good -> tracker 1; overflow -> scale / 2, tracker 0; three good steps -> scale x 2 at the third, tracker 0; good -> 1
    assert scales == [(1024.0, 1.0), (512.0, 0.0), (512.0, 1.0), (512.0, 2.0), (1024.0, 0.0), (1024.0, 1.0)], scales
    assert float(step) == 5.0


@pytest.mark.gpu
def test_loss_scale_and_overflow_flags():
    """The loss scale multiplies every gradient of vine_ln_heads_loss (dx, the LayerNorm / head partial sums, the head
    bias and log-sigma gradients), not its statistics; a 16-bit dx that overflows the format raises found_inf; the
    batched column sums raise it for non-finite results."""
    from vine_robot_isaacgymenvs_amd import native
    from vine_robot_isaacgymenvs_amd.abi import PPO_LOSS_SCRATCH_FLOATS, PPO_PARTIAL_BLOCKS
    lib = native.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    H, A = 256, 2
    NH = A + 1
    n = 4 * lib.vine_ln_heads_loss_rows()
    x = torch.randn(n, H, device=dev)
    gamma, beta = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
    w, wb = torch.randn(NH, H, device=dev) * 0.05, torch.randn(NH, device=dev) * 0.1
    logstd = torch.randn(A, device=dev) * 0.1
    actions, old_mu = torch.randn(n, A, device=dev), torch.randn(n, A, device=dev) * 0.3
    old_sigma = torch.rand(n, A, device=dev) + 0.5
    old_nlp, adv = torch.randn(n, device=dev) * 0.3 + 2.0, torch.randn(n, device=dev)
    old_values, returns = torch.randn(n, device=dev), torch.randn(n, device=dev)
    scal = (0.2, 1, 2.0, 0.0, 1e-4, 1.1)
    st = torch.cuda.current_stream().cuda_stream

    def run(scale, lp16):
        heads = torch.empty(n, NH, device=dev)
        dx = torch.empty(n, H, device=dev, dtype=fused.lp_dtype() if lp16 else torch.float32)
        part = torch.empty(PPO_PARTIAL_BLOCKS, 2 * H + NH * H, device=dev)
        stats, gls = torch.zeros(8, device=dev), torch.zeros(A, device=dev)
        gmb, gvb, acc = torch.zeros(A, device=dev), torch.zeros(1, device=dev), torch.zeros(A, device=dev)
        scratch = torch.empty(PPO_LOSS_SCRATCH_FLOATS, device=dev)
        s_t = torch.tensor([scale], device=dev) if scale is not None else None
        found = torch.zeros(1, device=dev)
        rc = lib.vine_ln_heads_loss(n, H, NH, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(), wb.data_ptr(),
                                    logstd.data_ptr(), actions.data_ptr(), old_nlp.data_ptr(), adv.data_ptr(),
                                    old_values.data_ptr(), returns.data_ptr(), old_mu.data_ptr(), old_sigma.data_ptr(), *scal,
                                    heads.data_ptr(), dx.data_ptr(), int(lp16), part.data_ptr(), stats.data_ptr(), gls.data_ptr(),
                                    gmb.data_ptr(), gvb.data_ptr(), scratch.data_ptr(), None, acc.data_ptr(), None, None,
                                    s_t.data_ptr() if s_t is not None else None, found.data_ptr(), st)
        assert rc == 0
        torch.cuda.synchronize()
        return dict(dx=dx.float(), part=part[:n // lib.vine_ln_heads_loss_rows()].clone(), stats=stats, gls=gls, gmb=gmb,
                    gvb=gvb, acc=acc, found=float(found))

    base, scaled = run(None, False), run(4096.0, False)
    assert torch.equal(base["stats"], scaled["stats"]) and base["found"] == scaled["found"] == 0.0
    for k in ("dx", "part", "gls", "gmb", "gvb", "acc"):        # a power-of-two scale: exact
        assert torch.equal(scaled[k], base[k] * 4096.0), k
    if fused.lp_dtype() == torch.float16:
        ok = run(4096.0, True)
        assert ok["found"] == 0.0 and float((ok["dx"] - scaled["dx"]).abs().max()) <= 1e-3 * float(scaled["dx"].abs().max())
        big = 65504.0 * 4.0 / float(base["dx"].abs().max())      # pushes the largest dx element out of the fp16 range
        assert run(big, True)["found"] == 1.0 and run(big, False)["found"] == 0.0
    # column sums: a non-finite result raises the flag, finite ones leave it alone
    src = torch.randn(64, 512, device=dev)
    out = torch.empty(512, device=dev)
    found = torch.zeros(1, device=dev)
    b = fused.ColumnSumBatch(found_inf=found)
    b.add(src, out)
    b.flush(src)
    torch.cuda.synchronize()
    assert float(found) == 0.0 and torch.allclose(out, src.sum(0), atol=1e-4)
    src[3, 100] = float("nan")
    b = fused.ColumnSumBatch(found_inf=found)
    b.add(src, out)
    b.flush(src)
    torch.cuda.synchronize()
    assert float(found) == 1.0


@pytest.mark.gpu
def test_fp16_update_recovers_from_an_absurd_loss_scale():
    """End to end: the default mixed-precision update (fp16 operands, device-side GradScaler) started at a loss scale that
    overflows float16: the first optimiser steps are skipped (parameters unchanged) while the scale backs off, then
    training proceeds; nothing ever becomes non-finite, and the learning-rate schedule keeps running."""
    if fused.lp_dtype() != torch.float16:
        pytest.skip("bf16 build: no loss scaling")
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    cfg = load_config(overrides=["num_envs=512", "minibatch_size=2048"])
    cfg["task"]["seed"] = 42
    env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                  graphics_device_id=0, headless=True)
    params = cfg["train"]["params"]
    params["config"].update(write_files=False, print_stats=False, use_graphs=True, mixed_precision=True,
                            loss_scale_init=2.0 ** 40)
    torch.manual_seed(0)
    agent = A2CAgent("t", params, vec_env=env)
    assert agent.fused_mixed and agent.optimizer.amp_state is not None and agent._fast_op_is_fp32()
    agent.init_tensors()
    agent.obs = agent.env_reset()["obs"]
    p0 = agent.optimizer.flat_params.clone()
    scales, steps = [], []
    for _ in range(4):
        _, _, stats = agent.train_epoch()
        torch.cuda.synchronize()
        scales.append(agent.optimizer.loss_scale)
        steps.append(float(agent.optimizer.step_t))
        assert torch.isfinite(agent.optimizer.flat_params).all() and all(np.isfinite(float(v)) for v in stats.values())
    n_steps = 4 * agent.mini_epochs_num * agent.num_minibatches
    assert scales[0] < 2.0 ** 40 and steps[-1] < n_steps            # some steps were skipped while the scale backed off
    assert steps[-1] > 0 and not torch.equal(agent.optimizer.flat_params, p0)     # ... and then it trained
    assert scales[-1] >= 1.0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("obs_type", ["POS_AND_FD_VEL_AND_OBJ_INFO", "TIP_AND_CART_AND_OBJ_INFO"])
def test_operand_preparation_riding_in_the_mlp_launch_is_bit_identical(obs_type, monkeypatch):
    """Round 4: the optimiser step's parameter-derived operands (LSTM weight tiles, transposed weights, merged heads, bias
    sum) are built by the workgroups of the one-launch MLP as a side job (vine_mlp3_elu_mfma_prep), which also pads W1
    [256, F_in] itself (F_in = 28 and 18) -- same bytes as the former copy_batched launch in front of it: training with
    and without the ride is bit-identical, graph-replayed included."""
    if fused.lp_dtype() != torch.float16:
        pytest.skip("bf16 build")
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map

    def run(ride):
        monkeypatch.setattr(fused, "MLP3_PREP", ride)
        # (a 16384-sample minibatch: the MLP launch must be large enough to carry the ~720 blocks of moves)
        cfg = load_config(overrides=["num_envs=2048", "minibatch_size=16384", "OBSERVATION_TYPE=%s" % obs_type])
        cfg["task"]["seed"] = 42
        env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
        params = cfg["train"]["params"]
        params["config"].update(write_files=False, print_stats=False, use_graphs=True, mixed_precision=True)
        torch.manual_seed(0)
        agent = A2CAgent("t", params, vec_env=env)
        assert agent.fused_mixed
        agent.init_tensors()
        agent.obs = agent.env_reset()["obs"]
        rides0 = fused.RIDES[0]
        for _ in range(4):          # (iterations 3 and 4 replay the captured update)
            agent.train_epoch()
        torch.cuda.synchronize()
        assert agent.graph_status["update"].startswith("graph")
        assert (fused.RIDES[0] > rides0) == ride          # the side job really ran (or really did not)
        out = agent.optimizer.flat_params.clone()
        env.close()
        return out

    with_ride, without = run(True), run(False)
    assert torch.isfinite(with_ride).all() and torch.equal(with_ride, without)


@pytest.mark.gpu
def test_backward_phases_in_one_launch_are_bit_identical(monkeypatch):
    """Round 4: the persistent LSTM backward and the one-launch MLP backward as two phases of ONE launch
    (vine_lstm_seq_backward_mlp3_mfma: workgroup b owns sequences [32 b, 32 b + 32) = rows [128 b, 128 b + 128) in both)
    against the two launches: same arithmetic on the same rows, so training is bit-identical (32768-sample minibatches: the
    fused form needs the 128-row workgroups of the MLP phase)."""
    if fused.lp_dtype() != torch.float16:
        pytest.skip("bf16 build")
    from vine_robot_isaacgymenvs_amd import load_config, native
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    lib = native.load()
    calls = {"n": 0}
    real = lib.vine_lstm_seq_backward_mlp3_mfma

    def run(phases, trunk=False):
        monkeypatch.setattr(fused, "BWD_PHASES", phases)
        monkeypatch.setattr(fused, "TRUNK_PHASES", trunk)
        cfg = load_config(overrides=["num_envs=2048", "minibatch_size=32768"])
        cfg["task"]["seed"] = 42
        env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
        params = cfg["train"]["params"]
        params["config"].update(write_files=False, print_stats=False, use_graphs=False, mixed_precision=True)
        torch.manual_seed(0)
        agent = A2CAgent("t", params, vec_env=env)
        agent.init_tensors()
        agent.obs = agent.env_reset()["obs"]
        for _ in range(2):
            agent.train_epoch()
        torch.cuda.synchronize()
        out = agent.optimizer.flat_params.clone()
        env.close()
        return out

    def counting(*a):
        calls["n"] += 1
        return real(*a)
    counting.argtypes, counting.restype = real.argtypes, real.restype
    monkeypatch.setattr(lib, "vine_lstm_seq_backward_mlp3_mfma", counting)
    one = run(True)
    assert calls["n"] == 2 * 4        # 2 iterations x 4 mini-epochs x 1 minibatch: the fused launch really ran
    two = run(False)
    assert calls["n"] == 8
    assert torch.isfinite(one).all() and torch.equal(one, two)
    # ... and the four-phase launch (LSTM forward + LayerNorm / heads / loss + LSTM backward + MLP backward: vine_trunk_phases)
    p0 = fused.PHASE_LAUNCHES[0]
    four = run(True, trunk=True)
    assert fused.PHASE_LAUNCHES[0] - p0 == 8 and calls["n"] == 8
    assert torch.equal(four, two)


@pytest.mark.gpu
def test_truncate_grads_with_the_fp16_fused_update():
    """``truncate_grads: True`` on the default mixed-precision update: the loss-scaled gradient block is
    clipped against its UNSCALED norm.  A threshold nothing reaches leaves training bit-identical to ``truncate_grads:
    False`` (round 3 clipped scale * g: every step was clipped and the updates collapsed); a tight threshold clips, and the
    clipped run's first Adam step still moves every parameter by about the learning rate."""
    if fused.lp_dtype() != torch.float16:
        pytest.skip("bf16 build: no loss scaling")
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map

    def run(truncate, grad_norm):
        cfg = load_config(overrides=["num_envs=512", "minibatch_size=2048"])
        cfg["task"]["seed"] = 42
        env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                      graphics_device_id=0, headless=True)
        params = cfg["train"]["params"]
        params["config"].update(write_files=False, print_stats=False, use_graphs=False, mixed_precision=True,
                                truncate_grads=truncate, grad_norm=grad_norm)
        torch.manual_seed(0)
        agent = A2CAgent("t", params, vec_env=env)
        assert agent.fused_mixed and agent.optimizer.amp_state is not None
        agent.init_tensors()
        agent.obs = agent.env_reset()["obs"]
        p0 = agent.optimizer.flat_params.clone()
        for _ in range(2):
            agent.train_epoch()
        torch.cuda.synchronize()
        out = agent.optimizer.flat_params.clone(), p0, float(agent.optimizer.step_t)
        env.close()
        return out

    free, p0, steps = run(False, 1.0)
    loose, _, steps_l = run(True, 1e9)
    tight, _, steps_t = run(True, 1e-3)
    assert steps == steps_l == steps_t > 0
    assert torch.equal(free, loose)                       # clip coefficient exactly 1: nothing changes
    assert not torch.equal(free, tight)                   # the tight threshold does clip
    moved = (tight - p0).abs()
    assert torch.isfinite(tight).all() and float(moved.max()) > 1e-5      # Adam normalises: clipped steps still move


# --------------------------------------------------------------------------- fp32 matrix-core rollout kernels
@pytest.mark.gpu
@pytest.mark.parametrize("F", [28, 18])
def test_mlp3_elu_f32_against_float64_torch(F):
    """vine_mlp3_elu_f32 (observation normalisation + three ELU layers on v_mfma_f32_16x16x4_f32, fp32 operands and
    accumulation: the reference's rollout precision) against the float64 torch composition; also the normalised
    observation block it writes behind the MLP output, zero-padded to 32 columns.  fp32 tolerance."""
    from vine_robot_isaacgymenvs_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(F)
    n, ldx = 1024, 352
    raw = torch.randn(n, F, device=dev) * 2.0 + 0.3
    mean = torch.randn(F, device=dev, dtype=torch.float64) * 0.2
    var = torch.rand(F, device=dev, dtype=torch.float64) + 0.3
    Ws = [torch.randn(o, i, device=dev) / np.sqrt(i) for o, i in ((256, F), (128, 256), (64, 128))]
    bs = [torch.randn(o, device=dev) * 0.1 for o in (256, 128, 64)]
    x = torch.full((n, ldx), 7.0, device=dev)
    w1p = torch.zeros(256, 32, device=dev)
    w1p[:, :F] = Ws[0]
    rc = lib.vine_mlp3_elu_f32(n, x.data_ptr(), ldx, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                               w1p.data_ptr(), 32, bs[0].data_ptr(), 256, Ws[1].data_ptr(), Ws[1].stride(0),
                               bs[1].data_ptr(), 128, Ws[2].data_ptr(), Ws[2].stride(0), bs[2].data_ptr(), 64, 1.0,
                               torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    xn = torch.clamp((raw - mean.float()) / torch.sqrt(var.float() + 1e-5), -5.0, 5.0)      # vine_normalize_obs' arithmetic
    a = xn.double()
    for W, b in zip(Ws, bs):
        a = torch.nn.functional.elu(a @ W.double().t() + b.double())
    assert float((x[:, :64].double() - a).abs().max()) < 2e-5 * max(1.0, float(a.abs().max()))
    assert torch.equal(x[:, 64:64 + F], xn) and float(x[:, 64 + F:96].abs().max()) == 0.0
    assert float((x[:, 96:] - 7.0).abs().max()) == 0.0                    # nothing else touched
    assert lib.vine_mlp3_elu_f32(n + 8, x.data_ptr(), ldx, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                 w1p.data_ptr(), 32, bs[0].data_ptr(), 256, Ws[1].data_ptr(), 256, bs[1].data_ptr(), 128,
                                 Ws[2].data_ptr(), 128, bs[2].data_ptr(), 64, 1.0, torch.cuda.current_stream().cuda_stream) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("F,n,terms", [(28, 1024, 9 + 256 * 1), (18, 1024, 9 + 256 * 2), (28, 4096, 9), (28, 16384, 9),
                                       (18, 8192, 9), (28, 2048, 6 + 256 * 4), (28, 4096, 6),
                                       (28, 16384, 6 + (1 << 16)), (18, 4096, 6 + (1 << 16)), (28, 1024, 6 + 256 + (1 << 16)),
                                       (28, 2048, 9 + 256 * 2 + (1 << 16)), (18, 8192, 9 + (1 << 16))])
def test_mlp3_elu_f32_split_against_float64_torch(F, n, terms):
    """vine_mlp3_elu_f32_split (products formed exactly from bf16 pieces on the bf16 matrix cores, four waves sharing the rows
    and splitting the units, activations exchanged through LDS as pieces) against the float64 torch composition: held to
    the native fp32 kernel's bound AND to that kernel's own error on the same inputs; the observation block is bit-identical
    to the native kernel's.  ``terms``: piece pairs in the low byte, row tiles per workgroup in the second (0: from n);
    bit 16: two accumulators per tile (what the rollout runs) -- then BOTH the max and the rms error must be at or below the
    native kernel's (the criterion the 6-pair default rests on)."""
    from vine_robot_isaacgymenvs_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(F + n)
    ldx = 352
    raw = torch.randn(n, F, device=dev) * 2.0 + 0.3
    mean = torch.randn(F, device=dev, dtype=torch.float64) * 0.2
    var = torch.rand(F, device=dev, dtype=torch.float64) + 0.3
    # weights spanning several binades, so that a piece rounded away would show
    Ws = [torch.randn(o, i, device=dev) / np.sqrt(i) * torch.exp2(torch.randint(-6, 2, (o, i), device=dev).float())
          for o, i in ((256, F), (128, 256), (64, 128))]
    bs = [torch.randn(o, device=dev) * 0.1 for o in (256, 128, 64)]
    st = torch.cuda.current_stream().cuda_stream
    wt = torch.empty(288 * 512, device=dev, dtype=torch.bfloat16)
    assert lib.vine_mlp3_tile_weights_split(Ws[0].data_ptr(), Ws[0].stride(0), F, Ws[1].data_ptr(), Ws[1].stride(0),
                                            Ws[2].data_ptr(), Ws[2].stride(0), wt.data_ptr(), st) == 0
    # the pieces of a fragment sum back to the fp32 weights (layer 2: [wave][k-block][tile][piece][lane][8])
    l2 = wt[48 * 512:(48 + 192) * 512].view(4, 8, 2, 3, 64, 8).float().sum(dim=3)
    wv, kb, t, lane, e = torch.meshgrid(*(torch.arange(k, device=dev) for k in l2.shape), indexing="ij")
    assert torch.equal(l2, Ws[1][32 * wv + 16 * t + (lane & 15), 32 * kb + 8 * (lane >> 4) + e])
    x = torch.full((n, ldx), 7.0, device=dev)
    rc = lib.vine_mlp3_elu_f32_split(n, x.data_ptr(), ldx, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                     wt.data_ptr(), bs[0].data_ptr(), bs[1].data_ptr(), bs[2].data_ptr(), 1.0, terms,
                                     None, 0.0, None, None, 0, st)
    assert rc == 0
    # the native fp32 matrix-core kernel on the same inputs
    x_ref = torch.full((n, ldx), 7.0, device=dev)
    w1p = torch.zeros(256, 32, device=dev)
    w1p[:, :F] = Ws[0]
    assert lib.vine_mlp3_elu_f32(n, x_ref.data_ptr(), ldx, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                 w1p.data_ptr(), 32, bs[0].data_ptr(), 256, Ws[1].data_ptr(), Ws[1].stride(0),
                                 bs[1].data_ptr(), 128, Ws[2].data_ptr(), Ws[2].stride(0), bs[2].data_ptr(), 64, 1.0, st) == 0
    torch.cuda.synchronize()
    xn = torch.clamp((raw - mean.float()) / torch.sqrt(var.float() + 1e-5), -5.0, 5.0)
    a = xn.double()
    for W, b in zip(Ws, bs):
        a = torch.nn.functional.elu(a @ W.double().t() + b.double())
    err = float((x[:, :64].double() - a).abs().max())
    err_ref = float((x_ref[:, :64].double() - a).abs().max())
    rms, rms_ref = float((x[:, :64].double() - a).pow(2).mean().sqrt()), float((x_ref[:, :64].double() - a).pow(2).mean().sqrt())
    print("terms 0x%x: max |y - f64| %.3e (native fp32 MFMA kernel %.3e), rms %.3e (native %.3e), max |y| %.2f"
          % (terms, err, err_ref, rms, rms_ref, float(a.abs().max())))
    assert err < 2e-5 * max(1.0, float(a.abs().max()))
    assert err <= 1.5 * err_ref + 1e-7
    assert rms <= rms_ref                 # (not above the native fp32 matrix-core kernel's rms error)
    if terms >> 16:
        assert err <= err_ref and rms <= 0.6 * rms_ref
    if (terms & 0xff) == 6:               # ... and the 6-pair form as accurate as the exact-product 9-pair form
        x9 = torch.full((n, ldx), 7.0, device=dev)
        assert lib.vine_mlp3_elu_f32_split(n, x9.data_ptr(), ldx, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0,
                                           wt.data_ptr(), bs[0].data_ptr(), bs[1].data_ptr(), bs[2].data_ptr(), 1.0,
                                           (terms & ~0xff) | 9, None, 0.0, None, None, 0, st) == 0
        torch.cuda.synchronize()
        e9 = float((x9[:, :64].double() - a).abs().max())
        r9 = float((x9[:, :64].double() - a).pow(2).mean().sqrt())
        assert err <= (1.3 if terms >> 16 else 1.15) * e9 + 1e-9 and rms <= 1.01 * r9
    assert torch.equal(x[:, 64:], x_ref[:, 64:])                       # observation block; nothing else touched
    assert torch.equal(x[:, 64:64 + F], xn) and float(x[:, 64 + F:96].abs().max()) == 0.0
    bad = lib.vine_mlp3_elu_f32_split(n + 8, x.data_ptr(), ldx, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5,
                                      5.0, wt.data_ptr(), bs[0].data_ptr(), bs[1].data_ptr(), bs[2].data_ptr(), 1.0, terms,
                                      None, 0.0, None, None, 0, st)
    assert bad == -2


@pytest.mark.gpu
def test_lstm_step_f32_against_float64_torch():
    """vine_lstm_step_f32 (gate GEMM over [x | h] on the fp32 matrix cores + the cell update as its epilogue) against
    float64 torch with torch.nn.LSTM's gate order; the second copy of h (next step's operand block) too."""
    from vine_robot_isaacgymenvs_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    N, H, K = 1024, 256, 352
    xh = torch.randn(N, K, device=dev)
    xh[:, 92:96] = 0.0
    wcat = torch.randn(4 * H, K, device=dev) / np.sqrt(K)
    bias = torch.randn(4 * H, device=dev) * 0.1
    c_prev = torch.randn(N, H, device=dev)
    wt = torch.empty(4 * H * K, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.vine_lstm_tile_weights_f32(H, K, wcat.data_ptr(), wcat.stride(0), wt.data_ptr(), st) == 0
    h_out, c_out = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
    nxt = torch.zeros(N, K, device=dev)
    assert lib.vine_lstm_step_f32(N, H, K, xh.data_ptr(), K, wt.data_ptr(), bias.data_ptr(), c_prev.data_ptr(), h_out.data_ptr(),
                                  H, c_out.data_ptr(), nxt.data_ptr() + 4 * 96, K, st) == 0
    torch.cuda.synchronize()
    g = xh.double() @ wcat.double().t() + bias.double()
    i, f, gg, o = (g[:, k * H:(k + 1) * H] for k in range(4))
    c = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    assert float((c_out.double() - c).abs().max()) < 1e-5 and float((h_out.double() - h).abs().max()) < 1e-5
    assert torch.equal(nxt[:, 96:], h_out) and float(nxt[:, :96].abs().max()) == 0.0
    # in place on the cell state (what the rollout does) gives the same result
    c_io = c_prev.clone()
    assert lib.vine_lstm_step_f32(N, H, K, xh.data_ptr(), K, wt.data_ptr(), bias.data_ptr(), c_io.data_ptr(), h_out.data_ptr(),
                                  H, c_io.data_ptr(), None, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(c_io, c_out)
    assert lib.vine_lstm_step_f32(N + 64, H, K, xh.data_ptr(), K, wt.data_ptr(), bias.data_ptr(), c_prev.data_ptr(),
                                  h_out.data_ptr(), H, c_out.data_ptr(), None, 0, st) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("terms", [9, 6, 9 + (2 << 8), 6 + (4 << 8), 9 + (4 << 8), 9 + (1 << 16), 6 + (1 << 16),
                                   9 + (3 << 16), 6 + (3 << 16)])
def test_lstm_step_f32_split_against_float64_torch(terms):
    """vine_lstm_step_f32_split (fp32 operands split exactly into three bf16 pieces, every piece product exact in the
    fp32 accumulator of the bf16 matrix cores) against float64 torch: the pre-activations must be at least as close to the
    float64 product as those of the native fp32 matrix-core kernel (9 terms: no bit of a product is dropped), and
    h / c within the same 1e-5 the native kernel is held to.  Operand magnitudes span 2^-20 .. 2^6 so that a piece that
    were rounded away would show.  ``terms``: piece pairs in the low byte, row tiles per wave (a tuning knob) in the second.
    Bit 16 of ``terms``: the one-gate-per-wave kernel (four waves share 64 rows, operand pieces exchanged through LDS);
    bit 17: with two accumulators per tile (what the rollout runs) -- then BOTH the max and the rms error of h and c must be
    at or below the native kernel's (the criterion the 6-pair default rests on)."""
    from vine_robot_isaacgymenvs_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    N, H, K = 2048, 256, 352
    xh = torch.randn(N, K, device=dev) * torch.exp2(torch.randint(-20, 3, (N, K), device=dev).float())
    xh[:, 92:96] = 0.0
    wcat = torch.randn(4 * H, K, device=dev) / np.sqrt(K) * torch.exp2(torch.randint(-12, 3, (4 * H, K), device=dev).float())
    bias = torch.randn(4 * H, device=dev) * 0.1
    c_prev = torch.randn(N, H, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(3 * 4 * H * K, device=dev, dtype=torch.bfloat16)
    assert lib.vine_lstm_tile_weights_split(H, K, wcat.data_ptr(), wcat.stride(0), ws.data_ptr(), st) == 0
    # the three pieces sum back to the fp32 weight exactly (checked through the tile order: sums over pieces of a chunk)
    pieces = ws.view(8, K // 32, 3, 8, 64, 8).float()
    back = pieces.sum(dim=2)                                   # [ub, j, tile, lane, e]: exact in fp32 (hi + mid + lo)
    ub, j, tile, lane, e = torch.meshgrid(*(torch.arange(n, device=dev) for n in back.shape), indexing="ij")
    rows = (tile >> 1) * H + ub * 32 + 16 * (tile & 1) + (lane & 15)
    cols = 32 * j + 8 * (lane >> 4) + e
    assert torch.equal(back, wcat[rows, cols])
    h_out, c_out = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
    nxt = torch.zeros(N, K, device=dev)
    assert lib.vine_lstm_step_f32_split(N, H, K, xh.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c_prev.data_ptr(),
                                        h_out.data_ptr(), H, c_out.data_ptr(), nxt.data_ptr() + 4 * 96, K, terms, st) == 0
    torch.cuda.synchronize()
    g = xh.double() @ wcat.double().t() + bias.double()
    i, f, gg, o = (g[:, k * H:(k + 1) * H] for k in range(4))
    c = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    err_c, err_h = float((c_out.double() - c).abs().max()), float((h_out.double() - h).abs().max())
    rms_c, rms_h = float((c_out.double() - c).pow(2).mean().sqrt()), float((h_out.double() - h).pow(2).mean().sqrt())
    # the native fp32 matrix-core kernel on the same operands
    wt = torch.empty(4 * H * K, device=dev)
    assert lib.vine_lstm_tile_weights_f32(H, K, wcat.data_ptr(), wcat.stride(0), wt.data_ptr(), st) == 0
    h_ref, c_ref = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
    assert lib.vine_lstm_step_f32(N, H, K, xh.data_ptr(), K, wt.data_ptr(), bias.data_ptr(), c_prev.data_ptr(), h_ref.data_ptr(),
                                  H, c_ref.data_ptr(), None, 0, st) == 0
    torch.cuda.synchronize()
    ref_c, ref_h = float((c_ref.double() - c).abs().max()), float((h_ref.double() - h).abs().max())
    ref_rms_c, ref_rms_h = float((c_ref.double() - c).pow(2).mean().sqrt()), float((h_ref.double() - h).pow(2).mean().sqrt())
    print("terms 0x%x: max |c - f64| %.3e (native fp32 MFMA %.3e), max |h - f64| %.3e (native %.3e); rms %.3e / %.3e (native %.3e / %.3e)"
          % (terms, err_c, ref_c, err_h, ref_h, rms_c, rms_h, ref_rms_c, ref_rms_h))
    assert err_c < 1e-5 and err_h < 1e-5
    assert err_c <= 1.5 * ref_c + 1e-7 and err_h <= 1.5 * ref_h + 1e-7
    # VERDICT r4 item 7 (the criterion the 6-pair default rests on): the rms error against float64 is NOT ABOVE the native
    # fp32 matrix-core instruction's on the same inputs (the max over 5e5 outputs is a noisy statistic: within 1.5x above)
    assert rms_c <= ref_rms_c and rms_h <= ref_rms_h
    dual = (terms >> 17) & 1
    if dual:
        assert err_c <= ref_c and err_h <= ref_h and rms_c <= 0.75 * ref_rms_c and rms_h <= 0.75 * ref_rms_h
    if (terms & 0xff) == 6:
        # ... and the 6-pair form is as accurate as the exact-product 9-pair form: same max and rms error to 1 %
        h9, c9 = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
        assert lib.vine_lstm_step_f32_split(N, H, K, xh.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c_prev.data_ptr(),
                                            h9.data_ptr(), H, c9.data_ptr(), None, 0, (terms & ~0xff) | 9, st) == 0
        torch.cuda.synchronize()
        e9_c, e9_h = float((c9.double() - c).abs().max()), float((h9.double() - h).abs().max())
        r9_c, r9_h = float((c9.double() - c).pow(2).mean().sqrt()), float((h9.double() - h).pow(2).mean().sqrt())
        tol = 1.3 if dual else 1.01       # (with two accumulators the dropped pairs are a visible share of a SMALLER error)
        assert err_c <= tol * e9_c + 1e-9 and err_h <= tol * e9_h + 1e-9 and rms_c <= 1.01 * r9_c and rms_h <= 1.01 * r9_h
    assert torch.equal(nxt[:, 96:], h_out) and float(nxt[:, :96].abs().max()) == 0.0
    # in place on the cell state (what the rollout does) gives the same result
    c_io = c_prev.clone()
    assert lib.vine_lstm_step_f32_split(N, H, K, xh.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c_io.data_ptr(),
                                        h_out.data_ptr(), H, c_io.data_ptr(), None, 0, terms, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(c_io, c_out)
    assert lib.vine_lstm_step_f32_split(N + 64, H, K, xh.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c_prev.data_ptr(),
                                        h_out.data_ptr(), H, c_out.data_ptr(), None, 0, terms, st) == -2


@pytest.mark.gpu
def test_rollout_post_kernel_against_reference_text_golden():
    """F10 through the HIP kernel (vine_rollout_post): shaped rewards, done flags, episode accumulators after every step and
    the per-step totals of the finished episodes (meter[4..6] = sum of returns, sum of lengths, count) as the reference's
    in-tree play_steps keeps them (common_agent.py:257-316); the LSTM state rows of finished envs are cleared."""
    import os
    from vine_robot_isaacgymenvs_amd import native
    from vine_robot_isaacgymenvs_amd.abi import ROLLOUT_POST_SCRATCH_FLOATS as SF
    lib = native.load()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f10_rollout_bookkeeping.npz"))
    dev = torch.device("cuda:0")
    T, N = g["dones"].shape
    H = 256
    st = torch.cuda.current_stream().cuda_stream
    cur_r, cur_l = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    h, c = torch.ones(N, H, device=dev), torch.ones(N, H, device=dev)
    meter, counter = torch.zeros(8, device=dev), torch.zeros(1, device=dev, dtype=torch.int64)
    scratch = torch.zeros(SF, device=dev)
    tmo, val = torch.zeros(N, device=dev, dtype=torch.uint8), torch.zeros(N, device=dev)
    shaped, dones_out = torch.zeros(N, device=dev), torch.zeros(N, device=dev, dtype=torch.uint8)
    for t in range(T):
        rew = torch.from_numpy(g["rewards"][t, :, 0]).to(dev).contiguous()
        reset = torch.from_numpy(g["dones"][t].astype(np.int64)).to(dev)
        h.fill_(1.0); c.fill_(1.0)
        assert lib.vine_rollout_post(N, H, rew.data_ptr(), reset.data_ptr(), tmo.data_ptr(), val.data_ptr(), 0.0,
                                     float(g["reward_scale"]), 0.0, shaped.data_ptr(), dones_out.data_ptr(), cur_r.data_ptr(),
                                     cur_l.data_ptr(), h.data_ptr(), c.data_ptr(), meter.data_ptr(), 100.0, counter.data_ptr(),
                                     None, 0, 0, scratch.data_ptr(), st) == 0
        torch.cuda.synchronize()
        assert float((shaped.cpu() - torch.from_numpy(g["shaped"][t, :, 0])).abs().max()) < 1e-7
        assert np.array_equal(dones_out.cpu().numpy(), g["dones"][t])
        assert float((cur_r.cpu() - torch.from_numpy(g["cur_rewards"][t, :, 0])).abs().max()) < 1e-5
        assert np.array_equal(cur_l.cpu().numpy(), g["cur_lengths"][t])
        m = meter.cpu().double().numpy()
        assert abs(m[4] - g["finished_return_sum"][t]) < 1e-3 and m[5] == g["finished_length_sum"][t]
        assert m[6] == g["finished_count"][t]
        done = torch.from_numpy(g["dones"][t].astype(bool)).to(dev)
        assert float(h[done].abs().max() if bool(done.any()) else 0.0) == 0.0 and float(h[~done].min()) == 1.0
        assert float(c[done].abs().max() if bool(done.any()) else 0.0) == 0.0 and float(c[~done].min()) == 1.0
    assert int(counter) == T


@pytest.mark.gpu
def test_gae_kernel_against_reference_text_golden():
    """F9 through the HIP kernel (vine_gae: one env per lane, reverse scan in registers)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f9_gae.npz"))
    dev = torch.device("cuda:0")
    T, N = g["rewards"].shape[:2]
    rew = torch.from_numpy(g["rewards"]).to(dev).contiguous()
    val = torch.from_numpy(g["values"][:T]).to(dev).contiguous()
    last_v = torch.from_numpy(g["values"][T]).to(dev).contiguous()
    dones = torch.from_numpy(g["dones"][:T]).to(dev).to(torch.uint8).contiguous()
    last_d = torch.from_numpy(g["dones"][T]).to(dev).to(torch.uint8).contiguous()
    advs, rets = torch.empty_like(rew), torch.empty_like(rew)
    assert fused._lib().vine_gae(T, N, rew.data_ptr(), val.data_ptr(), dones.data_ptr(), last_v.data_ptr(), last_d.data_ptr(),
                                 float(g["gamma"]), float(g["tau"]), advs.data_ptr(), rets.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    ref = torch.from_numpy(g["advs"]).to(dev)
    assert float((advs - ref).abs().max()) < 2e-6 and float((rets - (ref + val)).abs().max()) < 2e-6
