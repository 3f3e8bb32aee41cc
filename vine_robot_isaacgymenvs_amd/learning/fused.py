"""Fused update-path ops: autograd Functions around the hand-written HIP kernels of ``csrc/ppo_kernels.hip``
(C ABI ``include/vine_ppo.h``) plus GEMM formulations chosen for MI355X.

Why they exist (rocprofv3 of one PPO iteration at 16384 envs, profiles/r01): the stock composition launched ~395
kernels per optimiser step; the dominant GEMMs were the weight gradients ``dW = dY^T X`` with a 32768-long reduction
and a tiny output (1024x92, 64x128, ...), for which the BLAS heuristics pick one 32x32 tile per output block looping
over the whole reduction.  Here:
  * ``splitk_tn``     : ``dY^T X`` as a batched GEMM over S slices of the reduction + a sum (split-K by hand);
  * ``linear``        : ``F.linear`` with that weight gradient;
  * ``lstm_sequence`` : ONE input-projection GEMM for all time steps, per step one recurrent GEMM + one fused
                        pointwise kernel (done-masking folded in); backward mirrors it, weight gradients by split-K;
  * ``ppo_loss``      : the whole PPO loss (surrogate, clipped value loss, bound loss, entropy, KL) forward AND
                        backward in one kernel.
On a CPU tensor every op falls back to the plain PyTorch composition it replaces (used by the gloo tests and as the
fp32 reference of the numerics tests); on a GPU tensor the HIP kernels are mandatory.
"""
import math

import torch
import torch.nn.functional as F


def _lib():
    from .. import native
    return native.load()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with status %d" % (what, rc))


# --------------------------------------------------------------------------- split-K weight gradient
def splitk_tn(dy, x, out=None):
    """``dy^T @ x`` for tall-skinny operands: dy [K, M], x [K, N] -> [M, N] (written into ``out`` when given)."""
    K = dy.shape[0]
    s = 1
    while K % (s * 2) == 0 and K // (s * 2) >= 1024 and s < 64:
        s *= 2
    if s == 1 or not dy.is_cuda:
        return torch.mm(dy.t(), x, out=out) if out is not None else dy.t().mm(x)
    part = torch.bmm(dy.view(s, K // s, dy.shape[1]).transpose(1, 2), x.view(s, K // s, x.shape[1]))
    return torch.sum(part, 0, out=out) if out is not None else part.sum(0)


def _grad_slot(p):
    """The parameter's persistent gradient buffer (a view into the optimiser's flat gradient block) if it can be
    written in place: the custom backward then stores the gradient there directly and returns None, which saves
    autograd's accumulate-add launch per parameter.  Each parameter is used once per forward, so overwrite == add."""
    g = getattr(p, "grad", None)
    return g if (g is not None and g.is_contiguous() and g.is_cuda) else None


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.slots = (_grad_slot(weight), _grad_slot(bias))
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy.mm(weight) if ctx.needs_input_grad[0] else None
        wslot, bslot = ctx.slots
        gw = splitk_tn(gy, x, out=wslot)
        gb = torch.sum(gy, 0, out=bslot) if bslot is not None else gy.sum(0)
        return gx, (None if wslot is not None else gw), (None if bslot is not None else gb)


def linear(x, weight, bias):
    if x.is_cuda and torch.is_grad_enabled() and x.shape[0] >= 4096 and x.dtype == torch.float32:
        return _Linear.apply(x, weight, bias)
    return F.linear(x, weight, bias)


class SplitKLinear(torch.nn.Linear):
    """``nn.Linear`` (same parameters / state-dict keys) whose weight gradient is the split-K formulation."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


# --------------------------------------------------------------------------- LSTM over a short sequence
def _lstm_reference(x, w_ih, w_hh, b_ih, b_hh, h0, c0, dones, T):
    """Plain PyTorch composition (CPU path and numerics reference).  x [B*T, F] sequence-major."""
    B = x.shape[0] // T
    xs = x.view(B, T, -1)
    d = None if dones is None else dones.view(B, T)
    h, c = h0, c0
    outs = []
    for t in range(T):
        if d is not None:
            keep = (1.0 - d[:, t].to(h.dtype)).unsqueeze(-1)
            h, c = h * keep, c * keep
        h, c = torch._VF.lstm_cell(xs[:, t], (h, c), w_ih, w_hh, b_ih, b_hh)
        outs.append(h)
    return torch.stack(outs, 1).reshape(B * T, -1), h, c


class _LSTMSeq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h0, c0, dones, T):
        lib = _lib()
        BT, H = x.shape[0], w_hh.shape[1]
        B = BT // T
        need_grad = any(ctx.needs_input_grad[:5])
        x = x.contiguous()
        ig = x.mm(w_ih.t())                                   # one input projection for every time step
        bias = (b_ih + b_hh).contiguous()
        out = torch.empty((BT, H), device=x.device, dtype=torch.float32)
        c_all = torch.empty((T + 1, B, H), device=x.device, dtype=torch.float32)
        c_all[0].copy_(c0)
        gates = torch.empty((T, B, 4 * H), device=x.device, dtype=torch.float32) if need_grad else None
        h0 = h0.contiguous()
        st = _stream(x)
        d_ptr = dones.data_ptr() if dones is not None else None
        out3 = out.view(B, T, H)
        for t in range(T):
            h_prev = h0 if t == 0 else out3[:, t - 1]
            hg = h_prev.mm(w_hh.t())
            _check(lib.vine_lstm_cell_forward(
                B, H, ig.data_ptr() + 4 * (t * 4 * H), T * 4 * H, hg.data_ptr(), bias.data_ptr(), c_all[t].data_ptr(),
                (d_ptr + t) if d_ptr is not None else None, T, out.data_ptr() + 4 * (t * H), T * H,
                c_all[t + 1].data_ptr(), gates[t].data_ptr() if need_grad else None, st), "vine_lstm_cell_forward")
        ctx.T = T
        ctx.has_dones = dones is not None
        ctx.slots = (_grad_slot(w_ih), _grad_slot(w_hh), _grad_slot(b_ih), _grad_slot(b_hh))
        if need_grad:
            ctx.save_for_backward(x, w_ih, w_hh, h0, out, c_all, gates, dones if dones is not None else x.new_empty(0))
        hT = out3[:, T - 1].contiguous()
        cT = c_all[T].clone()
        ctx.mark_non_differentiable(hT, cT)
        return out, hT, cT

    @staticmethod
    def backward(ctx, g_out, _g_h, _g_c):
        lib = _lib()
        x, w_ih, w_hh, h0, out, c_all, gates, dones = ctx.saved_tensors
        T = ctx.T
        BT, H = out.shape
        B = BT // T
        g_out = g_out.contiguous()
        dG = torch.empty((BT, 4 * H), device=x.device, dtype=torch.float32)
        dG3 = dG.view(B, T, 4 * H)
        dc = [torch.empty((B, H), device=x.device, dtype=torch.float32) for _ in range(2)]
        st = _stream(x)
        d_ptr = dones.data_ptr() if ctx.has_dones else None
        g_rec = None
        dc_next = None
        for t in reversed(range(T)):
            dn = (d_ptr + t + 1) if (d_ptr is not None and t < T - 1) else None
            _check(lib.vine_lstm_cell_backward(
                B, H, g_out.data_ptr() + 4 * (t * H), T * H, g_rec.data_ptr() if g_rec is not None else None,
                dc_next.data_ptr() if dc_next is not None else None, dn, T, gates[t].data_ptr(),
                c_all[t + 1].data_ptr(), c_all[t].data_ptr(), (d_ptr + t) if d_ptr is not None else None, T,
                dG.data_ptr() + 4 * (t * 4 * H), T * 4 * H, dc[t & 1].data_ptr(), st), "vine_lstm_cell_backward")
            dc_next = dc[t & 1]
            if t > 0:
                g_rec = dG3[:, t].mm(w_hh)
        # masked previous hidden state of every step, sequence-major like dG
        hp = torch.cat([h0.unsqueeze(1), out.view(B, T, H)[:, :-1]], dim=1)
        if ctx.has_dones:
            hp = hp * (1.0 - dones.view(B, T, 1).to(hp.dtype))
        hp = hp.reshape(BT, H)
        gx = dG.mm(w_ih) if ctx.needs_input_grad[0] else None
        s_ih, s_hh, s_bi, s_bh = ctx.slots
        g_ih = splitk_tn(dG, x, out=s_ih)
        g_hh = splitk_tn(dG, hp, out=s_hh)
        if s_bi is not None and s_bh is not None:
            torch.sum(dG, 0, out=s_bi)
            s_bh.copy_(s_bi)
            gbi = gbh = None
        else:
            gbi = gbh = dG.sum(0)
        return (gx, None if s_ih is not None else g_ih, None if s_hh is not None else g_hh, gbi, gbh,
                None, None, None, None)


def lstm_sequence(x, w_ih, w_hh, b_ih, b_hh, h0, c0, dones, T):
    """x [B*T, F] (row = seq*T + t), h0/c0 [B, H], dones uint8 [B*T] or None -> (out [B*T, H], hT, cT)."""
    if x.is_cuda and x.dtype == torch.float32:
        if dones is not None:
            dones = dones.to(torch.uint8).contiguous()
        return _LSTMSeq.apply(x, w_ih, w_hh, b_ih, b_hh, h0, c0, dones, T)
    return _lstm_reference(x, w_ih, w_hh, b_ih, b_hh, h0, c0, dones, T)


# --------------------------------------------------------------------------- PPO loss
def ppo_loss_reference(mu, logstd, value, actions, old_neglogp, adv, old_values, returns, old_mu, old_sigma, e_clip,
                       clip_value, critic_coef, entropy_coef, bounds_coef, soft_bound=1.1):
    """The stock composition (a2c_continuous.calc_gradients); returns (loss, stats dict)."""
    sigma = torch.exp(logstd)
    neglogp = (0.5 * (((actions - mu) / sigma) ** 2).sum(-1) + 0.5 * math.log(2.0 * math.pi) * actions.shape[-1]
               + logstd.expand_as(mu).sum(-1))
    ratio = torch.exp(old_neglogp - neglogp)
    a_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - e_clip, 1.0 + e_clip)).mean()
    v = value.view(-1)
    ov, r = old_values.view(-1), returns.view(-1)
    if clip_value:
        vc = ov + (v - ov).clamp(-e_clip, e_clip)
        c_loss = torch.max((v - r) ** 2, (vc - r) ** 2).mean()
    else:
        c_loss = ((r - v) ** 2).mean()
    b_loss = (torch.clamp_min(mu - soft_bound, 0.0) ** 2 + torch.clamp_max(mu + soft_bound, 0.0) ** 2).sum(-1).mean()
    entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd.expand_as(mu)).sum(-1).mean()
    loss = a_loss + 0.5 * c_loss * critic_coef - entropy * entropy_coef + b_loss * bounds_coef
    with torch.no_grad():
        s = sigma.expand_as(mu)
        c1 = torch.log(old_sigma / s + 1e-5)
        c2 = (s ** 2 + (old_mu - mu) ** 2) / (2.0 * (old_sigma ** 2 + 1e-5))
        kl = (c1 + c2 - 0.5).sum(-1).mean()
    return loss, {"a_loss": a_loss.detach(), "c_loss": c_loss.detach(), "b_loss": b_loss.detach(),
                  "entropy": entropy.detach(), "kl": kl, "loss": loss.detach()}


def ppo_loss_fused(mu, logstd, value, actions, old_neglogp, adv, old_values, returns, old_mu, old_sigma, e_clip,
                   clip_value, critic_coef, entropy_coef, bounds_coef, soft_bound=1.1):
    """One HIP kernel: returns (grad_mu [n,A], grad_value [n,1], grad_logstd [A], stats[8]) where the gradients are
    d(loss)/d(.) of the same scalar loss as ``ppo_loss_reference``."""
    lib = _lib()
    n, A = mu.shape
    mu_c = mu.detach().contiguous()
    val_c = value.detach().reshape(-1).contiguous()
    grad_mu = torch.empty_like(mu_c)
    grad_value = torch.empty_like(val_c)
    grad_logstd = torch.empty(A, device=mu.device, dtype=torch.float32)
    stats = torch.empty(8, device=mu.device, dtype=torch.float32)
    args = [t.detach().contiguous() for t in (actions, old_neglogp, adv, old_values.reshape(-1), returns.reshape(-1),
                                              old_mu, old_sigma)]
    ls = logstd.detach().contiguous()
    _check(lib.vine_ppo_loss(n, A, mu_c.data_ptr(), ls.data_ptr(), val_c.data_ptr(), *[a.data_ptr() for a in args],
                             float(e_clip), int(bool(clip_value)), float(critic_coef), float(entropy_coef),
                             float(bounds_coef), float(soft_bound), grad_mu.data_ptr(), grad_value.data_ptr(),
                             grad_logstd.data_ptr(), stats.data_ptr(), _stream(mu)), "vine_ppo_loss")
    return grad_mu, grad_value.view_as(value), grad_logstd, stats
