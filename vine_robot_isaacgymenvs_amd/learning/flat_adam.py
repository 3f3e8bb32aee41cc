"""Adam over ONE flat parameter block (torch.optim.Adam arithmetic; rl_games uses ``Adam(lr, eps=1e-8)``,
isaacgymenvs/learning/common_agent.py:80).

All parameters of the model are re-pointed to views of ``flat_params``; gradients, first and second moments are
flat buffers of the same length.  On the GPU the step is one hand-written kernel (``vine_adam_step``) instead of a
multi-tensor launch over 17 small tensors (95 us -> ~5 us per optimiser step on MI355X), it folds in the 1/world
scaling after the gradient all-reduce and re-zeroes the gradient block.  The learning rate is a device scalar so
the adaptive-KL schedule never synchronises with the host.  ``state_dict`` speaks torch.optim.Adam's format.
"""
import torch

ALIGN = 64   # floats


class FlatAdam:
    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        self.num_params = sum(p.numel() for p in self.params)
        # every tensor starts on a 256-byte boundary (vectorised kernels such as LayerNorm's need 16-B aligned
        # weight pointers; the padding stays zero and costs 1.5 % of a 1.6 MB all-reduce)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.flat_params = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        # the gradient block carries ALIGN extra floats (`aux`) that ride along in the multi-GPU all-reduce: the
        # agent puts the minibatch KL there, so gradient and KL averaging cost ONE collective per optimiser step
        self.comm_buffer = torch.zeros(self.numel + ALIGN, device=dev, dtype=torch.float32)
        self.flat_grads = self.comm_buffer[:self.numel]
        self.aux = self.comm_buffer[self.numel:]
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_params[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_params[off:off + n].view_as(p)
            p.grad = self.flat_grads[off:off + n].view_as(p)
        self.exp_avg = torch.zeros_like(self.flat_params)
        self.exp_avg_sq = torch.zeros_like(self.flat_params)
        self.step_t = torch.zeros((), device=dev, dtype=torch.float32)
        self.lr = lr if isinstance(lr, torch.Tensor) else torch.tensor(float(lr), device=dev)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.param_groups = [{"params": self.params, "lr": self.lr, "betas": betas, "eps": eps,
                              "weight_decay": weight_decay}]
        self.shadow = None          # bfloat16 copy of flat_params (GEMM operands of the mixed-precision update)
        self._lib = None
        if dev.type == "cuda":
            from .. import native
            self._lib = native.load()

    def enable_bf16_shadow(self):
        """Keep a bfloat16 copy of the parameter block: written by the Adam kernel itself after every step;
        ``refresh_shadow`` after anything else touched the parameters (checkpoint load, broadcast)."""
        self.shadow = self.flat_params.to(torch.bfloat16)
        return self.shadow

    def refresh_shadow(self):
        if self.shadow is not None:
            self.shadow.copy_(self.flat_params)

    def shadow_of(self, p):
        """The bfloat16 view matching parameter ``p`` (identity lookup)."""
        for q, off in zip(self.params, self.offsets):
            if q is p:
                return self.shadow[off:off + p.numel()].view_as(p)
        raise KeyError("parameter is not managed by this optimiser")

    def zero_grad(self, set_to_none=False):
        self.flat_grads.zero_()

    @torch.no_grad()
    def step(self, grad_scale=1.0, lr_schedule=None):
        """``lr_schedule`` = (kl device scalar, kl_scale, kl_threshold, min_lr, max_lr): rl_games' AdaptiveScheduler
        applied to ``self.lr`` AFTER this step used the old value, by the same launch (GPU path only)."""
        b1, b2 = self.betas
        if self._lib is not None:
            kl, kscale, thr, lo, hi = lr_schedule if lr_schedule is not None else (None, 0.0, 0.0, 0.0, 0.0)
            rc = self._lib.vine_adam_step_sched(self.numel, self.flat_params.data_ptr(), self.flat_grads.data_ptr(),
                                                self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.lr.data_ptr(),
                                                self.step_t.data_ptr(), b1, b2, self.eps, self.weight_decay,
                                                float(grad_scale),
                                                self.shadow.data_ptr() if self.shadow is not None else None,
                                                kl.data_ptr() if kl is not None else None, float(kscale), float(thr),
                                                float(lo), float(hi),
                                                torch.cuda.current_stream(self.flat_params.device).cuda_stream)
            if rc != 0:
                raise RuntimeError("vine_adam_step_sched failed with status %d" % rc)
            return
        assert lr_schedule is None, "the fused learning-rate schedule exists on the GPU path only"
        g = self.flat_grads * grad_scale
        if self.weight_decay:
            g = g + self.weight_decay * self.flat_params
        self.step_t += 1
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** float(self.step_t)
        bc2 = 1 - b2 ** float(self.step_t)
        denom = (self.exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(self.eps)
        self.flat_params.sub_(self.exp_avg / denom * (self.lr / bc1))
        self.flat_grads.zero_()
        self.refresh_shadow()

    # ---- torch.optim.Adam-compatible (de)serialisation
    def state_dict(self):
        state = {}
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": self.step_t.detach().clone().cpu(),
                        "exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
        group = {"lr": float(self.lr), "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_t.fill_(float(st["step"]))
