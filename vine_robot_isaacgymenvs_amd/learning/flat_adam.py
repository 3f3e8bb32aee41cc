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
        # agent puts the minibatch KL there (aux[0]) and the loss-scaling overflow flag lives there (aux[1]: any rank's
        # overflow reaches every rank with the gradients), so all of it costs ONE collective per optimiser step
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
        self.shadow = None          # 16-bit copy of flat_params (GEMM operands of the mixed-precision update)
        self.amp_state = None       # [loss scale, growth tracker, growth interval, -] (enable_loss_scaling)
        self.found_inf = self.aux[1:2]
        self.check_grads = False    # look for non-finite gradients here (set when the backward pass does not check itself)
        self._lib = None
        if dev.type == "cuda":
            from . import fused
            self._lib = fused._lib()        # (also readies the library's per-device bookkeeping outside any capture)

    def enable_lp16_shadow(self, dtype=None):
        """Keep a 16-bit copy of the parameter block (the library's operand format: float16, or bfloat16 in a
        -DVINE_LP_BF16 build): written by the Adam kernel itself after every step; ``refresh_shadow`` after anything
        else touched the parameters (checkpoint load, broadcast)."""
        if dtype is None:
            from . import fused
            dtype = fused.lp_dtype()
        self.shadow = self.flat_params.to(dtype)
        return self.shadow

    enable_bf16_shadow = enable_lp16_shadow      # round-2 name

    def enable_loss_scaling(self, init_scale=65536.0, growth_interval=2000):
        """torch.amp.GradScaler restated on the device (its defaults: scale 2^16, x2 after 2000 good steps, x0.5 and a
        skipped step on overflow) -- what rl_games wraps the reference's ``mixed_precision: True`` update in.  The
        loss kernels multiply their gradients by ``amp_state[0]`` and flag overflows in ``found_inf``; ``step``
        unscales, skips and adapts, all inside the Adam launch (no host synchronisation, graph-capturable)."""
        self.amp_state = torch.tensor([float(init_scale), 0.0, float(growth_interval), 0.0], device=self.flat_params.device,
                                      dtype=torch.float32)
        return self.amp_state[0:1], self.found_inf

    @property
    def loss_scale(self):
        return None if self.amp_state is None else float(self.amp_state[0])

    def refresh_shadow(self):
        if self.shadow is not None:
            self.shadow.copy_(self.flat_params)

    def shadow_of(self, p):
        """The 16-bit view matching parameter ``p`` (identity lookup)."""
        for q, off in zip(self.params, self.offsets):
            if q is p:
                return self.shadow[off:off + p.numel()].view_as(p)
        raise KeyError("parameter is not managed by this optimiser")

    def zero_grad(self, set_to_none=False):
        self.flat_grads.zero_()

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm):
        """``torch.nn.utils.clip_grad_norm_`` over the flat gradient block (the padding is zero).  With device-side loss
        scaling the block holds ``scale * g`` until the Adam kernel unscales it, so the threshold applies to the UNSCALED
        norm -- rl_games calls ``scaler.unscale_`` before clipping (``truncate_grads: True``); no host synchronisation.
        A non-finite norm (an inf / NaN anywhere in the block) would give coef = 0 or NaN: finite entries become 0 and Adam
        would take a zero-gradient step that decays the moments -- rl_games' unscale_ + clip + scaler.step SKIPS such a step.
        With loss scaling the overflow flag is therefore raised here (whoever delivered the gradients: also a bypassed
        batch or ``check_grads = False``), so the Adam kernel skips, clears the block and backs the scale off."""
        norm = torch.linalg.vector_norm(self.flat_grads)
        if self.amp_state is not None:
            norm = norm / self.amp_state[0]
            self.found_inf.add_((~torch.isfinite(norm)).to(torch.float32))
        coef = (float(max_norm) / (norm + 1e-6)).clamp(max=1.0)
        self.flat_grads.mul_(coef)
        return norm

    @torch.no_grad()
    def step(self, grad_scale=1.0, lr_schedule=None):
        """``lr_schedule`` = (kl device scalar, kl_scale, kl_threshold, min_lr, max_lr): rl_games' AdaptiveScheduler
        applied to ``self.lr`` AFTER this step used the old value, by the same launch (GPU path only)."""
        b1, b2 = self.betas
        if self._lib is not None:
            kl, kscale, thr, lo, hi = lr_schedule if lr_schedule is not None else (None, 0.0, 0.0, 0.0, 0.0)
            amp = self.amp_state is not None
            if amp and self.check_grads:
                # backward passes whose kernels do not flag overflows themselves: one reduction over the gradient block
                self.found_inf.add_((~torch.isfinite(self.flat_grads)).any().to(torch.float32))
            rc = self._lib.vine_adam_step_amp(self.numel, self.flat_params.data_ptr(), self.flat_grads.data_ptr(),
                                              self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.lr.data_ptr(),
                                              self.step_t.data_ptr(), b1, b2, self.eps, self.weight_decay,
                                              float(grad_scale),
                                              self.shadow.data_ptr() if self.shadow is not None else None,
                                              kl.data_ptr() if kl is not None else None, float(kscale), float(thr),
                                              float(lo), float(hi), self.amp_state.data_ptr() if amp else None,
                                              self.found_inf.data_ptr() if amp else None,
                                              torch.cuda.current_stream(self.flat_params.device).cuda_stream)
            if rc != 0:
                raise RuntimeError("vine_adam_step_amp failed with status %d" % rc)
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
        sd = {"state": state, "param_groups": [group]}
        if self.amp_state is not None:      # torch.amp.GradScaler.state_dict() keys
            sd["scaler"] = {"scale": float(self.amp_state[0]), "growth_factor": 2.0, "backoff_factor": 0.5,
                            "growth_interval": int(self.amp_state[2]), "_growth_tracker": int(self.amp_state[1])}
        return sd

    def load_state_dict(self, sd):
        sc = sd.get("scaler")
        if sc is not None and self.amp_state is not None:
            self.amp_state[0], self.amp_state[1], self.amp_state[2] = float(sc["scale"]), float(sc["_growth_tracker"]), float(sc["growth_interval"])
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_t.fill_(float(st["step"]))
