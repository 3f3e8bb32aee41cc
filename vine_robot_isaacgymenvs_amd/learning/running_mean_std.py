"""Running mean/variance normaliser (rl_games ``RunningMeanStd`` semantics: float64 statistics updated by
the parallel-variance formula in training mode, output clamped to +-5; ``normalize_input``/``normalize_value``
of cfg/train/Vine5LinkMovingBasePPO.yaml:54-55)."""
import torch
import torch.nn as nn


class RunningMeanStd(nn.Module):
    def __init__(self, insize, epsilon=1e-05):
        super().__init__()
        self.insize = insize if isinstance(insize, (tuple, list)) else (insize,)
        self.epsilon = epsilon
        self.register_buffer("running_mean", torch.zeros(self.insize, dtype=torch.float64))
        self.register_buffer("running_var", torch.ones(self.insize, dtype=torch.float64))
        self.register_buffer("count", torch.ones((), dtype=torch.float64))

    @torch.no_grad()
    def update(self, x):
        """Chan et al. merge of the batch moments into the running moments."""
        x = x.reshape(-1, *self.insize)
        batch_mean = x.mean(0).double()
        batch_var = x.var(0, unbiased=True).double() if x.shape[0] > 1 else torch.zeros_like(batch_mean)
        batch_count = float(x.shape[0])
        delta = batch_mean - self.running_mean
        tot = self.count + batch_count
        new_mean = self.running_mean + delta * batch_count / tot
        m2 = self.running_var * self.count + batch_var * batch_count + delta.pow(2) * self.count * batch_count / tot
        self.running_mean.copy_(new_mean)
        self.running_var.copy_(m2 / tot)
        self.count.copy_(tot)

    def _use_kernels(self, x):
        """MI355X path: the update is two hand-written launches and the normalisation one (csrc/ppo_kernels.hip),
        with a fixed summation order and no memsets, so both can live inside a captured hipGraph; same arithmetic as
        the torch composition below (which stays the CPU path and the reference of the numerics test)."""
        return (x.is_cuda and x.dtype == torch.float32 and len(self.insize) == 1 and self.insize[0] <= 64
                and x.dim() == 2 and x.shape[1] == self.insize[0] and not torch.is_autocast_enabled())

    def update_kernels(self, x):
        """Statistics update only (training mode; nothing in eval mode): for callers that normalise elsewhere."""
        if not self.training:
            return
        from .. import native
        from ..abi import RMS_BLOCKS
        lib = native.load()
        x = x.contiguous()
        n, F = x.shape
        scratch = torch.empty(RMS_BLOCKS * 2 * F, device=x.device, dtype=torch.float64)
        native.check(lib.vine_rms_update(n, F, x.data_ptr(), self.running_mean.data_ptr(), self.running_var.data_ptr(),
                                         self.count.data_ptr(), scratch.data_ptr(),
                                         torch.cuda.current_stream(x.device).cuda_stream), lib)

    def update_kernels_multi(self, x, k):
        """``k`` consecutive training-mode updates, one per slice of ``x`` [k n, F], in three launches
        (``vine_rms_update_multi``: bit-identical to k calls of ``update_kernels`` on the slices).  Returns
        (mean [k, F], var [k, F]): the running moments as they stand after slice 0, 1, ...; None in eval mode."""
        if not self.training:
            return None
        from .. import native
        from ..abi import RMS_BLOCKS
        lib = native.load()
        x = x.contiguous()
        rows, F = x.shape
        assert rows % k == 0
        scratch = torch.empty(k * (RMS_BLOCKS + 1) * 2 * F, device=x.device, dtype=torch.float64)
        snap = torch.empty((2, k, F), device=x.device, dtype=torch.float64)
        native.check(lib.vine_rms_update_multi(k, rows // k, F, x.data_ptr(), self.running_mean.data_ptr(),
                                               self.running_var.data_ptr(), self.count.data_ptr(), scratch.data_ptr(),
                                               snap[0].data_ptr(), snap[1].data_ptr(),
                                               torch.cuda.current_stream(x.device).cuda_stream), lib)
        return snap[0], snap[1]

    def _forward_kernels(self, x):
        from .. import native
        lib = native.load()
        x = x.contiguous()
        n, F = x.shape
        st = torch.cuda.current_stream(x.device).cuda_stream
        self.update_kernels(x)
        y = torch.empty_like(x)
        native.check(lib.vine_normalize_obs(n, F, x.data_ptr(), self.running_mean.data_ptr(), self.running_var.data_ptr(),
                                            float(self.epsilon), 5.0, y.data_ptr(), F, 0, st), lib)
        return y

    def forward(self, x, unnorm=False):
        if not unnorm and self._use_kernels(x):
            return self._forward_kernels(x)
        if self.training and not unnorm:
            self.update(x)
        mean = self.running_mean.float()
        std = torch.sqrt(self.running_var.float() + self.epsilon)
        if unnorm:
            y = torch.clamp(x, min=-5.0, max=5.0)
            return std * y + mean
        y = (x - mean) / std
        return torch.clamp(y, min=-5.0, max=5.0)
