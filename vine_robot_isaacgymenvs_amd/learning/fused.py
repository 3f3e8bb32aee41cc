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
                        backward in one kernel;
  * ``trunk``         : the whole network (MLP -> LSTM -> LayerNorm -> heads) as ONE autograd node with a hand-written
                        backward, in fp32 or with bfloat16 GEMM operands (``mixed_precision: True``); what the update
                        actually runs.  The smaller ops above remain as building blocks, for networks the trunk does
                        not cover, and as stepping stones of the numerics tests;
  * ``column_sums``   : deterministic reductions over rows that are safe inside a captured hipGraph.
On a CPU tensor every op falls back to the plain PyTorch composition it replaces (used by the gloo tests and as the
fp32 reference of the numerics tests); on a GPU tensor the HIP kernels are mandatory.
"""
import math

import torch
import torch.nn.functional as F


_runtime_ready = set()


def _lib():
    from .. import native
    lib = native.load()
    if torch.cuda.is_available():
        dev = torch.cuda.current_device()
        if dev not in _runtime_ready:       # per-device bookkeeping of the library, outside any stream capture
            _runtime_ready.add(dev)
            lib.vine_ppo_runtime_init()
    return lib


def _stream(t):
    # the kernels of csrc/ppo_kernels.hip launch on the CURRENT device (stream handle 0 = its null stream): the tensor
    # must live there, else the launch would run on another GPU against this one's pointers
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError("fused op on %s while the current device is cuda:%d: call torch.cuda.set_device first "
                           "(A2CAgent and train.launch do)" % (t.device, torch.cuda.current_device()))
    return torch.cuda.current_stream(t.device).cuda_stream


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with status %d" % (what, rc))


_LP16 = (torch.float16, torch.bfloat16)
_lp = None


def lp_dtype():
    """torch dtype of the library's 16-bit operand format (``vine_lp16_format()``): float16 in the default build -- the
    dtype the reference's ``mixed_precision: True`` autocasts to -- or bfloat16 (``-DVINE_LP_BF16``)."""
    global _lp
    if _lp is None:
        try:
            _lp = torch.float16 if _lib().vine_lp16_format() == b"fp16" else torch.bfloat16
        except (RuntimeError, OSError):      # library not built (CPU-only use of the fallback compositions)
            return torch.float16
    return _lp


# torch.amp.GradScaler restated on the device (fp16 operands need it, bf16 do not): `amp` = (loss scale, overflow flag),
# two 1-element device tensors owned by the optimiser (FlatAdam.enable_loss_scaling).  It travels explicitly -- in the
# loss pack of the trunk, as an argument of ppo_loss_fused -- never as module state: two agents in one process, or a test
# calling a kernel directly after an agent existed, must not inherit each other's scale.
def _amp_ptrs(amp):
    if amp is None:
        return None, None
    return amp[0].data_ptr(), amp[1].data_ptr()


def _mm(a, b):
    """a @ b with an fp32 result: plain fp32 GEMM, or 16-bit operands accumulated and written in fp32."""
    if a.dtype in _LP16:
        return torch.mm(a, b, out_dtype=torch.float32)
    return a.mm(b)


# --------------------------------------------------------------------------- split-K weight gradient
import os as _os
SPLITK_ROWS = int(_os.environ.get("VINE_SPLITK_ROWS", "1024"))      # rows of the reduction per slice (experiments)
def splitk_tn(dy, x, out=None, batch=None):
    """``dy^T @ x`` for tall-skinny operands: dy [K, M], x [K, N] -> [M, N] (written into ``out`` when given).
    ``batch``: defer the sum over the slices to a ``ColumnSumBatch`` (the result exists after its ``flush``)."""
    K = dy.shape[0]
    s = 1
    while K % (s * 2) == 0 and K // (s * 2) >= SPLITK_ROWS and s < 64:
        s *= 2
    if s == 1 or not dy.is_cuda:
        if batch is not None:
            batch.bypassed = True       # this gradient does not pass the overflow-checked column-sum launch
        if dy.dtype in _LP16:
            r = torch.mm(dy.t(), x, out_dtype=torch.float32)
            return out.copy_(r) if out is not None else r
        return torch.mm(dy.t(), x, out=out) if out is not None else dy.t().mm(x)
    # unflatten: also valid for operands whose rows are padded (a column block of a wider buffer)
    a, b = dy.unflatten(0, (s, K // s)).transpose(1, 2), x.unflatten(0, (s, K // s))
    if dy.dtype not in (torch.float32,) + _LP16 or dy.dtype != x.dtype:
        raise TypeError("splitk_tn: operands must both be fp32 or both 16-bit (got %s, %s)" % (dy.dtype, x.dtype))
    part = torch.bmm(a, b, out_dtype=torch.float32) if dy.dtype in _LP16 else torch.bmm(a, b)
    return column_sums(part, out if out is not None else torch.empty(part.shape[1:], device=part.device), batch=batch)


def _splitk_parts(dy, x):
    """The slices of ``splitk_tn`` before their sum: (s, part [s, M, N]) or (1, dy^T x [M, N])."""
    K = dy.shape[0]
    s = 1
    while K % (s * 2) == 0 and K // (s * 2) >= SPLITK_ROWS and s < 64:
        s *= 2
    if s == 1 or not dy.is_cuda:
        return 1, dy.t().mm(x)
    return s, torch.bmm(dy.unflatten(0, (s, K // s)).transpose(1, 2), x.unflatten(0, (s, K // s)))


WGRAD_WGS = int(_os.environ.get("VINE_WGRAD_WGS", "256"))      # workgroups a weight-gradient launch aims for
# largest M * Np the matrix-core weight-gradient kernel (vine_weight_grad_mfma) takes.  0 = off, the default: measured
# inside the whole update (MI355X, n = 32768) the library split-K product + column sums wins at every weight of the
# default network -- update 15.3 ms without the kernel, 16.0 ms with it for the four small weights, 17.8 ms for all
# five (w_hh [1024, 256] alone: 93 us against 38 us).  With one workgroup per CU and 32-row stages the kernel is bound
# by memory latency, not by the matrix cores; it stays as a tested building block (transposed LDS reads) behind this
# knob.
WGRAD_MAX_OUT = int(_os.environ.get("VINE_WGRAD_MAX_OUT", "0"))


def _wgrad_plan(dy, x):
    """(Np, slices) when ``vine_weight_grad_mfma`` covers dy^T x for these operands, else None.  Np: x's columns
    rounded up to a tile width of the kernel; the extra columns must lie inside x's own rows (a column block of a
    wider, padded buffer) -- they are read, their products are not stored."""
    if not (dy.is_cuda and dy.dtype == lp_dtype() and x.dtype == lp_dtype() and dy.dim() == 2 and x.dim() == 2):
        return None
    n, M = dy.shape
    N = x.shape[1]
    if x.shape[0] != n or M % 64 or n % 256 or dy.stride(1) != 1 or x.stride(1) != 1:
        return None
    Np = 32 if N <= 32 else (96 if 64 < N <= 96 else ((N + 127) // 128 * 128 if N > 96 else 0))
    if not Np or dy.stride(0) % 8 or x.stride(0) % 8 or dy.data_ptr() % 16 or x.data_ptr() % 16 or dy.stride(0) < M:
        return None
    if Np != N and (x.storage_offset() % x.stride(0)) + Np > x.stride(0):
        return None
    if x.stride(0) < Np or M * Np > WGRAD_MAX_OUT:
        return None
    wgs = (M // (128 if M % 128 == 0 else 64)) * (Np // (Np if Np <= 96 else 128))
    s = 1
    while wgs * s < WGRAD_WGS and n // (2 * s) >= 256 and n % (2 * s * 32) == 0:
        s *= 2
    return Np, s


def weight_grad(dy, x, out=None, batch=None):
    """``dy^T @ x`` -> [M, N] fp32 (into ``out`` when given): the matrix-core kernel with transposed LDS reads when the
    operands allow it (bf16, the mixed-precision update), else ``splitk_tn``."""
    if dy.dtype == lp_dtype() and x.dtype == lp_dtype() and dy.is_cuda:
        o = out if out is not None else torch.empty((dy.shape[1], x.shape[1]), device=dy.device, dtype=torch.float32)
        if o.is_contiguous() and weight_grad_cat(dy, None, x, None, o, batch=batch):
            return o
    plan = _wgrad_plan(dy, x)
    if plan is None:
        return splitk_tn(dy, x, out=out, batch=batch)
    Np, s = plan
    n, M = dy.shape
    N = x.shape[1]
    if out is None:
        out = torch.empty((M, N), device=dy.device, dtype=torch.float32)
    direct = s == 1 and out.is_contiguous()
    part = out.view(1, M, N) if direct else torch.empty((s, M, N), device=dy.device, dtype=torch.float32)
    _check(_lib().vine_weight_grad_mfma(n, M, Np, N, dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), s,
                                        part.data_ptr(), _stream(dy)), "vine_weight_grad_mfma")
    if direct:
        if batch is not None:
            batch.bypassed = True
        return out
    return column_sums(part, out, batch=batch)


HEADS_LOSS = _os.environ.get("VINE_HEADS_LOSS", "1") != "0"    # LayerNorm + heads + loss + backward in one launch (A/B knob)
ROLLOUT_F32_MFMA = _os.environ.get("VINE_ROLLOUT_F32_MFMA", "1") != "0"   # fp32 matrix-core rollout kernels (A/B knob)
# fp32 rollout inference with the products formed from bf16 pieces on the bf16 matrix cores (vine_lstm_step_f32_split,
# vine_mlp3_elu_f32_split): 6 = without the three piece pairs below 2^-24 of a product (the default), 9 = all nine pairs,
# 0 = the native fp32 matrix-core kernels (vine_lstm_step_f32, vine_mlp3_elu_f32).
# Round 5, measured ("worse on any case: keep 9"; scripts/ubench/split_terms_error.py,
# profiles/r05/split_terms_error.txt; asserted by test_*_f32_split_against_float64_torch): with ONE accumulator per tile the
# 6- and 9-pair forms have the same error against float64 to four digits (the accumulation's 66 / 99 roundings dominate, not
# the dropped pairs), 10-25 % below the native fp32 instruction's in rms but up to 1.19x (LSTM test inputs: 1.63x) ABOVE it
# in max.  With TWO accumulators (ROLLOUT_F32_DUAL below) both forms are at most 0.60x (LSTM) / 0.45x (MLP) of the native
# instruction's error on every input, max and rms, and 6 against 9 pairs differ by < 1 % of that error on rms and either
# way on max: 6 pairs + two accumulators is the default (rollout 2.20 -> 1.93 ms), the 9-pair iteration rides in the bench
# line as `extra_split9_rollout`.
ROLLOUT_F32_SPLIT = int(_os.environ.get("VINE_ROLLOUT_F32_SPLIT", "6"))
# the one-gate-per-wave form of the LSTM kernel (four waves share 64 rows, operand pieces exchanged through LDS; bit 16 of
# `terms`): "auto" = below 16384 rows, where it is faster (23.1 against 24.9 us at 4096 rows, 14.7 against 22.0 at 2048;
# level at 16384: profiles/r04/rollout_kernels_round4.txt); "1" / "0" = always / never
ROLLOUT_F32_NSPLIT = _os.environ.get("VINE_LSTM_STEP_NSPLIT", "auto")
# two accumulators per tile (bit 17 of the LSTM kernel's `terms`, bit 16 of the MLP kernel's): the hi x hi pair apart from
# the smaller pairs -- the smaller pairs' sum is rounded at 2^-8 of the result's magnitude, which leaves one rounding per
# k-block at full magnitude (11 in the LSTM step instead of 66-99).  On (the default) it makes the one-gate-per-wave kernel
# the LSTM kernel at every size (the only one that has the second accumulator).
ROLLOUT_F32_DUAL = _os.environ.get("VINE_ROLLOUT_F32_DUAL", _os.environ.get("VINE_LSTM_STEP_DUAL", "1")) != "0"


def rollout_f32_nsplit(rows):
    """Bits 16-17 of vine_lstm_step_f32_split's `terms` for a batch of `rows`: 1 = one gate per wave, 3 = with two accumulators."""
    if ROLLOUT_F32_NSPLIT == "0":
        return 0
    if ROLLOUT_F32_DUAL:
        return 3
    return 1 if (ROLLOUT_F32_NSPLIT == "1" or rows < 16384) else 0


# the fp32 rollout MLP with the same piece products (vine_mlp3_elu_f32_split: four waves share the rows and split the units);
# 0 = the native fp32 matrix-core kernel (vine_mlp3_elu_f32).  VINE_MLP3_F32_SPLIT_RT: row tiles per workgroup (0: from N)
MLP3_F32_SPLIT = _os.environ.get("VINE_MLP3_F32_SPLIT", "1") != "0"
MLP3_F32_SPLIT_RT = int(_os.environ.get("VINE_MLP3_F32_SPLIT_RT", "0"))
MLP3 = _os.environ.get("VINE_MLP3", "1") != "0"                # the three MLP layers in one launch (A/B knob)
MLP3_PREP = _os.environ.get("VINE_MLP3_PREP", "1") != "0"      # the step's operand preparation rides in that launch (A/B knob)
COPY_SCATTER = _os.environ.get("VINE_COPY_SCATTER", "1") != "0"  # coalesced-load / scattered-store forms of tiles and transposes
RIDES = [0]      # forward passes whose operand preparation rode in the MLP launch (tests look at it)
# LSTM backward + MLP backward as two phases of one launch (same rows per workgroup); 0: two launches (A/B knob)
BWD_PHASES = _os.environ.get("VINE_BWD_PHASES", "1") != "0"
# LSTM forward + LayerNorm / heads / loss + LSTM backward + MLP backward as four phases of one launch; 0: separate launches
TRUNK_PHASES = _os.environ.get("VINE_TRUNK_PHASES", "1") != "0"
PHASE_LAUNCHES = [0]      # vine_trunk_phases launches (tests look at it)
WGRAD_SIDE = _os.environ.get("VINE_WGRAD_SIDE", "0") == "1"      # MLP weight gradients on a parallel branch (experiment)
_SIDE_STREAMS = {}


def _side_stream(dev):
    key = (dev.type, dev.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]
WGRAD_CAT = _os.environ.get("VINE_WGRAD_CAT", "1") != "0"      # second-generation weight-gradient kernel (A/B knob)
WGRAD_CAT_WGS = int(_os.environ.get("VINE_WGRAD_CAT_WGS", "512"))   # workgroups a launch aims for (2 per CU)
WGRAD_WIDE_BM = int(_os.environ.get("VINE_WGRAD_WIDE_BM", "128"))   # 128 | 64 rows of dy^T per workgroup tile (A/B knob)
WGRAD_GROUP_WGS = int(_os.environ.get("VINE_WGRAD_GROUP_WGS", "256"))   # per problem of a grouped launch
WGRAD_CAT_WIDE = _os.environ.get("VINE_WGRAD_CAT_WIDE", "1") != "0"  # 128 x 352 tiles for the LSTM's [x | h] (A/B knob)


def _cat_operand(x, n):
    """(columns read, columns stored) of a bf16 operand of ``vine_weight_grad_cat_mfma``: the columns are rounded up to
    a multiple of 16 that must lie inside the rows of the buffer x is a column block of (read, never stored)."""
    if not (x.is_cuda and x.dtype == lp_dtype() and x.dim() == 2 and x.shape[0] == n and x.stride(1) == 1
            and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0):
        return None
    N = x.shape[1]
    Np = (N + 15) // 16 * 16
    if Np != N and (x.storage_offset() % x.stride(0)) + Np > x.stride(0):
        return None
    return Np, N


def _wgrad_cat_plan(dy, x1, x2, out1, out2, allow_wide=True, wgs=None):
    """Arguments of ``vine_weight_grad_cat_mfma`` for ``dy^T @ [x1 | x2]`` -> (n, M, N1p, Nv1, N2p, Nv2, NT, S) or None."""
    if not (WGRAD_CAT and dy.is_cuda and dy.dtype == lp_dtype() and dy.dim() == 2 and dy.stride(1) == 1
            and dy.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and out2.is_contiguous()
            and (out1 is None or out1.is_contiguous())):
        return None
    n, M = dy.shape
    o2 = _cat_operand(x2, n)
    o1 = _cat_operand(x1, n) if x1 is not None else (0, 0)
    if o1 is None or o2 is None or M % 64:
        return None
    Nt = o1[0] + o2[0]
    NT = 11 if Nt % 176 == 0 else (8 if Nt % 128 == 0 else (2 if Nt == 32 else 0))
    if not NT:
        return None
    wide = allow_wide and WGRAD_CAT_WIDE and o1[0] == 96 and o2[0] == 256 and M % 128 == 0
    if wide:        # one 128 x 352 (or 64 x 352) tile per workgroup, one workgroup per CU
        NT, tiles, target = (22, M // 128, 256) if WGRAD_WIDE_BM == 128 else (21, M // 64, 256)
    else:
        tiles, target = (M // 64) * (Nt // (16 * NT)), (wgs or WGRAD_CAT_WGS)
    S = 8
    while tiles * S < target and n % (128 * S) == 0 and n // (64 * S) >= 8:
        S *= 2
    if n % (64 * S if wide else 32 * S):
        return None
    return n, M, o1[0], o1[1], o2[0], o2[1], NT, S


def h_once_ok(T):
    """``vine_weight_grad_cat_seq_mfma`` forms the masked, shifted recurrent operand itself when T divides 32."""
    return H_ONCE and T >= 1 and 32 % T == 0


def masked_previous_hidden(h_all, dones, T):
    """[n, H] operand of the recurrent weight gradient from the "h once" tensor [B * (T + 1), H] (the slow way, in torch:
    callers whose shapes ``vine_weight_grad_cat_seq_mfma`` does not cover): row seq * T + t = (1 - done) * h_{t-1}."""
    H = h_all.shape[1]
    B = h_all.shape[0] // (T + 1)
    prev = h_all.view(B, T + 1, H)[:, :T].reshape(B * T, H)
    if dones is None or dones.numel() == 0:
        return prev.contiguous()
    return (prev * (1 - dones.view(B * T, 1).to(prev.dtype))).contiguous()


def weight_grad_cat(dy, x1, x2, out1, out2, batch=None, seq=None):
    """``dy^T @ [x1 | x2]`` in ONE pass over ``dy`` on the matrix cores (``vine_weight_grad_cat_mfma``); ``x1`` / ``out1``
    may be None (a plain ``dy^T @ x2``).  dy [n, M], x1 [n, N1], x2 [n, N2] bf16; out1 [M, N1], out2 [M, N2] fp32 receive the
    sums over the row slices (through ``batch`` when given).  Returns False when the shapes are not covered.
    ``seq`` = (dones [n] uint8, T): x2 is the LSTM's one copy of its hidden states [n / T * (T + 1), N2] and the operand
    row k is (1 - done[k]) * h_{t-1}, formed inside the kernel (``vine_weight_grad_cat_seq_mfma``, wide tile only)."""
    plan = _wgrad_cat_plan(dy, x1, x2 if seq is None else x2[:dy.shape[0]], out1, out2)
    if plan is None:
        return False
    n, M, N1p, Nv1, N2p, Nv2, NT, S = plan
    if seq is not None and (NT not in (21, 22) or x1 is None or not h_once_ok(seq[1]) or seq[0].data_ptr() % 16
                            or n // S // 32 > 1024 or x2.shape[0] != n // seq[1] * (seq[1] + 1) or not x2.is_contiguous()):
        return False
    part2 = torch.empty((S, M, Nv2), device=dy.device, dtype=torch.float32)
    part1 = torch.empty((S, M, Nv1), device=dy.device, dtype=torch.float32) if x1 is not None else None
    if seq is not None:
        dones, T = seq
        assert dones.numel() == n and dones.dtype == torch.uint8
        _check(_lib().vine_weight_grad_cat_seq_mfma(n, M, dy.data_ptr(), dy.stride(0), x1.data_ptr(), x1.stride(0), Nv1,
                                                    x2.data_ptr(), x2.stride(0), dones.data_ptr(), T, Nv2, NT, S,
                                                    part1.data_ptr(), part2.data_ptr(), _stream(dy)),
               "vine_weight_grad_cat_seq_mfma")
    else:
        _check(_lib().vine_weight_grad_cat_mfma(n, M, dy.data_ptr(), dy.stride(0), x1.data_ptr() if x1 is not None else None,
                                                x1.stride(0) if x1 is not None else 0, N1p, Nv1, x2.data_ptr(), x2.stride(0),
                                                N2p, Nv2, NT, S, part1.data_ptr() if part1 is not None else None,
                                                part2.data_ptr(), _stream(dy)), "vine_weight_grad_cat_mfma")
    if part1 is not None:
        column_sums(part1, out1, batch=batch)
    column_sums(part2, out2, batch=batch)
    return True


class WeightGradGroup:
    """Several small-tile weight-gradient products in ONE launch (``vine_weight_grad_group``): ``add`` plans a product
    and registers its slice sums with the column-sum batch, ``flush`` launches them all.  The operands must stay alive
    (and unchanged) until the flush; the group keeps references."""
    MAX = 6

    def __init__(self):
        self.jobs = []

    def add(self, dy, x, out, batch):
        if len(self.jobs) >= self.MAX or batch is None:
            return False
        # (workgroups PER PROBLEM: the problems of a group share the launch, so each needs fewer row slices -- and leaves
        # fewer partial sums for the column-sum kernel -- than a launch of its own: 256 measured best, update -0.2 ms)
        plan = _wgrad_cat_plan(dy, None, x, None, out, allow_wide=False, wgs=WGRAD_GROUP_WGS)
        if plan is None:
            return False
        n, M, _n1p, _nv1, N2p, Nv2, NT, S = plan
        part = torch.empty((S, M, Nv2), device=dy.device, dtype=torch.float32)
        self.jobs.append((n, M, dy, x, N2p, Nv2, NT, S, part))
        column_sums(part, out, batch=batch)
        return True

    def flush(self):
        if not self.jobs:
            return
        import ctypes as C
        k = len(self.jobs)
        I64, VP = C.c_int64 * k, C.c_void_p * k
        col = lambda f: [f(j) for j in self.jobs]
        zeros, nulls = I64(*([0] * k)), VP(*([None] * k))
        _check(_lib().vine_weight_grad_group(
            k, I64(*col(lambda j: j[0])), I64(*col(lambda j: j[1])), VP(*col(lambda j: j[2].data_ptr())),
            I64(*col(lambda j: j[2].stride(0))), nulls, zeros, zeros, zeros, VP(*col(lambda j: j[3].data_ptr())),
            I64(*col(lambda j: j[3].stride(0))), I64(*col(lambda j: j[4])), I64(*col(lambda j: j[5])),
            I64(*col(lambda j: j[6])), I64(*col(lambda j: j[7])), nulls, VP(*col(lambda j: j[8].data_ptr())),
            _stream(self.jobs[0][2])), "vine_weight_grad_group")
        self.jobs = []


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
    # (not under torch.autocast: the hand-written backward passes assume fp32 tensors)
    if (x.is_cuda and torch.is_grad_enabled() and x.shape[0] >= 4096 and x.dtype == torch.float32
            and not torch.is_autocast_enabled()):
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


def _lstm_state_buffers(x, w_hh, h0, c0, dones, T, need_grad, prep=None, copy_c0=True, c_dtype=torch.float32,
                        mask_h0=True, h_once=False):
    """out / c_all / gates / hp of ``_lstm_forward_steps`` with the step-0 slots initialised (c_all[0] = c0,
    hp[:, 0] = masked h0 in the operand dtype): through ``prep`` (a CopyBatch flushed by the caller) or directly.
    ``h_once`` (persistent kernels only): ``out`` [B * (T + 1), H] is the ONE copy of the hidden states, 16-bit and
    unmasked, slot 0 of every sequence = h0, slot t + 1 = h_t (see ``vine_lstm_seq_forward_mfma``); ``hp`` is None."""
    op = w_hh.dtype
    BT, H = x.shape[0], w_hh.shape[1]
    B = BT // T
    dev = x.device
    if h_once:
        assert not mask_h0 and not copy_c0 and c_dtype == op == lp_dtype() and need_grad
        return (torch.empty((B * (T + 1), H), device=dev, dtype=op), torch.empty((T + 1, B, H), device=dev, dtype=c_dtype),
                torch.empty((T, B, 4 * H), device=dev, dtype=op), None)
    out = torch.empty((BT, H), device=dev, dtype=torch.float32)
    # (c_dtype bfloat16: the persistent sequence kernels' low-precision store of the cell states the backward pass re-reads)
    c_all = torch.empty((T + 1, B, H), device=dev, dtype=c_dtype)
    gates = torch.empty((T, B, 4 * H), device=dev, dtype=op) if need_grad else None     # backward-only: operand dtype
    hp = torch.empty((B, T, H), device=dev, dtype=op)
    if prep is not None:
        if copy_c0:      # else: the caller hands c0 itself to the step kernels (``c0_direct``), slot 0 stays unused
            prep.add(CopyBatch.COPY, c_all[0], c0)
        if mask_h0:      # else: the persistent forward kernel forms the masked bf16 state itself (h0 handed to it)
            prep.add(CopyBatch.MASKED, hp[:, 0], h0, dones, aux=T)
    else:
        c_all[0].copy_(c0)
        if dones is not None:
            hp[:, 0].copy_(h0 * (1.0 - dones.view(B, T)[:, 0:1].to(torch.float32)))
        else:
            hp[:, 0].copy_(h0)
    return out, c_all, gates, hp


# saved cell states and the hidden-state gradient of the persistent LSTM kernels as bfloat16 (backward-only data; the
# recurrence, the returned state and every parameter gradient stay fp32); 0: fp32 (A/B knob)
LSTM_LP = _os.environ.get("VINE_LSTM_LP", "1") != "0"
LSTM_SEQ = _os.environ.get("VINE_LSTM_SEQ", "1") != "0"      # 0: one launch per time step (the round-1 kernels), for A/B runs
# the hidden states of the update's LSTM stored ONCE (16-bit, unmasked) instead of fp32 for the LayerNorm + masked 16-bit
# for the recurrent weight gradient; 0: both copies (A/B knob)
H_ONCE = _os.environ.get("VINE_H_ONCE", "1") != "0"
# the loss kernel's serial last step deferred into the column-sum launch that ends the backward pass; 0: inside the loss
# kernel (ticket + last workgroup), as its stand-alone callers get it (A/B knob)
LOSS_DEFER = _os.environ.get("VINE_LOSS_DEFER", "1") != "0"


def lstm_seq_ok(B, H, T, wpad):
    """Shapes the persistent sequence kernels (vine_lstm_seq_forward_mfma / _backward_mfma) cover."""
    return LSTM_SEQ and B % 32 == 0 and H == 256 and 1 <= T <= 8 and wpad in (32, 64, 96, 128)


def _lstm_forward_steps(lib, x, ig, w_hh, bias, h0, c0, dones, T, need_grad, wcat=None, buffers=None, c0_direct=None,
                        wtile=None, c_last=None, h0_direct=None):
    """T LSTM steps from the input projection ``ig`` [B*T, 4H]: per step one recurrent GEMM on the MASKED previous
    hidden state + the fused pointwise kernel, which also emits the masked state for the next step (``hp``).
    ``w_hh`` in bfloat16 selects bf16 GEMM operands (``hp`` is then stored in bfloat16); everything else is fp32.
    ``wcat`` = [w_ih | 0 | w_hh] (bf16, K = padded input width + H) with ``ig`` None: no input projection at all; the
    step kernel multiplies the two operand blocks x_t (rows of ``x``, padded width) and h_{t-1} in one product."""
    op = w_hh.dtype
    BT, H = x.shape[0], w_hh.shape[1]
    B = BT // T
    out, c_all, gates, hp = buffers if buffers is not None else _lstm_state_buffers(x, w_hh, h0, c0, dones, T, need_grad)
    # c_{t-1} of step t: c_all[t], except that step 0 may read the caller's c0 in place (saves the 2 x B x H copy)
    c_prev = [c_all[t] for t in range(T)]
    if c0_direct is not None:
        assert c0_direct.dtype == torch.float32 and c0_direct.is_contiguous() and c0_direct.shape == c_all[0].shape
        c_prev[0] = c0_direct
    st = _stream(x)
    d_ptr = dones.data_ptr() if dones is not None else None
    if wtile is not None:
        # ONE launch for the whole sequence: h_t stays in LDS, c_t in registers, weights stream from the
        # fragment-ordered copy (``wtile``); x = the padded operand buffer whose first K1 columns are the step input
        K1 = wtile.numel() // (4 * H) - H
        h_once = hp is None                   # (buffers made with h_once: out is [B * (T + 1), H], 16-bit)
        assert op == lp_dtype() and lstm_seq_ok(B, H, T, K1) and x.stride(0) >= K1 and (h_once or hp.is_contiguous())
        assert not h_once or (out.shape == (B * (T + 1), H) and h0_direct is not None and c_last is not None)
        _check(lib.vine_lstm_seq_forward_mfma(B, T, H, K1, x.data_ptr(), x.stride(0), None if h_once else hp.data_ptr(), T * H,
                                              wtile.data_ptr(), bias.data_ptr(), c_prev[0].data_ptr(), d_ptr,
                                              out.data_ptr(), c_all.data_ptr(), gates.data_ptr() if need_grad else None,
                                              int(c_all.dtype == lp_dtype()) | (2 if h_once else 0),
                                              c_last.data_ptr() if c_last is not None else None,
                                              h0_direct.data_ptr() if h0_direct is not None else None, st),
               "vine_lstm_seq_forward_mfma")
        return out, c_all, gates, hp
    w_hh_t = w_hh.t()
    # mixed precision: the recurrent GEMM runs inside the step kernel on the matrix cores (vine_lstm_step_mfma)
    mfma = (op == lp_dtype() and B % 64 == 0 and H in (128, 256, 512))
    for t in range(T):
        last = t == T - 1
        if mfma and wcat is not None:
            K1 = wcat.shape[1] - H
            _check(lib.vine_lstm_step_mfma(
                B, H, K1 + H, x.data_ptr() + 2 * (t * x.stride(0)), T * x.stride(0), hp.data_ptr() + 2 * (t * H), T * H,
                K1, wcat.data_ptr(), wcat.stride(0), None, 4 * H, bias.data_ptr(), c_prev[t].data_ptr(),
                (d_ptr + t) if d_ptr is not None else None, T, out.data_ptr() + 4 * (t * H), T * H,
                c_all[t + 1].data_ptr(), gates[t].data_ptr() if need_grad else None,
                None if last else hp.data_ptr() + 2 * ((t + 1) * H),
                (d_ptr + t + 1) if (d_ptr is not None and not last) else None, T, T * H, st), "vine_lstm_step_mfma")
            continue
        if mfma:
            _check(lib.vine_lstm_step_mfma(
                B, H, H, hp.data_ptr() + 2 * (t * H), T * H, None, 0, 0, w_hh.data_ptr(), w_hh.stride(0),
                ig.data_ptr() + 4 * (t * 4 * H), T * 4 * H, bias.data_ptr(), c_prev[t].data_ptr(),
                (d_ptr + t) if d_ptr is not None else None, T, out.data_ptr() + 4 * (t * H), T * H,
                c_all[t + 1].data_ptr(), gates[t].data_ptr() if need_grad else None,
                None if last else hp.data_ptr() + 2 * ((t + 1) * H),
                (d_ptr + t + 1) if (d_ptr is not None and not last) else None, T, T * H, st), "vine_lstm_step_mfma")
            continue
        hg = _mm(hp[:, t], w_hh_t)
        # hg is already built from the masked state: no second masking inside the kernel (done = NULL), except for c
        _check(lib.vine_lstm_cell_forward(
            B, H, ig.data_ptr() + 4 * (t * 4 * H), T * 4 * H, hg.data_ptr(), bias.data_ptr(), c_prev[t].data_ptr(),
            (d_ptr + t) if d_ptr is not None else None, T, out.data_ptr() + 4 * (t * H), T * H,
            c_all[t + 1].data_ptr(), gates[t].data_ptr() if need_grad else None,
            None if last else hp.data_ptr() + hp.element_size() * ((t + 1) * H),
            (d_ptr + t + 1) if (d_ptr is not None and not last) else None, T, int(op == lp_dtype()), 0, st),
            "vine_lstm_cell_forward")
    return out, c_all, gates, hp


def lstm_bwd_mfma_ok(B, H):
    """Shapes vine_lstm_step_backward_mfma covers."""
    return B % 64 == 0 and H in (128, 256)


def _lstm_backward_steps(lib, g_out, w_hh, c_all, gates, dones, T, w_hh_t=None, c0_direct=None, w_hh_tiled=None,
                         c_last=None):
    """Gate gradients dG [B*T, 4H] of the T steps (reverse order) and the per-workgroup bias-gradient partials.
    dG is only ever a GEMM operand: it is stored in ``w_hh``'s dtype (bfloat16 in the mixed-precision update).
    ``w_hh_t`` ([H, 4H] bf16, the transposed recurrent weight): every step is ONE matrix-core kernel that forms the
    recurrent input gradient dG_{t+1} w_hh itself (vine_lstm_step_backward_mfma)."""
    from ..abi import PPO_PARTIAL_BLOCKS
    B, H = c_all.shape[1], c_all.shape[2]
    dev = g_out.device
    g_out = g_out.contiguous()
    dG = torch.empty((B * T, 4 * H), device=dev, dtype=w_hh.dtype)
    dG3 = dG.view(B, T, 4 * H)
    dc = [torch.empty((B, H), device=dev, dtype=torch.float32) for _ in range(2)]
    c_prev = [c_all[t] for t in range(T)]
    if c0_direct is not None:            # the forward pass read c0 in place: c_all[0] was never written
        c_prev[0] = c0_direct
    assert w_hh_tiled is not None or (g_out.dtype == torch.float32 and c_all.dtype == torch.float32)
    if w_hh_tiled is not None:           # ONE launch for all T steps (dG_{t+1} in LDS, dc / c in registers)
        assert dG.dtype == lp_dtype() and lstm_seq_ok(B, H, T, 32) and gates.is_contiguous() and c_all.is_contiguous()
        bias_partial = torch.empty((B // 32, 4 * H), device=dev, dtype=torch.float32)
        _check(lib.vine_lstm_seq_backward_mfma(B, T, H, g_out.data_ptr(), w_hh_tiled.data_ptr(), gates.data_ptr(),
                                               c_all.data_ptr(), c_prev[0].data_ptr(),
                                               dones.data_ptr() if dones is not None else None, dG.data_ptr(),
                                               bias_partial.data_ptr(), int(c_all.dtype == lp_dtype()),
                                               c_last.data_ptr() if c_last is not None else None,
                                               int(g_out.dtype == lp_dtype()), _stream(g_out)),
               "vine_lstm_seq_backward_mfma")
        return dG, bias_partial
    if w_hh_t is not None:
        assert dG.dtype == lp_dtype() and lstm_bwd_mfma_ok(B, H)
        bias_partial = torch.empty((2, B // 64, 4 * H), device=dev, dtype=torch.float32)
        st = _stream(g_out)
        d_ptr = dones.data_ptr() if dones is not None else None
        for t in reversed(range(T)):
            last = t == T - 1
            _check(lib.vine_lstm_step_backward_mfma(
                B, H, g_out.data_ptr() + 4 * (t * H), T * H,
                None if last else dG.data_ptr() + 2 * ((t + 1) * 4 * H), T * 4 * H,
                None if last else w_hh_t.data_ptr(), w_hh_t.stride(0),
                None if last else dc[(t + 1) & 1].data_ptr(),
                (d_ptr + t + 1) if (d_ptr is not None and not last) else None, T, gates[t].data_ptr(),
                c_all[t + 1].data_ptr(), c_prev[t].data_ptr(), (d_ptr + t) if d_ptr is not None else None, T,
                dG.data_ptr() + 2 * (t * 4 * H), T * 4 * H, dc[t & 1].data_ptr(), bias_partial[t & 1].data_ptr(),
                None if last else bias_partial[(t + 1) & 1].data_ptr(), st), "vine_lstm_step_backward_mfma")
        return dG, bias_partial[0]
    use_partial = H <= 1024 and 256 % (H // 4) == 0
    # one [PPO_PARTIAL_BLOCKS, 4H] block per step, chained: step t adds the rows of step t+1, the last one (t = 0)
    # holds the partial sums of the whole sequence
    bias_partial = torch.empty((2, PPO_PARTIAL_BLOCKS, 4 * H), device=dev, dtype=torch.float32) if use_partial else None
    st = _stream(g_out)
    d_ptr = dones.data_ptr() if dones is not None else None
    g_rec = None
    dc_next = None
    for t in reversed(range(T)):
        dn = (d_ptr + t + 1) if (d_ptr is not None and t < T - 1) else None
        _check(lib.vine_lstm_cell_backward(
            B, H, g_out.data_ptr() + 4 * (t * H), T * H, g_rec.data_ptr() if g_rec is not None else None,
            dc_next.data_ptr() if dc_next is not None else None, dn, T, gates[t].data_ptr(),
            c_all[t + 1].data_ptr(), c_prev[t].data_ptr(), (d_ptr + t) if d_ptr is not None else None, T,
            dG.data_ptr() + dG.element_size() * (t * 4 * H), T * 4 * H, dc[t & 1].data_ptr(),
            bias_partial[t & 1].data_ptr() if use_partial else None,
            bias_partial[(t + 1) & 1].data_ptr() if (use_partial and t < T - 1) else None,
            int(dG.dtype == lp_dtype()), st), "vine_lstm_cell_backward")
        dc_next = dc[t & 1]
        if t > 0:
            g_rec = _mm(dG3[:, t], w_hh)
    return dG, (bias_partial[0] if use_partial else None)     # t = 0 ran last


def _sum_rows(partial, full, out=None):
    """Column sums: of the small per-workgroup ``partial`` block when the kernel produced one, else of ``full``."""
    src = partial.view(-1, partial.shape[-1]) if partial is not None else full
    if src.is_cuda and src.dtype == torch.float32:
        return column_sums(src, out)
    return torch.sum(src, 0, out=out) if out is not None else src.sum(0)


def _as_rows(src):
    """[R, ...] fp32 tensor -> 2-D [R, C] view with unit column stride (rows may be padded / a column block)."""
    R = src.shape[0]
    C = src.numel() // R
    if src.dim() == 2:
        return src if src.stride(1) == 1 else src.contiguous()
    if src[0].is_contiguous():
        return src.as_strided((R, C), (src.stride(0), 1))
    return src.reshape(R, C)


class ColumnSumBatch:
    """Collects column-sum jobs and runs them in ONE launch (``vine_column_sums_batched``, 16 jobs per launch): the
    network's backward pass ends with ~13 of them, each far too small to fill the chip."""

    def __init__(self, found_inf=None):
        self.jobs, self.keep = [], []
        self.found_inf = found_inf      # 1-element device tensor set to 1 by a non-finite result (loss-scaled backward)
        self.fin = None                 # abi.LossFinalize: the deferred last step of vine_ln_heads_loss rides in the launch
        self.bypassed = False           # a weight gradient was written straight into its slot (single-slice fallbacks)

    def add(self, src, out, out1=None, n0=0, dup=False):
        flat = _as_rows(src)
        self.keep.append((flat, out, out1))
        self.jobs.append((flat.shape[0], flat.shape[1], flat.data_ptr(), flat.stride(0), out.data_ptr(), int(n0),
                          out1.data_ptr() if out1 is not None else 0, int(bool(dup))))
        return out

    def flush(self, ref):
        import ctypes as C
        lib, st = _lib(), _stream(ref)
        for k in range(0, len(self.jobs), 16):
            js = self.jobs[k:k + 16]
            n = len(js)
            cols = list(zip(*js))
            i64 = lambda v: (C.c_int64 * n)(*v)
            ptr = lambda v: (C.c_void_p * n)(*v)
            fin = self.fin if k + 16 >= len(self.jobs) else None      # (with the last launch)
            _check(lib.vine_column_sums_batched_fin(n, i64(cols[0]), i64(cols[1]), ptr(cols[2]), i64(cols[3]), ptr(cols[4]),
                                                    i64(cols[5]), ptr(cols[6]), (C.c_int32 * n)(*cols[7]),
                                                    self.found_inf.data_ptr() if self.found_inf is not None else None,
                                                    C.byref(fin) if fin is not None else None, st),
                   "vine_column_sums_batched_fin")
        assert self.fin is None or self.jobs, "a deferred loss finalize needs at least one column-sum job to ride with"
        self.jobs, self.keep, self.fin = [], [], None


class CopyBatch:
    """Collects small 2-D element moves (copy / zero / transpose / fp32->bf16 cast / fp32 add) and runs them in ONE
    launch (``vine_copy_batched``): the operand preparation of an optimiser step is a dozen such moves."""
    COPY, ZERO, TRANSPOSE, CAST_BF16, ADD, MASKED, LSTM_TILE_FWD, LSTM_TILE_BWD = 0, 1, 2, 3, 4, 5, 6, 7
    # scatter forms of the last three (round 4): coalesced 8-B loads of 4 consecutive SOURCE elements, stores to their
    # places (the gather forms read 2 bytes per cache line); same bytes in the destination
    LSTM_TILE_FWD_S, LSTM_TILE_BWD_S, TRANSPOSE_S = 8, 9, 10

    def add_lstm_tiles(self, w_ih, w_hh, wpad, fwd_dst, bwd_dst):
        """Fragment-ordered bf16 copies of the LSTM weights for the persistent sequence kernels (csrc/ppo_kernels.hip,
        lstm_tile_weights_kernel): ``fwd_dst`` [4H * (wpad + H)] from [w_ih | 0 | w_hh], ``bwd_dst`` [H * 4H] from
        w_hh^T (either may be None)."""
        H4, width = w_ih.shape
        H = w_hh.shape[1]
        assert w_ih.dtype == w_hh.dtype == lp_dtype() and w_ih.stride(1) == 1 and w_hh.stride(1) == 1 and H == 256
        scatter = (COPY_SCATTER and width % 4 == 0 and wpad % 4 == 0 and w_ih.stride(0) % 4 == 0 and w_hh.stride(0) % 4 == 0
                   and w_ih.data_ptr() % 8 == 0 and w_hh.data_ptr() % 8 == 0)
        if fwd_dst is not None:
            assert fwd_dst.numel() == H4 * (wpad + H) and fwd_dst.is_contiguous() and width <= wpad < 65536
            self.keep.append((fwd_dst, w_ih, w_hh))
            self.jobs.append((self.LSTM_TILE_FWD_S if scatter else self.LSTM_TILE_FWD, 2, w_ih.data_ptr(), w_hh.data_ptr(),
                              fwd_dst.data_ptr(), 1, fwd_dst.numel(), w_ih.stride(0), w_hh.stride(0), width | (wpad << 16)))
        if bwd_dst is not None:
            assert bwd_dst.numel() == H4 * H and bwd_dst.is_contiguous()
            self.keep.append((bwd_dst, w_hh))
            self.jobs.append((self.LSTM_TILE_BWD_S if scatter else self.LSTM_TILE_BWD, 2, w_hh.data_ptr(), 0,
                              bwd_dst.data_ptr(), 1, bwd_dst.numel(), w_hh.stride(0), 0, 0))

    def __init__(self):
        self.jobs, self.keep = [], []

    def add(self, op, dst, src=None, src2=None, aux=0):
        """dst: 2-D view with unit column stride (1-D tensors are taken as one row).  MASKED: src2 = uint8 mask with
        one entry per row, ``aux`` elements apart (None: plain copy/cast of an fp32 source)."""
        d2 = dst if dst.dim() == 2 else dst.view(1, -1)
        s2 = None if src is None else (src if src.dim() == 2 else src.view(1, -1))
        if op == self.MASKED:
            assert tuple(s2.shape) == tuple(d2.shape) and s2.dtype == torch.float32 and s2.stride(1) == 1 and d2.stride(1) == 1
            assert src2 is None or src2.dtype == torch.uint8
            self.keep.append((d2, s2, src2))
            self.jobs.append((op, d2.element_size(), s2.data_ptr(), 0 if src2 is None else src2.data_ptr(), d2.data_ptr(),
                              d2.shape[0], d2.shape[1], s2.stride(0), d2.stride(0), int(aux)))
            return dst
        t2 = None if src2 is None else (src2 if src2.dim() == 2 else src2.view(1, -1))
        assert d2.stride(1) == 1 and (s2 is None or s2.stride(1) == 1) and (t2 is None or t2.stride(1) == 1)
        rows, cols = d2.shape
        if op == self.TRANSPOSE:
            assert tuple(s2.shape) == (cols, rows) and s2.dtype == d2.dtype
            if (COPY_SCATTER and d2.element_size() == 2 and rows % 4 == 0 and s2.stride(0) % 4 == 0 and s2.data_ptr() % 8 == 0
                    and d2.data_ptr() % 8 == 0 and rows * cols < 2 ** 31):
                op = self.TRANSPOSE_S
        elif op in (self.COPY, self.ADD):
            assert tuple(s2.shape) == (rows, cols) and s2.dtype == d2.dtype
        elif op == self.CAST_BF16:
            assert tuple(s2.shape) == (rows, cols) and s2.dtype == torch.float32 and d2.dtype == lp_dtype()
        if op == self.ADD:
            assert t2.stride(0) == s2.stride(0) and d2.dtype == torch.float32
        self.keep.append((d2, s2, t2))
        self.jobs.append((op, d2.element_size(), 0 if s2 is None else s2.data_ptr(), 0 if t2 is None else t2.data_ptr(),
                          d2.data_ptr(), rows, cols, 0 if s2 is None else s2.stride(0), d2.stride(0), 0))
        return dst

    def packed(self):
        """(njobs, the ten ctypes arrays of vine_copy_batched) for a launch that carries these moves as a side job
        (``vine_mlp3_elu_mfma_prep``), or None when they do not fit one launch; the batch stays filled (``keep`` holds the
        tensors) until ``clear``."""
        import ctypes as C
        n = len(self.jobs)
        if n == 0 or n > 24:
            return None
        cols = list(zip(*self.jobs))
        i32 = lambda v: (C.c_int32 * n)(*v)
        i64 = lambda v: (C.c_int64 * n)(*v)
        ptr = lambda v: (C.c_void_p * n)(*v)
        return (n, i32(cols[0]), i32(cols[1]), ptr(cols[2]), ptr(cols[3]), ptr(cols[4]), i64(cols[5]), i64(cols[6]),
                i64(cols[7]), i64(cols[8]), i64(cols[9]))

    def clear(self):
        self.jobs, self.keep = [], []

    def flush(self, ref):
        import ctypes as C
        lib, st = _lib(), _stream(ref)
        for k in range(0, len(self.jobs), 24):
            js = self.jobs[k:k + 24]
            n = len(js)
            cols = list(zip(*js))
            i32 = lambda v: (C.c_int32 * n)(*v)
            i64 = lambda v: (C.c_int64 * n)(*v)
            ptr = lambda v: (C.c_void_p * n)(*v)
            _check(lib.vine_copy_batched(n, i32(cols[0]), i32(cols[1]), ptr(cols[2]), ptr(cols[3]), ptr(cols[4]),
                                         i64(cols[5]), i64(cols[6]), i64(cols[7]), i64(cols[8]), i64(cols[9]), st),
                   "vine_copy_batched")
        self.jobs, self.keep = [], []


def column_sums(src, out=None, out1=None, n0=0, dup=False, batch=None):
    """Sum over dim 0 of a [R, ...] fp32 tensor with the hand-written kernel (deterministic; unlike ATen's
    multi-block reductions it needs no memset-cleared scratch, so it is safe inside a captured hipGraph).
    ``out``/``out1``: see vine_column_sums (split at column n0, or ``dup`` to write both)."""
    if src.dtype != torch.float32 or not src.is_cuda or (out is not None and out.dtype != torch.float32):
        raise TypeError("column_sums: fp32 GPU tensors only (got %s -> %s)" % (src.dtype, None if out is None else out.dtype))
    if out is None:
        out = torch.empty(src.shape[1:], device=src.device, dtype=torch.float32)
    if batch is not None:
        return batch.add(src, out, out1, n0, dup)
    flat = _as_rows(src)
    R, C = flat.shape
    _check(_lib().vine_column_sums(R, C, flat.data_ptr(), flat.stride(0), out.data_ptr(), int(n0),
                                   out1.data_ptr() if out1 is not None else None, int(bool(dup)), _stream(src)),
           "vine_column_sums")
    return out


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
        out, c_all, gates, hp = _lstm_forward_steps(lib, x, ig, w_hh, bias, h0, c0, dones, T, need_grad)
        out3 = out.view(B, T, H)
        ctx.T = T
        ctx.has_dones = dones is not None
        ctx.slots = (_grad_slot(w_ih), _grad_slot(w_hh), _grad_slot(b_ih), _grad_slot(b_hh))
        if need_grad:
            ctx.save_for_backward(x, w_ih, w_hh, hp, out, c_all, gates, dones if dones is not None else x.new_empty(0))
        hT = out3[:, T - 1].contiguous()
        cT = c_all[T].clone()
        ctx.mark_non_differentiable(hT, cT)
        ctx.set_materialize_grads(False)
        return out, hT, cT

    @staticmethod
    def backward(ctx, g_out, _g_h, _g_c):
        lib = _lib()
        x, w_ih, w_hh, hp, out, c_all, gates, dones = ctx.saved_tensors
        T = ctx.T
        BT, H = out.shape
        dG, bias_partial = _lstm_backward_steps(lib, g_out, w_hh, c_all, gates, dones if ctx.has_dones else None, T)
        gx = dG.mm(w_ih) if ctx.needs_input_grad[0] else None
        s_ih, s_hh, s_bi, s_bh = ctx.slots
        g_ih = splitk_tn(dG, x, out=s_ih)
        g_hh = splitk_tn(dG, hp.view(BT, H), out=s_hh)
        gb = _sum_rows(bias_partial, dG, out=s_bi)
        if s_bi is not None and s_bh is not None:
            s_bh.copy_(s_bi)
            gbi = gbh = None
        else:
            gbi = gbh = gb
        return (gx, None if s_ih is not None else g_ih, None if s_hh is not None else g_hh, gbi, gbh,
                None, None, None, None)


def lstm_sequence(x, w_ih, w_hh, b_ih, b_hh, h0, c0, dones, T):
    """x [B*T, F] (row = seq*T + t), h0/c0 [B, H], dones uint8 [B*T] or None -> (out [B*T, H], hT, cT)."""
    if x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
        if dones is not None:
            dones = dones.to(torch.uint8).contiguous()
        return _LSTMSeq.apply(x, w_ih, w_hh, b_ih, b_hh, h0, c0, dones, T)
    return _lstm_reference(x, w_ih, w_hh, b_ih, b_hh, h0, c0, dones, T)


# --------------------------------------------------------------------------- whole actor-critic trunk
def _heads_loss_route(n, H, NH, loss_pack, head_bias_external, lib):
    """Does the trunk run LayerNorm + heads + loss + their backward as the one ``vine_ln_heads_loss`` launch?"""
    rows = lib.vine_ln_heads_loss_rows()
    return (H == 256 and 2 <= NH <= 5 and loss_pack is not None and HEADS_LOSS and n % rows == 0 and n // rows <= 1024
            and bool(head_bias_external))


class _Trunk(torch.autograd.Function):
    """The training forward/backward of the Vine5LinkMovingBasePPO network as ONE autograd node:
    obs -> [Linear+ELU]*L -> concat obs -> LSTM(T steps) -> LayerNorm -> [mu | value] heads (one GEMM).
    Every GEMM stays in hipBLASLt; everything between them is a hand-written kernel of csrc/ppo_kernels.hip, and the
    backward is written out by hand so that
      * weight gradients are split-K GEMMs written straight into the optimiser's flat gradient block,
      * bias / LayerNorm-parameter gradients come from per-workgroup partial sums the pointwise backward kernels
        emit anyway (no extra pass over the activations),
      * only the MLP columns of the LSTM input gradient are computed (the observation columns need none),
      * both heads share one forward GEMM, one input-gradient GEMM and one weight-gradient GEMM.
    Argument order: obs_n, h0, c0, dones, T, concat, n_mlp, op_weights (None, or the bfloat16 copies of
    W_1..W_L, w_ih, w_hh for the mixed-precision update), then the parameters
    (W_1, b_1, ..., W_L, b_L, w_ih, w_hh, b_ih, b_hh, ln_gamma, ln_beta, ln_eps, mu_w, mu_b, v_w, v_b)."""

    @staticmethod
    def forward(ctx, obs_n, h0, c0, dones, T, concat, n_mlp, op_weights, head_bias_external, norm, loss_pack, *params):
        lib = _lib()
        mlp = [(params[2 * i], params[2 * i + 1]) for i in range(n_mlp)]
        w_ih, w_hh, b_ih, b_hh, ln_g, ln_b, ln_eps, mu_w, mu_b, v_w, v_b = params[2 * n_mlp:]
        # GEMM operands: the fp32 parameters themselves, or their bfloat16 shadows (W_1..W_L, w_ih, w_hh)
        mixed = op_weights is not None
        op = lp_dtype() if mixed else torch.float32
        Wop = list(op_weights[:n_mlp]) if mixed else [W for W, _ in mlp]
        w_ih_op, w_hh_op = (op_weights[n_mlp], op_weights[n_mlp + 1]) if mixed else (w_ih, w_hh)
        n, F_in = obs_n.shape
        dev = obs_n.device
        st = _stream(obs_n)
        U = mlp[-1][0].shape[0]
        H = w_hh.shape[1]
        B = n // T
        width = U + (F_in if concat else 0)
        # rows padded to 64 B so that every row (and the column block the ELU kernels address) is 16-B aligned
        wpad = (width + 15) // 16 * 16
        # layer 1 on the matrix cores: its operand is the observation block of the LSTM operand buffer plus the zero
        # pad columns behind it (K = 32); all three layers in one launch when the shapes are the default network's
        l1_mfma = mixed and concat and wpad - U == 32 and linear_elu_mfma_ok(n, Wop[0].shape[0], 32)
        mlp3 = (mixed and MLP3 and l1_mfma and n_mlp == 3 and n % 64 == 0 and U == 64
                and tuple(W.shape[0] for W in Wop) == (256, 128, 64) and Wop[1].shape[1] == 256 and Wop[2].shape[1] == 128)
        # ``norm`` = (running mean, running var, eps): obs_n holds RAW observations and the one-launch MLP normalises
        # them itself; every other route normalises here first (vine_normalize_obs, same arithmetic)
        raw = None
        if norm is not None:
            obs_c = obs_n.contiguous()
            if mlp3 and obs_c.dtype == torch.float32 and F_in <= 32:
                raw = obs_c
            else:
                y = torch.empty_like(obs_c)
                _check(lib.vine_normalize_obs(n, F_in, obs_c.data_ptr(), norm[0].data_ptr(), norm[1].data_ptr(),
                                              float(norm[2]), 5.0, y.data_ptr(), F_in, 0, st), "vine_normalize_obs")
                obs_n = y
        xfull = torch.empty((n, wpad), device=dev, dtype=op)
        xcat = xfull[:, :width]
        A_ = mu_w.shape[0]
        w_heads = torch.empty((A_ + v_w.shape[0], H), device=dev, dtype=torch.float32)
        b_heads = torch.empty(A_ + v_w.shape[0], device=dev, dtype=torch.float32)
        bias = torch.empty(4 * H, device=dev, dtype=torch.float32)
        seq = mixed and lstm_seq_ok(B, H, T, wpad)      # persistent sequence kernels (one launch per direction)
        no_proj = seq or (mixed and B % 64 == 0 and H == 256 and wpad % 32 == 0 and wpad <= 128)
        wcat, wts, w_hh_t = None, [None] * n_mlp, None
        wtile = w_hh_tiled = None
        if mixed:
            # every operand derived from the parameters or the observations, in ONE launch: bf16 cast of the
            # observations (layer-1 operand and the LSTM operand's obs block), zero pad columns, [w_ih | 0 | w_hh],
            # transposed MLP weights for the backward kernels, merged head weights, b_ih + b_hh
            obs_c = obs_n.contiguous()
            prep = CopyBatch()
            if raw is None:        # (else the MLP kernel writes the normalised observation block and its zero pad)
                if concat:
                    prep.add(CopyBatch.CAST_BF16, xfull[:, U:width], obs_c)
                if wpad > width:
                    prep.add(CopyBatch.ZERO, xfull[:, width:])
            # the one-launch MLP opens the forward pass: the operands below are derived from the parameters alone and first
            # read by the launches behind it, so they ride in ITS launch (vine_mlp3_elu_mfma_prep) instead of one of their
            # own; it then pads W1 itself on the way into LDS.  Any operand built from the minibatch keeps the own launch
            ride = (MLP3_PREP and mlp3 and raw is not None and Wop[0].is_contiguous() and Wop[0].shape[1] == F_in
                    and F_in % 2 == 0)
            if l1_mfma and ride:
                x0 = xfull[:, U:width]
                w1p = None
            elif l1_mfma:
                x0 = xfull[:, U:width]                        # strided view; also the operand of layer 1's weight gradient
                w1p = torch.empty((Wop[0].shape[0], 32), device=dev, dtype=op)
                prep.add(CopyBatch.COPY, w1p[:, :F_in], Wop[0])
                prep.add(CopyBatch.ZERO, w1p[:, F_in:])
            else:
                x0 = torch.empty((n, F_in), device=dev, dtype=op)
                prep.add(CopyBatch.CAST_BF16, x0, obs_c)
            if seq:
                # persistent sequence kernels: fragment-ordered weight copies instead of [w_ih | 0 | w_hh] / w_hh^T
                wtile = torch.empty(4 * H * (wpad + H), device=dev, dtype=op)
                w_hh_tiled = torch.empty(4 * H * H, device=dev, dtype=op)
                prep.add_lstm_tiles(w_ih_op, w_hh_op, wpad, wtile, w_hh_tiled)
            elif no_proj:
                wcat = torch.empty((4 * H, wpad + H), device=dev, dtype=op)
                prep.add(CopyBatch.COPY, wcat[:, :width], w_ih_op)
                if wpad > width:
                    prep.add(CopyBatch.ZERO, wcat[:, width:wpad])
                prep.add(CopyBatch.COPY, wcat[:, wpad:], w_hh_op)
            for i in range(1, n_mlp):
                if linear_bwd_mfma_ok(n, Wop[i].shape[1], Wop[i].shape[0]):
                    wts[i] = torch.empty((Wop[i].shape[1], Wop[i].shape[0]), device=dev, dtype=op)
                    prep.add(CopyBatch.TRANSPOSE, wts[i], Wop[i])
            # transposed MLP block of w_ih: the LSTM's input gradient and the last ELU's backward are one kernel
            if n % 64 == 0 and U % 64 == 0 and 4 * H in (512, 1024):
                wts[0] = torch.empty((U, 4 * H), device=dev, dtype=op)
                prep.add(CopyBatch.TRANSPOSE, wts[0], w_ih_op[:, :U])
            # transposed recurrent weight: the backward step forms dG_{t+1} w_hh inside its own kernel
            if not seq and lstm_bwd_mfma_ok(B, H):
                w_hh_t = torch.empty((H, 4 * H), device=dev, dtype=op)
                prep.add(CopyBatch.TRANSPOSE, w_hh_t, w_hh_op)
            prep.add(CopyBatch.COPY, w_heads[:A_], mu_w)
            prep.add(CopyBatch.COPY, w_heads[A_:], v_w)
            prep.add(CopyBatch.COPY, b_heads[:A_], mu_b)
            prep.add(CopyBatch.COPY, b_heads[A_:], v_b)
            prep.add(CopyBatch.ADD, bias, b_ih, b_hh)
            c0_direct = c0 if (c0.dtype == torch.float32 and c0.is_contiguous()) else None
            # persistent kernels: the saved cell states (backward-only) and the gradient w.r.t. the hidden states travel
            # as bfloat16 (LSTM_LP); c_T, the state handed back, stays fp32 in its own buffer
            lp = seq and LSTM_LP and c0_direct is not None
            c_last = torch.empty((B, H), device=dev, dtype=torch.float32) if lp else None
            h0_direct = h0 if (seq and h0.dtype == torch.float32 and h0.is_contiguous()) else None
            # the hidden states once, 16-bit: when their only readers are the fused LayerNorm + heads + loss kernel and
            # the weight-gradient kernel (which both take them that way)
            h_once = (lp and h0_direct is not None and dones is not None and h_once_ok(T)
                      and _heads_loss_route(n, H, A_ + v_w.shape[0], loss_pack, head_bias_external, lib))
            n_param_jobs = len(prep.jobs)
            lstm_buffers = _lstm_state_buffers(xfull, w_hh_op, h0, c0, dones, T, True, prep=prep,
                                               copy_c0=c0_direct is None, c_dtype=lp_dtype() if lp else torch.float32,
                                               mask_h0=h0_direct is None, h_once=h_once)
            side = prep.packed() if (ride and len(prep.jobs) == n_param_jobs) else None
            if ride and side is None:      # (minibatch-derived moves, or too many for one launch: the own launch after all)
                w1p = torch.empty((Wop[0].shape[0], 32), device=dev, dtype=op)
                prep.add(CopyBatch.COPY, w1p[:, :F_in], Wop[0])
                prep.add(CopyBatch.ZERO, w1p[:, F_in:])
            if side is None:
                prep.flush(obs_n)
        else:
            lstm_buffers = None
            c0_direct = None
            side = None
            lp, c_last, h0_direct, h_once = False, None, None, False
            x0 = obs_n.contiguous()
            torch.cat([mu_w, v_w], 0, out=w_heads)
            torch.cat([mu_b, v_b], 0, out=b_heads)
            torch.add(b_ih, b_hh, out=bias)
        acts = []
        x = x0
        if mlp3:
            # the whole MLP in one launch, activations carried in registers (vine_mlp3_elu_mfma)
            acts = [torch.empty((n, 256), device=dev, dtype=op), torch.empty((n, 128), device=dev, dtype=op)]
            if side is not None:
                rc = lib.vine_mlp3_elu_mfma_prep(n, xfull.data_ptr() + 2 * U, xfull.stride(0), raw.data_ptr(), F_in,
                                                 norm[0].data_ptr(), norm[1].data_ptr(), float(norm[2]), 5.0,
                                                 Wop[0].data_ptr(), F_in, mlp[0][1].data_ptr(), 256, Wop[1].data_ptr(),
                                                 Wop[1].stride(0), mlp[1][1].data_ptr(), 128, Wop[2].data_ptr(),
                                                 Wop[2].stride(0), mlp[2][1].data_ptr(), 64, 1.0, acts[0].data_ptr(),
                                                 acts[1].data_ptr(), xfull.data_ptr(), xfull.stride(0), *side, st)
                if rc == -2:       # VINE_ERR_UNSUPPORTED (a move the side job does not cover, or too small a launch to carry
                    side = None    # them): nothing has run -- the moves get their own launch after all
                    w1p = torch.empty((Wop[0].shape[0], 32), device=dev, dtype=op)
                    prep.add(CopyBatch.COPY, w1p[:, :F_in], Wop[0])
                    prep.add(CopyBatch.ZERO, w1p[:, F_in:])
                    prep.flush(obs_n)
                else:
                    _check(rc, "vine_mlp3_elu_mfma_prep")
                    prep.clear()
                    RIDES[0] += 1
            if side is None:
                _check(lib.vine_mlp3_elu_mfma(n, xfull.data_ptr() + 2 * U, xfull.stride(0),
                                              raw.data_ptr() if raw is not None else None, F_in,
                                              norm[0].data_ptr() if raw is not None else None,
                                              norm[1].data_ptr() if raw is not None else None,
                                              float(norm[2]) if raw is not None else 0.0, 5.0,
                                              w1p.data_ptr(), mlp[0][1].data_ptr(), 256, Wop[1].data_ptr(), Wop[1].stride(0),
                                              mlp[1][1].data_ptr(), 128, Wop[2].data_ptr(), Wop[2].stride(0),
                                              mlp[2][1].data_ptr(), 64, 1.0, acts[0].data_ptr(), acts[1].data_ptr(),
                                              xfull.data_ptr(), xfull.stride(0), st), "vine_mlp3_elu_mfma")
        for i, (W, b) in enumerate(mlp if not mlp3 else ()):
            last = i == n_mlp - 1
            if mixed:
                C_, K_ = Wop[i].shape
                a = xcat if last else torch.empty((n, C_), device=dev, dtype=op)
                if i == 0 and l1_mfma:
                    _check(lib.vine_linear_elu_mfma(n, C_, 32, xfull.data_ptr() + 2 * U, xfull.stride(0), w1p.data_ptr(), 32,
                                                    b.data_ptr(), 1.0, a.data_ptr(), a.stride(0), st),
                           "vine_linear_elu_mfma")
                elif linear_elu_mfma_ok(n, C_, K_):      # GEMM + bias + ELU in one matrix-core kernel
                    _check(lib.vine_linear_elu_mfma(n, C_, K_, x.data_ptr(), x.stride(0), Wop[i].data_ptr(),
                                                    Wop[i].stride(0), b.data_ptr(), 1.0, a.data_ptr(), a.stride(0), st),
                           "vine_linear_elu_mfma")
                else:
                    z = _mm(x, Wop[i].t())
                    _check(lib.vine_bias_elu(n, C_, z.data_ptr(), b.data_ptr(), 1.0, a.data_ptr(), a.stride(0), 1, st),
                           "vine_bias_elu")
            else:
                z = torch.addmm(b, x, W.t())
                if last:
                    torch.ops.aten.elu.out(z, out=xcat[:, :U])
                else:
                    a = F.elu_(z)
            if not last:
                acts.append(a)
                x = a
        if concat and not mixed:
            xcat[:, U:].copy_(obs_n)
        NH_ = mu_w.shape[0] + v_w.shape[0]
        slots_early = [_grad_slot(p) if isinstance(p, torch.Tensor) else None for p in params]
        # Round 4: LSTM forward -> LayerNorm + heads + loss -> LSTM backward -> MLP backward as four phases of ONE launch
        # (vine_trunk_phases: same 128 rows per workgroup in all four).  The update's default configuration only; everything
        # the backward pass of this node would compute up to the weight gradients is then done HERE, in forward.
        phases = (TRUNK_PHASES and mixed and seq and h_once and lp and mlp3 and no_proj and T == 4 and wpad == 96 and H == 256
                  and NH_ == 3 and dones is not None and c0_direct is not None and h0_direct is not None
                  and _heads_loss_route(n, H, NH_, loss_pack, head_bias_external, lib) and LOSS_DEFER
                  and lib.vine_ln_heads_loss_rows() == 128 and any(ctx.needs_input_grad)
                  and all(sl is not None for sl, p in zip(slots_early, params) if isinstance(p, torch.Tensor))
                  and all(w is not None for w in wts) and w_hh_tiled is not None
                  and tuple(wts[i].shape for i in (1, 2)) == ((256, 128), (128, 64)) and B % 32 == 0 and n // 128 <= 1024)
        ctx.phases_done = None
        if _os.environ.get("VINE_DEBUG_PHASES") and not phases:
            print("trunk phases off:", dict(mixed=mixed, seq=seq, h_once=h_once, lp=lp, mlp3=mlp3, no_proj=no_proj, T=T, wpad=wpad,
                  H=H, NH=NH_, dones=dones is not None, c0=c0_direct is not None, h0=h0_direct is not None,
                  route=_heads_loss_route(n, H, NH_, loss_pack, head_bias_external, lib), rows=lib.vine_ln_heads_loss_rows(),
                  grad=any(ctx.needs_input_grad), slots=[sl is not None for sl in slots_early], wts=[w is not None for w in wts],
                  tiled=w_hh_tiled is not None, B=B, n=n), flush=True)
        if phases:
            from ..abi import LossFinalize, TrunkArgs
            out, c_all, gates, hp = lstm_buffers
            lpk = loss_pack
            heads = torch.empty((n, NH_), device=dev, dtype=torch.float32)
            d_out = torch.empty((n, H), device=dev, dtype=lp_dtype())
            ln_part = torch.empty((n // 128, (2 + NH_) * H), device=dev, dtype=torch.float32)
            dG = torch.empty((n, 4 * H), device=dev, dtype=lp_dtype())
            bias_partial = torch.empty((B // 32, 4 * H), device=dev, dtype=torch.float32)
            gzs = [torch.empty((n, c), device=dev, dtype=lp_dtype()) for c in (256, 128, 64)]       # layers 1, 2, 3
            parts = [torch.empty((n // 128, c), device=dev, dtype=torch.float32) for c in (256, 128, 64)]
            amp_ = _amp_ptrs(lpk.get("amp"))
            ta = TrunkArgs()
            ta.B, ta.T = B, T
            ta.x, ta.ldx, ta.w_tiled, ta.bias = xfull.data_ptr(), xfull.stride(0), wtile.data_ptr(), bias.data_ptr()
            ta.c0, ta.h0, ta.done = c0_direct.data_ptr(), h0_direct.data_ptr(), dones.data_ptr()
            ta.h_out, ta.c_all, ta.gates, ta.c_last = out.data_ptr(), c_all.data_ptr(), gates.data_ptr(), c_last.data_ptr()
            ta.ln_gamma, ta.ln_beta, ta.ln_eps = ln_g.data_ptr(), ln_b.data_ptr(), float(ln_eps)
            ta.w_heads, ta.b_heads, ta.logstd = w_heads.data_ptr(), b_heads.data_ptr(), lpk["logstd"].data_ptr()
            (ta.actions, ta.old_neglogp, ta.advantages, ta.old_values, ta.returns, ta.old_mu,
             ta.old_sigma) = [t.data_ptr() for t in lpk["args"]]
            ta.e_clip, ta.clip_value, ta.critic_coef, ta.entropy_coef, ta.bounds_coef, ta.soft_bound = lpk["scal"]
            ta.alpha = 1.0
            ta.heads, ta.d_out, ta.ln_partial, ta.loss_partial = heads.data_ptr(), d_out.data_ptr(), ln_part.data_ptr(), lpk["scratch"].data_ptr()
            ta.stats, ta.grad_logstd = lpk["stats"].data_ptr(), lpk["grad_logstd"].data_ptr()
            ta.grad_mu_bias, ta.grad_value_bias = lpk["head_bias_grads"][0].data_ptr(), lpk["head_bias_grads"][1].data_ptr()
            ta.kl_out, ta.logstd_grad_accum, ta.mu_store, ta.sigma_store = lpk["extra"]
            ta.loss_scale, ta.found_inf = amp_
            ta.w_hh_tiled, ta.dgates, ta.bias_partial = w_hh_tiled.data_ptr(), dG.data_ptr(), bias_partial.data_ptr()
            ta.wt0, ta.ldw0, ta.wt1, ta.ldw1 = wts[0].data_ptr(), wts[0].stride(0), wts[2].data_ptr(), wts[2].stride(0)
            ta.wt2, ta.ldw2 = wts[1].data_ptr(), wts[1].stride(0)
            ta.a3, ta.a3_stride, ta.a2, ta.a1 = xcat.data_ptr(), xcat.stride(0), acts[1].data_ptr(), acts[0].data_ptr()
            ta.gz3, ta.gz2, ta.gz1 = gzs[2].data_ptr(), gzs[1].data_ptr(), gzs[0].data_ptr()
            ta.part3, ta.part2, ta.part1 = parts[2].data_ptr(), parts[1].data_ptr(), parts[0].data_ptr()
            import ctypes as C_
            rc = lib.vine_trunk_phases(C_.byref(ta), st)
            if rc == -2:
                phases = False
                if _os.environ.get("VINE_DEBUG_PHASES"):
                    print("vine_trunk_phases: unsupported shapes", B, T, xfull.stride(0), flush=True)
            else:
                _check(rc, "vine_trunk_phases")
                ctx.loss_fin = LossFinalize(lpk["scratch"].data_ptr(), n // 128, NH_ - 1, n, lpk["logstd"].data_ptr(),
                                            lpk["scal"][2], lpk["scal"][3], lpk["scal"][4], lpk["stats"].data_ptr(),
                                            lpk["grad_logstd"].data_ptr(), lpk["head_bias_grads"][0].data_ptr(),
                                            lpk["head_bias_grads"][1].data_ptr(), lpk["extra"][0], lpk["extra"][1], amp_[0])
                ctx.loss_fused = (d_out, ln_part)
                ctx.loss_pack = loss_pack
                ctx.h_once = h_once
                ctx.phases_done = (dG, bias_partial, gzs, parts)
                PHASE_LAUNCHES[0] += 1
        if phases:
            pass
        elif no_proj:
            # no input projection: the step kernel multiplies [x_t | h_{t-1}] by [w_ih | 0 | w_hh] in one product
            out, c_all, gates, hp = _lstm_forward_steps(lib, xfull, None, w_hh_op, bias, h0, c0, dones, T, True, wcat=wcat,
                                                        buffers=lstm_buffers, c0_direct=c0_direct, wtile=wtile,
                                                        c_last=c_last, h0_direct=h0_direct if mixed else None)
        else:
            ig = _mm(xcat, w_ih_op.t())
            out, c_all, gates, hp = _lstm_forward_steps(lib, xcat, ig, w_hh_op, bias, h0, c0, dones, T, True,
                                                        buffers=lstm_buffers, c0_direct=c0_direct)
            del ig
        # LayerNorm and the two heads stay in fp32 in both modes (3 output columns: nothing to gain, and mu feeds
        # the probability ratio directly)
        mean = torch.empty(n, device=dev, dtype=torch.float32)
        rstd = torch.empty(n, device=dev, dtype=torch.float32)
        NH = w_heads.shape[0]
        fuse_heads = H == 256 and 2 <= NH <= 5
        if not phases:
            ctx.loss_fused = None
        ctx.loss_pack = loss_pack       # (its "amp" entry: loss scale / overflow flag of the device-side GradScaler)
        lhl_rows = lib.vine_ln_heads_loss_rows()
        ctx.h_once = h_once
        if phases:
            y = out.new_empty(0)        # (LayerNorm + heads + loss ran as the second phase of vine_trunk_phases)
        elif _heads_loss_route(n, H, NH, loss_pack, head_bias_external, lib):
            # LayerNorm + heads + PPO loss + their backward in ONE launch: the gradient w.r.t. the LSTM output is
            # known before this node's backward runs (which ignores the gradient it is handed for `heads`)
            y = out.new_empty(0)
            heads = torch.empty((n, NH), device=dev, dtype=torch.float32)
            d_out = torch.empty((n, H), device=dev, dtype=lp_dtype() if lp else torch.float32)
            ln_part = torch.empty((n // lhl_rows, (2 + NH) * H), device=dev, dtype=torch.float32)
            lpk = loss_pack
            # the kernel's serial last step (fold of the per-workgroup loss rows -> statistics, KL slot, log-sigma and
            # head-bias gradients) rides in the column-sum launch that ends this node's backward pass, when there is one
            slots_now = [_grad_slot(p) if isinstance(p, torch.Tensor) else None for p in params]
            defer = (LOSS_DEFER and mixed and any(ctx.needs_input_grad)
                     and all(sl is not None for sl, p in zip(slots_now, params) if isinstance(p, torch.Tensor)))
            ctx.loss_fin = None
            if defer:
                from ..abi import LossFinalize
                amp_ = _amp_ptrs(lpk.get("amp"))
                ctx.loss_fin = LossFinalize(lpk["scratch"].data_ptr(), n // lhl_rows, NH - 1, n, lpk["logstd"].data_ptr(),
                                            lpk["scal"][2], lpk["scal"][3], lpk["scal"][4], lpk["stats"].data_ptr(),
                                            lpk["grad_logstd"].data_ptr(), lpk["head_bias_grads"][0].data_ptr(),
                                            lpk["head_bias_grads"][1].data_ptr(), lpk["extra"][0], lpk["extra"][1], amp_[0])
            _check(lib.vine_ln_heads_loss(n, H, NH, out.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), float(ln_eps),
                                          w_heads.data_ptr(), b_heads.data_ptr(), lpk["logstd"].data_ptr(),
                                          *[t.data_ptr() for t in lpk["args"]], *lpk["scal"], heads.data_ptr(),
                                          d_out.data_ptr(), int(lp) | ((2 | (T << 8)) if h_once else 0) | (4 if defer else 0),
                                          ln_part.data_ptr(),
                                          lpk["stats"].data_ptr(),
                                          lpk["grad_logstd"].data_ptr(), lpk["head_bias_grads"][0].data_ptr(),
                                          lpk["head_bias_grads"][1].data_ptr(), lpk["scratch"].data_ptr(), *lpk["extra"],
                                          *_amp_ptrs(lpk.get("amp")), st),
                   "vine_ln_heads_loss")
            ctx.loss_fused = (d_out, ln_part)
        elif fuse_heads:      # LayerNorm + both heads in one kernel; LN(x) is never written
            y = out.new_empty(0)
            heads = torch.empty((n, NH), device=dev, dtype=torch.float32)
            _check(lib.vine_layernorm_heads_forward(n, H, NH, out.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(),
                                                    float(ln_eps), w_heads.data_ptr(), b_heads.data_ptr(),
                                                    heads.data_ptr(), mean.data_ptr(), rstd.data_ptr(), st),
                   "vine_layernorm_heads_forward")
        else:
            y = torch.empty_like(out)
            _check(lib.vine_layernorm_forward(n, H, out.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), float(ln_eps),
                                              y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), st),
                   "vine_layernorm_forward")
            heads = torch.addmm(b_heads, y, w_heads.t())          # [n, A + 1] = [mu | value]
        ctx.meta = (T, concat, n_mlp, U, float(ln_eps), mu_w.shape[0], dones is not None, mixed, head_bias_external)
        ctx.slots = [_grad_slot(p) if isinstance(p, torch.Tensor) else None for p in params]
        ctx.pshapes = [tuple(p.shape) if isinstance(p, torch.Tensor) else None for p in params]
        ctx.fuse_heads = fuse_heads
        ctx.wts = wts
        ctx.w_hh_t = w_hh_t
        ctx.w_hh_tiled = w_hh_tiled
        ctx.has_c0 = c0_direct is not None   # the caller's own c0 buffer is read again in backward (saved below)
        ctx.has_clast = c_last is not None
        ctx.save_for_backward(x0, xcat, hp, out, c_all, gates, y, mean, rstd, w_heads, w_ih_op, w_hh_op, ln_g, ln_b,
                              dones if dones is not None else obs_n.new_empty(0), *acts, *Wop,
                              *([c_last] if c_last is not None else []),
                              *([c0_direct] if c0_direct is not None else []))
        # final LSTM state as views (no copies): the update discards it, other callers may clone
        hT = out.view(B, T + 1, H)[:, T] if h_once else out.view(B, T, H)[:, T - 1]      # (h_once: 16-bit)
        cT = c_last if c_last is not None else c_all[T]
        ctx.mark_non_differentiable(hT, cT)
        ctx.set_materialize_grads(False)       # no zero-filled [B, H] gradients for the two state outputs
        return heads, hT, cT

    @staticmethod
    def backward(ctx, g_heads, _gh, _gc):
        from ..abi import PPO_PARTIAL_BLOCKS
        lib = _lib()
        T, concat, n_mlp, U, ln_eps, A, has_dones, mixed, head_bias_external = ctx.meta
        saved = ctx.saved_tensors
        x0, xcat, hp, out, c_all, gates, y, mean, rstd, w_heads, w_ih, w_hh, ln_g, ln_b, dones = saved[:15]
        acts = list(saved[15:15 + n_mlp - 1])
        weights = list(saved[15 + n_mlp - 1:15 + 2 * n_mlp - 1])
        c0_direct = saved[-1] if ctx.has_c0 else None
        c_last = saved[-2 if ctx.has_c0 else -1] if ctx.has_clast else None
        slots = ctx.slots
        n, H = xcat.shape[0], out.shape[1]      # (with h_once ``out`` has T + 1 rows per sequence)
        dev = out.device
        st = _stream(out)
        grads = [None] * len(slots)

        def deliver(idx, write):
            """Write a parameter gradient into its flat-block slot (autograd then gets None) or hand it back."""
            if slots[idx] is not None:
                write(slots[idx])
            else:
                grads[idx] = torch.empty(ctx.pshapes[idx], device=dev, dtype=torch.float32)
                write(grads[idx])

        base = 2 * n_mlp
        # with the optimiser's gradient slots in place all column sums are deferred into one launch at the end
        amp = ctx.loss_pack.get("amp") if ctx.loss_pack is not None else None
        batch = (ColumnSumBatch(found_inf=amp[1] if (mixed and amp is not None) else None)
                 if all(sl is not None for k, sl in enumerate(slots) if ctx.pshapes[k] is not None) else None)
        if getattr(ctx, "loss_fin", None) is not None:
            assert batch is not None, "the loss finalize was deferred to a column-sum launch that does not exist"
            batch.fin = ctx.loss_fin
            ctx.loss_fin = None
        if amp is not None:
            # do ALL parameter gradients of this backward pass end in the overflow-checked column-sum launch?  (else the
            # optimiser checks the gradient block itself before it steps: FlatAdam.step)
            ctx.loss_pack["amp_covered"] = bool(mixed and batch is not None)
        NH = w_heads.shape[0]
        if ctx.loss_fused is not None:
            d_out, part = ctx.loss_fused             # computed in forward by the fused LayerNorm + heads + loss kernel
            ctx.loss_fused = None
        else:
            g_heads = g_heads.contiguous()
            d_out = torch.empty_like(out)
            part = None
        if part is not None:
            pass
        elif not head_bias_external:      # else: the loss kernel has already added them into the two bias gradients
            gb = g_heads.sum(0)
            deliver(base + 8, lambda o: o.copy_(gb[:A]))
            deliver(base + 10, lambda o: o.copy_(gb[A:]))
        if ctx.fuse_heads:
            # ---- heads + LayerNorm in one kernel: dy = g W is never stored; partial sums of {d gamma | d beta | d W}
            if part is None:
                part = torch.empty((PPO_PARTIAL_BLOCKS, (2 + NH) * H), device=dev, dtype=torch.float32)
                _check(lib.vine_layernorm_heads_backward(n, H, NH, g_heads.data_ptr(), out.data_ptr(), mean.data_ptr(),
                                                         rstd.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(),
                                                         w_heads.data_ptr(), d_out.data_ptr(), part.data_ptr(), st),
                       "vine_layernorm_heads_backward")
            ln_part, w_part = part[:, :2 * H], part[:, 2 * H:]
            if slots[base + 7] is not None and slots[base + 9] is not None:
                column_sums(w_part, slots[base + 7], out1=slots[base + 9], n0=A * H, batch=batch)
            else:
                deliver(base + 7, lambda o: column_sums(w_part[:, :A * H], o))
                deliver(base + 9, lambda o: column_sums(w_part[:, A * H:], o))
        else:
            # ---- heads: one weight-gradient GEMM, one input-gradient GEMM
            s_heads, wpart = _splitk_parts(g_heads, y)                 # [s, A+1, H] slices of the weight gradient
            deliver(base + 7, lambda o: column_sums(wpart[:, :A], o) if s_heads > 1 else o.copy_(wpart[:A]))
            deliver(base + 9, lambda o: column_sums(wpart[:, A:], o) if s_heads > 1 else o.copy_(wpart[A:]))
            dy = g_heads.mm(w_heads)
            # ---- LayerNorm
            ln_part = torch.empty((PPO_PARTIAL_BLOCKS, 2 * H), device=dev, dtype=torch.float32)
            _check(lib.vine_layernorm_backward(n, H, dy.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               ln_g.data_ptr(), d_out.data_ptr(), ln_part.data_ptr(), st),
                   "vine_layernorm_backward")
            del dy
        if slots[base + 4] is not None and slots[base + 5] is not None:
            column_sums(ln_part, slots[base + 4], out1=slots[base + 5], n0=H, batch=batch)     # {d gamma | d beta}
        else:
            deliver(base + 4, lambda o: column_sums(ln_part[:, :H], o))
            deliver(base + 5, lambda o: column_sums(ln_part[:, H:], o))
        # ---- LSTM (+ the MLP backward chain as a second phase of the same launch, round 4: vine_lstm_seq_backward_mlp3_mfma)
        mlp3b = (mixed and MLP3 and n_mlp == 3 and all(w is not None for w in ctx.wts) and n % 64 == 0 and U == 64
                 and 4 * H == 1024 and tuple(ctx.wts[i].shape for i in (1, 2)) == ((256, 128), (128, 64))
                 and acts[0].shape[1] == 256 and acts[1].shape[1] == 128)
        gzs = parts = None
        B_ = n // T
        if getattr(ctx, "phases_done", None) is not None:      # vine_trunk_phases ran the LSTM / MLP backward in forward
            dG, bias_partial, gzs, parts = ctx.phases_done
            ctx.phases_done = None
        elif (BWD_PHASES and mlp3b and ctx.w_hh_tiled is not None and T == 4 and c_last is not None and c0_direct is not None
                and d_out.dtype == lp_dtype() and c_all.dtype == lp_dtype() and n % 128 == 0 and n >= 32768 and B_ % 32 == 0
                and d_out.is_contiguous() and gates.is_contiguous() and c_all.is_contiguous()):
            dG = torch.empty((n, 4 * H), device=dev, dtype=lp_dtype())
            bias_partial = torch.empty((B_ // 32, 4 * H), device=dev, dtype=torch.float32)
            gzs = [torch.empty((n, c), device=dev, dtype=lp_dtype()) for c in (256, 128, 64)]       # layers 1, 2, 3
            parts = [torch.empty((n // 128, c), device=dev, dtype=torch.float32) for c in (256, 128, 64)]
            rc = lib.vine_lstm_seq_backward_mlp3_mfma(
                B_, T, H, d_out.data_ptr(), ctx.w_hh_tiled.data_ptr(), gates.data_ptr(), c_all.data_ptr(),
                c0_direct.data_ptr(), dones.data_ptr() if has_dones else None, dG.data_ptr(), bias_partial.data_ptr(),
                c_last.data_ptr(), ctx.wts[0].data_ptr(), ctx.wts[0].stride(0), ctx.wts[2].data_ptr(), ctx.wts[2].stride(0),
                ctx.wts[1].data_ptr(), ctx.wts[1].stride(0), xcat.data_ptr(), xcat.stride(0), acts[1].data_ptr(),
                acts[0].data_ptr(), 1.0, gzs[2].data_ptr(), gzs[1].data_ptr(), gzs[0].data_ptr(), parts[2].data_ptr(),
                parts[1].data_ptr(), parts[0].data_ptr(), st)
            if rc == -2:
                gzs = parts = None
            else:
                _check(rc, "vine_lstm_seq_backward_mlp3_mfma")
        if gzs is None:
            dG, bias_partial = _lstm_backward_steps(lib, d_out, w_hh, c_all, gates, dones if has_dones else None, T,
                                                    w_hh_t=ctx.w_hh_t, c0_direct=c0_direct, w_hh_tiled=ctx.w_hh_tiled,
                                                    c_last=c_last)
        fused2 = False
        ctx.wgrad_fork = None
        if WGRAD_SIDE and gzs is not None:
            cur = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            side.wait_stream(cur)            # (fork: everything the MLP weight gradients read is complete on `cur`)
            ctx.wgrad_fork = (side, cur)
        if ctx.h_once and mixed and slots[base + 0] is not None and slots[base + 1] is not None:
            # ``out`` is the one 16-bit copy of the hidden states (slot 0 = h0): the kernel shifts and masks
            fused2 = weight_grad_cat(dG, xcat, out, slots[base + 0], slots[base + 1], batch=batch, seq=(dones, T))
        if ctx.h_once and not fused2:
            hp = masked_previous_hidden(out, dones if has_dones else None, T)
        if not fused2 and mixed and slots[base + 0] is not None and slots[base + 1] is not None:
            # dW_ih and dW_hh from one pass over dG (the largest tensor of the backward pass)
            fused2 = weight_grad_cat(dG, xcat, hp.view(n, H), slots[base + 0], slots[base + 1], batch=batch)
        if not fused2:
            deliver(base + 0, lambda o: weight_grad(dG, xcat, out=o, batch=batch))
            deliver(base + 1, lambda o: weight_grad(dG, hp.view(n, H), out=o, batch=batch))
        if bias_partial is not None and slots[base + 2] is not None and slots[base + 3] is not None:
            column_sums(bias_partial.view(-1, 4 * H), slots[base + 2], out1=slots[base + 3], dup=True, batch=batch)
        else:
            deliver(base + 2, lambda o: _sum_rows(bias_partial, dG.float(), out=o))
            deliver(base + 3, lambda o: o.copy_(slots[base + 2] if slots[base + 2] is not None else grads[base + 2]))
        gz = part = g = None
        wgroup = WeightGradGroup()
        if mlp3b:
            # the whole MLP backward in one launch (vine_mlp3_bwd_elu_mfma): LSTM input gradient (MLP columns) x ELU' ->
            # gz3 -> gz2 -> gz1 carried in registers, bias partial sums per workgroup
            if gzs is None:      # (else: it ran as the second phase of the LSTM backward launch above)
                rows_wg = 128 if (n % 128 == 0 and n >= 32768) else 64
                gzs = [torch.empty((n, c), device=dev, dtype=lp_dtype()) for c in (256, 128, 64)]       # layers 1, 2, 3
                parts = [torch.empty((n // rows_wg, c), device=dev, dtype=torch.float32) for c in (256, 128, 64)]
                _check(lib.vine_mlp3_bwd_elu_mfma(n, dG.data_ptr(), dG.stride(0), 4 * H, ctx.wts[0].data_ptr(),
                                                  ctx.wts[0].stride(0), ctx.wts[2].data_ptr(), ctx.wts[2].stride(0),
                                                  ctx.wts[1].data_ptr(), ctx.wts[1].stride(0), xcat.data_ptr(), xcat.stride(0),
                                                  acts[1].data_ptr(), acts[0].data_ptr(), 64, 128, 256, 1.0,
                                                  gzs[2].data_ptr(), gzs[1].data_ptr(), gzs[0].data_ptr(),
                                                  parts[2].data_ptr(), parts[1].data_ptr(), parts[0].data_ptr(), st),
                       "vine_mlp3_bwd_elu_mfma")
            del dG
            for i in range(n_mlp):
                x_in = acts[i - 1] if i > 0 else x0
                deliver(2 * i, lambda o, gz=gzs[i], x_in=x_in: wgroup.add(gz, x_in, o, batch) or weight_grad(gz, x_in, out=o, batch=batch))
                deliver(2 * i + 1, lambda o, part=parts[i]: column_sums(part, o, batch=batch))
            n_mlp_loop = 0
        else:
            n_mlp_loop = n_mlp
        if mlp3b:
            pass
        elif mixed and ctx.wts[0] is not None:
            # LSTM input gradient (MLP columns only) x ELU' of the last MLP layer + its bias partial sums: one
            # matrix-core kernel streaming the 4H-long reduction in k chunks
            gz = torch.empty((n, U), device=dev, dtype=lp_dtype())
            part = torch.empty((n // 64, U), device=dev, dtype=torch.float32)
            _check(lib.vine_linear_bwd_elu_mfma(n, U, 4 * H, dG.data_ptr(), dG.stride(0), ctx.wts[0].data_ptr(),
                                                ctx.wts[0].stride(0), xcat.data_ptr(), xcat.stride(0), 1.0,
                                                gz.data_ptr(), U, part.data_ptr(), st), "vine_linear_bwd_elu_mfma")
        else:
            g = _mm(dG, w_ih[:, :U] if concat else w_ih)    # only the MLP columns of the LSTM input need a gradient
        if not mlp3b:
            del dG
        # ---- MLP, last layer first: ELU' from the stored OUTPUT, bias gradient from the kernels' partial sums.
        # gz = gradient w.r.t. layer i's pre-activation.  In mixed precision the step from layer i to layer i-1
        # (input-gradient GEMM + ELU backward + bias partial sums) is one matrix-core kernel.
        for i in reversed(range(n_mlp_loop)):
            if gz is None:
                a = xcat if i == n_mlp - 1 else acts[i]
                C_ = g.shape[1]
                part = torch.empty((PPO_PARTIAL_BLOCKS, C_), device=dev, dtype=torch.float32)
                gz = torch.empty((n, C_), device=dev, dtype=lp_dtype()) if mixed else g     # fp32: in place
                _check(lib.vine_elu_backward(n, C_, g.data_ptr(), C_, a.data_ptr(), a.stride(0), 1.0, gz.data_ptr(), C_,
                                             part.data_ptr(), int(mixed), int(mixed), st), "vine_elu_backward")
            x_in = acts[i - 1] if i > 0 else x0
            deliver(2 * i, lambda o, gz=gz, x_in=x_in: wgroup.add(gz, x_in, o, batch) or weight_grad(gz, x_in, out=o, batch=batch))
            deliver(2 * i + 1, lambda o, part=part: column_sums(part, o, batch=batch))
            if i == 0:
                break
            C_in = weights[i].shape[1]
            if mixed and ctx.wts[i] is not None:
                wt = ctx.wts[i]                                                    # [C_in, C_i] bf16, made in forward
                gz_next = torch.empty((n, C_in), device=dev, dtype=lp_dtype())
                part = torch.empty((n // 64, C_in), device=dev, dtype=torch.float32)
                _check(lib.vine_linear_bwd_elu_mfma(n, C_in, gz.shape[1], gz.data_ptr(), gz.stride(0), wt.data_ptr(),
                                                    wt.stride(0), acts[i - 1].data_ptr(), acts[i - 1].stride(0), 1.0,
                                                    gz_next.data_ptr(), C_in, part.data_ptr(), st),
                       "vine_linear_bwd_elu_mfma")
                gz = gz_next
            else:
                g = _mm(gz, weights[i])
                gz = None
        # the MLP weight gradients: one launch, all operands exist by now.  WGRAD_SIDE (round 4 experiment): on a side stream --
        # a parallel branch of the captured graph -- beside the LSTM weight-gradient kernel issued above: that kernel holds
        # 8 waves x 196 VGPRs and 64 KB of LDS per CU, which leaves room for exactly one 4-wave workgroup of this one
        if WGRAD_SIDE and getattr(ctx, "wgrad_fork", None) is not None:
            side, cur = ctx.wgrad_fork
            with torch.cuda.stream(side):
                wgroup.flush()
            cur.wait_stream(side)
            ctx.wgrad_fork = None
        else:
            wgroup.flush()
        if batch is not None:
            if batch.bypassed and amp is not None:
                ctx.loss_pack["amp_covered"] = False     # (coverage is per delivery, not per batch)
            batch.flush(out)
        return (None, None, None, None, None, None, None, None, None, None, None, *grads)


def linear_elu_mfma_ok(n, N, K):
    """Shapes vine_linear_elu_mfma covers (else: GEMM + vine_bias_elu)."""
    return n % 64 == 0 and N % 64 == 0 and K in (32, 64, 128, 256)


def linear_bwd_mfma_ok(n, N, K):
    """Shapes vine_linear_bwd_elu_mfma covers (else: GEMM + vine_elu_backward)."""
    return n % 64 == 0 and N % 64 == 0 and K in (64, 128, 256)


def trunk_supported(obs_n, mlp_units, activation_is_elu, H, has_ln, T):
    """The fused trunk covers the reference's network (PY:10-40) on an MI355X in fp32."""
    return (obs_n.is_cuda and obs_n.dtype == torch.float32 and torch.is_grad_enabled() and activation_is_elu and has_ln
            and H in (256, 512, 1024) and all(u % 4 == 0 and 256 % (u // 4) == 0 for u in mlp_units)
            and obs_n.shape[0] % T == 0 and obs_n.shape[0] >= 256)


def trunk(obs_n, h0, c0, dones, T, concat, mlp_params, lstm_params, ln, heads, op_weights=None,
          head_bias_external=False, norm=None, loss_pack=None):
    """-> (heads [n, A+1] = [mu | value], hT, cT).  ``mlp_params`` = [(W, b), ...]; ``lstm_params`` =
    (w_ih, w_hh, b_ih, b_hh); ``ln`` = (gamma, beta, eps); ``heads`` = (mu_w, mu_b, value_w, value_b).
    ``op_weights`` = bfloat16 copies of (W_1, ..., W_L, w_ih, w_hh) selects the mixed-precision update: those GEMMs
    take bf16 operands (fp32 accumulate and output), activations exist only as bf16 GEMM operands; LSTM cell state,
    gates, LayerNorm, heads, loss, gradients w.r.t. parameters and the optimiser stay fp32.
    ``head_bias_external``: the caller obtains the two head-bias gradients elsewhere (the PPO loss kernel adds them to
    the parameters' gradient slots), so the backward skips that column sum over all n rows.
    ``norm`` = (running mean, running var, eps) of the observation normaliser: ``obs_n`` then holds the RAW observations
    and the trunk normalises them itself (inside the one-launch MLP kernel where that applies).
    ``loss_pack`` (``ppo_loss_pack``): the PPO loss and its gradient are computed inside the node (one launch for
    LayerNorm + heads + loss + their backward); ``heads`` is still returned, the statistics land in the pack's tensors,
    and ``torch.autograd.backward([heads], [anything of the same shape])`` runs the rest of the backward pass."""
    if dones is not None:
        dones = dones.to(torch.uint8).contiguous()
    flat = [p for wb in mlp_params for p in wb]
    return _Trunk.apply(obs_n, h0.contiguous(), c0.contiguous(), dones, T, bool(concat), len(mlp_params),
                        tuple(op_weights) if op_weights is not None else None, bool(head_bias_external),
                        tuple(norm) if norm is not None else None, loss_pack, *flat,
                        *lstm_params, ln[0], ln[1], float(ln[2]), *heads)


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


def ppo_loss_pack(logstd, actions, old_neglogp, adv, old_values, returns, old_mu, old_sigma, e_clip, clip_value,
                  critic_coef, entropy_coef, bounds_coef, head_bias_grads, soft_bound=1.1, kl_out=None, logstd_grad=None,
                  update_old=False, stats_out=None, amp=None):
    """Arguments of the PPO loss for the trunk's one-launch LayerNorm + heads + loss kernel (``trunk(loss_pack=...)``):
    same meaning as ``ppo_loss_fused``; ``pack["stats"]`` / ``pack["grad_logstd"]`` receive the results."""
    from ..abi import PPO_LOSS_SCRATCH_FLOATS
    dev = logstd.device
    args = [t.detach().contiguous() for t in (actions, old_neglogp, adv, old_values.reshape(-1), returns.reshape(-1),
                                              old_mu, old_sigma)]
    if update_old:
        assert args[5].data_ptr() == old_mu.data_ptr() and args[6].data_ptr() == old_sigma.data_ptr(), \
            "update_old needs contiguous old_mu / old_sigma (they are written in place)"
    ls = logstd.detach().contiguous()
    stats = stats_out if stats_out is not None else torch.empty(8, device=dev, dtype=torch.float32)
    assert stats.is_contiguous() and stats.numel() == 8 and stats.dtype == torch.float32
    return {"logstd": ls, "args": args,
            "scal": (float(e_clip), int(bool(clip_value)), float(critic_coef), float(entropy_coef), float(bounds_coef),
                     float(soft_bound)),
            "stats": stats, "grad_logstd": torch.empty(ls.shape[0], device=dev, dtype=torch.float32),
            "head_bias_grads": head_bias_grads,
            "scratch": torch.empty(PPO_LOSS_SCRATCH_FLOATS, device=dev, dtype=torch.float32),
            "extra": (kl_out.data_ptr() if kl_out is not None else None,
                      logstd_grad.data_ptr() if logstd_grad is not None else None,
                      args[5].data_ptr() if update_old else None, args[6].data_ptr() if update_old else None),
            "keep": (kl_out, logstd_grad),
            # (loss scale, overflow flag) of the device-side GradScaler, or None; "amp_covered" is filled in by the backward
            "amp": amp, "amp_covered": False}


def ppo_loss_fused(mu, logstd, value, actions, old_neglogp, adv, old_values, returns, old_mu, old_sigma, e_clip,
                   clip_value, critic_coef, entropy_coef, bounds_coef, soft_bound=1.1, heads=None, head_bias_grads=None,
                   kl_out=None, logstd_grad=None, update_old=False, stats_out=None, amp=None):
    """One HIP kernel: returns (grad_mu [n,A], grad_value [n,1], grad_logstd [A], stats[8]) where the gradients are
    d(loss)/d(.) of the same scalar loss as ``ppo_loss_reference``.  With ``heads`` ([n, A+1] = [mu | value], the
    output of the fused trunk) mu/value are read from it in place and the first return value is the matching
    [n, A+1] gradient (second is None).  ``head_bias_grads`` = (mu.bias.grad, value.bias.grad): the kernel adds the
    column sums of the head gradients to them (they must hold zeros, as the optimiser leaves them).
    ``kl_out`` (1 float) receives the mean KL, ``logstd_grad`` ([A]) gets the log-sigma gradient added, ``update_old``
    writes the new mu / sigma of every sample over ``old_mu`` / ``old_sigma`` (dataset.update_mu_sigma) -- each saves
    the update a small launch.  ``stats_out`` (8 floats): where the statistics go (default: a new tensor)."""
    lib = _lib()
    stats_dev = (heads if heads is not None else mu).device
    args = [t.detach().contiguous() for t in (actions, old_neglogp, adv, old_values.reshape(-1), returns.reshape(-1),
                                              old_mu, old_sigma)]
    if update_old:
        assert args[5].data_ptr() == old_mu.data_ptr() and args[6].data_ptr() == old_sigma.data_ptr(), \
            "update_old needs contiguous old_mu / old_sigma (they are written in place)"
    extra = (kl_out.data_ptr() if kl_out is not None else None, logstd_grad.data_ptr() if logstd_grad is not None else None,
             args[5].data_ptr() if update_old else None, args[6].data_ptr() if update_old else None)
    ls = logstd.detach().contiguous()
    A = ls.shape[0]
    grad_logstd = torch.empty(A, device=stats_dev, dtype=torch.float32)
    stats = stats_out if stats_out is not None else torch.empty(8, device=stats_dev, dtype=torch.float32)
    assert stats.is_contiguous() and stats.numel() == 8 and stats.dtype == torch.float32
    from ..abi import PPO_LOSS_SCRATCH_FLOATS
    scratch = torch.empty(PPO_LOSS_SCRATCH_FLOATS, device=stats_dev, dtype=torch.float32)
    scal = (float(e_clip), int(bool(clip_value)), float(critic_coef), float(entropy_coef), float(bounds_coef),
            float(soft_bound))
    if heads is not None:
        hd = heads.detach()
        assert hd.is_contiguous() and hd.shape[1] == A + 1
        n = hd.shape[0]
        g = torch.empty_like(hd)
        _check(lib.vine_ppo_loss(n, A, hd.data_ptr(), ls.data_ptr(), hd.data_ptr() + 4 * A, *[a.data_ptr() for a in args],
                                 *scal, g.data_ptr(), g.data_ptr() + 4 * A, grad_logstd.data_ptr(), stats.data_ptr(),
                                 A + 1, A + 1, head_bias_grads[0].data_ptr() if head_bias_grads else None,
                                 head_bias_grads[1].data_ptr() if head_bias_grads else None, scratch.data_ptr(), *extra,
                                 _amp_ptrs(amp)[0], _stream(hd)),
               "vine_ppo_loss")
        return g, None, grad_logstd, stats
    n = mu.shape[0]
    mu_c = mu.detach().contiguous()
    val_c = value.detach().reshape(-1).contiguous()
    grad_mu = torch.empty_like(mu_c)
    grad_value = torch.empty_like(val_c)
    _check(lib.vine_ppo_loss(n, A, mu_c.data_ptr(), ls.data_ptr(), val_c.data_ptr(), *[a.data_ptr() for a in args],
                             *scal, grad_mu.data_ptr(), grad_value.data_ptr(), grad_logstd.data_ptr(),
                             stats.data_ptr(), 0, 0, None, None, scratch.data_ptr(), *extra, _amp_ptrs(amp)[0], _stream(mu)),
           "vine_ppo_loss")
    return grad_mu, grad_value.view_as(value), grad_logstd, stats
