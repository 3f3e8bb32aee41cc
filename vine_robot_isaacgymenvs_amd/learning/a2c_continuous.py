"""``A2CAgent``: PPO with an LSTM actor-critic as rl_games' ``a2c_continuous`` runs it for
cfg/train/Vine5LinkMovingBasePPO.yaml (rows R1-R6 of SURVEY 8a; in-tree text of the same arithmetic:
isaacgymenvs/learning/common_agent.py:184-255 train_epoch, 257-317 play_steps, 319-411 calc_gradients,
413-435 discount_values/bound_loss).

MI355X-first choices (none changes the arithmetic):
  * everything in the rollout and in the update stays on the device: done-handling uses masks instead of
    index lists, the adaptive-KL learning rate (``schedule_type: legacy``: after EVERY minibatch) lives in a
    device scalar read by the fused Adam kernel -- no ``.item()`` inside an iteration;
  * gradients live in ONE flat buffer (``param.grad`` are views into it): the multi-GPU step is a single
    in-place RCCL all-reduce of 1.63 MB over xGMI per optimiser step, no concat / scatter copies;
  * the 16-step rollout is captured in one hipGraph (``use_graphs``): policy GEMMs + the fused env-step
    kernel + bookkeeping replay with a single launch; from the second iteration on every optimiser step is two
    hipGraph replays as well (forward/backward | Adam + schedule) with the all-reduce between them;
  * ``mixed_precision: True`` (the reference YAML's value) = the reference's precisions, hand-written: the UPDATE takes
    fp16 GEMM operands with fp32 accumulation, state, loss and optimiser (learning/fused.py ``trunk``) and
    torch.amp.GradScaler's semantics restated on the device (FlatAdam.enable_loss_scaling), not autocast; rollout
    inference runs in fp32 on the matrix cores (``_infer``: vine_mlp3_elu_f32 / vine_lstm_step_f32).
"""
import copy
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import fused
from .. import abi
from ..abi import ROLLOUT_POST_SCRATCH_FLOATS
from .flat_adam import FlatAdam
from .network import ModelA2CContinuousLogStd


class _Range:
    """roctx range (torch.cuda.nvtx is roctx on ROCm) around a phase of the iteration -- rollout, update, each graph
    replay -- so that `rocprofv3 --marker-trace` timelines carry the phase names.  Off unless VINE_ROCTX=1: a push/pop
    pair is host work inside the launch-bound loop."""
    enabled = os.environ.get("VINE_ROCTX", "0") == "1"

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _Range.enabled:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *a):
        if _Range.enabled:
            torch.cuda.nvtx.range_pop()


def wait_bounded(done, timeout_s, clock=time.monotonic, sleep=time.sleep, poll_s=0.002):
    """Poll ``done()`` (an event's ``query``, a work handle's ``is_completed``) until it answers True or ``timeout_s``
    has passed; returns whether it answered.  The waits on the first captured collectives use this instead of an
    unbounded ``synchronize``: a replayed RCCL node that a peer never joins would otherwise hang the job silently."""
    deadline = clock() + float(timeout_s)
    while True:
        if done():
            return True
        if clock() >= deadline:
            return False
        sleep(poll_s)


PROBE_EXIT_CODE = 3


def collective_probe_protocol(capture, replay, agree, abort, log=print):
    """Decide, with every rank, whether the gradient all-reduce may live inside a captured graph (VERDICT r4 item 6,
    ADVICE r4).  Two agreed stages, so that no rank ever replays a collective a peer will not join:

      1. ``capture()`` -> (ok, why, handle): capture the throw-away graph (nothing executes).  ``agree(ok)``: MIN over
         the ranks -- unless ALL of them captured, nobody replays.
      2. ``replay(handle)`` -> "ok" | "wrong sum" | "timeout": replay with a BOUNDED wait.  ``agree(result == "ok")``.

    ``agree`` returns True / False, or None when the agreement collective itself did not answer within its deadline.  A
    replay that timed out leaves an unfinished collective on this rank's stream: if the ranks can still agree they all
    fall back to the eager collective between per-step graphs; if not, ``abort(message)`` ends the process with a
    non-zero code and the instruction to relaunch with VINE_COLLECTIVE_IN_GRAPH=0 (a process that has touched the GPU is
    never re-executed in place).  Returns (in_graph, why)."""
    ok, why, handle = capture()
    every = agree(bool(ok))
    if every is None:
        abort("collective probe: the ranks could not agree on the capture outcome (%s); relaunch with "
              "VINE_COLLECTIVE_IN_GRAPH=0" % why)
        return False, "no agreement after capture: " + why
    if not every:
        return False, (why if not ok else "another rank could not capture the all-reduce")
    res = replay(handle)
    good = res == "ok"
    if res == "timeout":
        log("collective probe: the captured all-reduce did not complete within its deadline on this rank; asking the "
            "other ranks to fall back to the eager collective")
    every = agree(good)
    if every is None:
        abort("collective probe: a replayed all-reduce hung (%s) and the ranks no longer answer; relaunch with "
              "VINE_COLLECTIVE_IN_GRAPH=0 (all-reduce between per-step graphs)" % res)
        return False, "captured all-reduce: %s, no agreement" % res
    if not every:
        return False, ("captured all-reduce: %s" % res if not good else "another rank's captured all-reduce failed")
    return True, "captured all-reduce replays correctly"


def swap_and_flatten01(arr):
    """[T, N, ...] -> [N*T, ...] with index = env * T + t (each env's steps contiguous)."""
    if arr is None:
        return arr
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma):
    """KL(p0 || p1) of diagonal Gaussians with rl_games' +1e-5 guards, mean over the batch."""
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    kl = (c1 + c2 - 0.5).sum(dim=-1)
    return kl.mean()


def actor_loss(old_neglogp, neglogp, advantage, e_clip):
    """Clipped surrogate, common_agent.py:482-493."""
    ratio = torch.exp(old_neglogp - neglogp)
    surr1 = advantage * ratio
    surr2 = advantage * torch.clamp(ratio, 1.0 - e_clip, 1.0 + e_clip)
    return torch.max(-surr1, -surr2)


def critic_loss(value_preds, values, e_clip, returns, clip_value):
    """common_agent.py:503-511."""
    if clip_value:
        value_pred_clipped = value_preds + (values - value_preds).clamp(-e_clip, e_clip)
        return torch.max((values - returns) ** 2, (value_pred_clipped - returns) ** 2)
    return (returns - values) ** 2


def bound_loss(mu, soft_bound=1.1):
    """common_agent.py:427-435 with rl_games' soft bound of 1.1 (the in-tree AMP copy uses 1.0)."""
    mu_loss_high = torch.clamp_min(mu - soft_bound, 0.0) ** 2
    mu_loss_low = torch.clamp_max(mu + soft_bound, 0.0) ** 2
    return (mu_loss_low + mu_loss_high).sum(dim=-1)


def discount_values(gamma, tau, fdones, last_values, mb_fdones, mb_values, mb_rewards):
    """GAE in next-nonterminal form (rl_games a2c_common.discount_values)."""
    horizon = mb_rewards.shape[0]
    lastgaelam = torch.zeros_like(last_values)
    mb_advs = torch.zeros_like(mb_rewards)
    for t in reversed(range(horizon)):
        if t == horizon - 1:
            nextnonterminal = 1.0 - fdones
            nextvalues = last_values
        else:
            nextnonterminal = 1.0 - mb_fdones[t + 1]
            nextvalues = mb_values[t + 1]
        nextnonterminal = nextnonterminal.unsqueeze(1)
        delta = mb_rewards[t] + gamma * nextvalues * nextnonterminal - mb_values[t]
        lastgaelam = delta + gamma * tau * nextnonterminal * lastgaelam
        mb_advs[t] = lastgaelam
    return mb_advs


class DeviceAverageMeter:
    """rl_games ``AverageMeter`` (windowed mean of finished-episode returns) with masked, sync-free updates."""

    def __init__(self, in_shape, max_size, device):
        self.max_size = float(max_size)
        self.current_size = torch.zeros((), device=device)
        self.mean = torch.zeros(in_shape, device=device)

    def update(self, values, mask):
        """values [N, in_shape], mask [N] (1 where an episode just finished)."""
        m = mask.to(values.dtype)
        size = m.sum()
        new_mean = (values * m.unsqueeze(-1)).sum(0) / torch.clamp(size, min=1.0)
        size_c = torch.clamp(size, 0.0, self.max_size)
        old_size = torch.minimum(self.max_size - size_c, self.current_size)
        size_sum = old_size + size_c
        mean = (self.mean * old_size + new_mean * size_c) / torch.clamp(size_sum, min=1.0)
        has = (size > 0).to(values.dtype)
        self.mean = has * mean + (1.0 - has) * self.mean
        self.current_size = has * size_sum + (1.0 - has) * self.current_size

    def clear(self):
        self.current_size.zero_()
        self.mean.zero_()

    def get_mean(self):
        return self.mean


class ScalarLog:
    """The ``SummaryWriter`` rl_games logs through (tensorboardX; not on the image): same ``add_scalar(tag, value, step)``
    call, written twice -- as a TensorBoard event file (``utils/tfevents.py`` encodes the format itself, so a stock
    ``tensorboard --logdir`` elsewhere reads the run) and appended to ``summaries/scalars.csv``."""

    def __init__(self, directory):
        from ..utils.tfevents import EventFileWriter
        os.makedirs(directory, exist_ok=True)
        self.path = os.path.join(directory, "scalars.csv")
        self._f = open(self.path, "a")
        self.events = EventFileWriter(directory)

    def add_scalar(self, tag, value, step):
        self._f.write("%s,%s,%s\n" % (tag, float(value), int(step)))
        self.events.add_scalar(tag, float(value), int(step))

    def flush(self):
        self._f.flush()
        self.events.flush()


class A2CAgent:
    def __init__(self, base_name, params, vec_env=None, algo_observer=None):
        self.params = params
        self.config = config = params["config"]
        self.network_params = params["network"]
        self.vec_env = vec_env
        from ..utils.rlgames_utils import RLGPUEnv
        if self.vec_env is None:
            self.vec_env = RLGPUEnv(config["env_name"], config["num_actors"])
        elif not hasattr(self.vec_env, "get_env_info"):
            self.vec_env = RLGPUEnv.wrap(self.vec_env)
        self.env_info = self.vec_env.get_env_info()

        self.multi_gpu = bool(config.get("multi_gpu", False))
        self.rank, self.rank_size = 0, 1
        if self.multi_gpu:
            self.rank = int(os.getenv("LOCAL_RANK", "0"))
            self.rank_size = int(os.getenv("WORLD_SIZE", "1"))
            if str(config.get("device", "cuda:0")).startswith("cuda") and not config.get("device_pinned", False):
                config["device"] = "cuda:" + str(self.rank)
        self.ppo_device = self.device = torch.device(config.get("device", "cuda:0"))
        self.is_cuda = self.device.type == "cuda"
        if self.is_cuda and self.device.index is None:       # "cuda": the process's current device, made explicit
            self.ppo_device = self.device = torch.device("cuda", torch.cuda.current_device())
        if self.is_cuda:
            # The hand-written kernels are launched through ctypes on `torch.cuda.current_stream(device)`; for the
            # default stream that handle is 0 = "the null stream of the CURRENT device", and torch.cuda.graph() opens
            # its capture stream on the current device too: the rank's GPU must be the current device of this process
            # (train.py:71-75 of the reference relies on rl_games doing the same).
            torch.cuda.set_device(self.device)
        if self.multi_gpu and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {"device_id": self.device} if self.is_cuda else {}
            dist.init_process_group("nccl" if self.is_cuda else "gloo", rank=int(os.getenv("RANK", self.rank)),
                                    world_size=self.rank_size, **kw)
        # what actually runs, for logs and bench.py: "off" | "graph" | "eager (capture refused)" per phase
        self.graph_status = {"rollout": "off", "update": "off"}

        self.name = base_name
        self.ppo = config.get("ppo", True)
        self.max_epochs = config.get("max_epochs", 1e6)
        self.num_actors = config["num_actors"]
        self.num_agents = 1
        self.horizon_length = config["horizon_length"]
        self.seq_len = config.get("seq_len", 4)
        self.normalize_advantage = config["normalize_advantage"]
        self.normalize_input = config["normalize_input"]
        self.normalize_value = config.get("normalize_value", False)
        self.truncate_grads = config.get("truncate_grads", False)
        self.grad_norm = config["grad_norm"]
        self.gamma, self.tau = config["gamma"], config["tau"]
        self.e_clip = config["e_clip"]
        self.clip_value = config["clip_value"]
        self.critic_coef = config["critic_coef"]
        self.entropy_coef = config["entropy_coef"]
        self.bounds_loss_coef = config.get("bounds_loss_coef", None)
        self.mini_epochs_num = config["mini_epochs"]
        self.value_bootstrap = config.get("value_bootstrap")
        self.reward_scale = config.get("reward_shaper", {}).get("scale_value", 1.0)
        self.reward_shift = config.get("reward_shaper", {}).get("shift_value", 0.0)
        # mixed_precision (PY:53): rl_games runs the UPDATE under fp16 autocast with a GradScaler; rollout inference stays
        # fp32.  On the MI355X with the fused ops this is the hand-written mixed-precision update (`fused_mixed`): GEMM
        # operands and backward-only saved activations in the library's 16-bit format -- float16 in the default build,
        # i.e. the reference's dtype, with the GradScaler restated on the device (loss scale in the loss kernel, overflow
        # flags, skip / back-off / growth in the Adam kernel) -- fp32 accumulation, state, loss and optimiser; no
        # autocast.  `mixed_precision_dtype` other than the library's format (or use_fused_ops: False) selects torch
        # autocast instead, the reference's literal mechanism, slower than fp32 here.
        # `rollout_precision` (not an rl_games key): "fp32" = the reference's (hand-written fp32 matrix-core inference
        # kernels); "lp16" = 16-bit GEMM operands in the rollout too (the round-2 behaviour, an extra).
        want_mixed = bool(config.get("mixed_precision", False)) and self.is_cuda
        mp_dtype = config.get("mixed_precision_dtype", "fp16")
        self.use_fused = bool(config.get("use_fused_ops", True))
        self.amp_dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[mp_dtype]
        self.fused_mixed = want_mixed and self.use_fused and self.amp_dtype == fused.lp_dtype()
        self.mixed_precision = want_mixed and not self.fused_mixed          # the torch-autocast path
        self.rollout_lp16 = self.fused_mixed and config.get("rollout_precision", "fp32") != "fp32"
        # fp32 rollout LSTM step: 9 / 6 = products formed from bf16 pieces (vine_lstm_step_f32_split), 0 = native fp32 MFMA
        self.rollout_f32_terms = int(config.get("rollout_f32_terms", fused.ROLLOUT_F32_SPLIT))
        self.save_freq = config.get("save_frequency", 0)
        self.save_best_after = config.get("save_best_after", 100)
        self.print_stats = config.get("print_stats", True)
        self.games_to_track = config.get("games_to_track", 100)
        self.use_graphs = bool(config.get("use_graphs", False)) and self.is_cuda
        self.schedule_type = config.get("schedule_type", "legacy")
        self.is_adaptive_lr = config["lr_schedule"] == "adaptive"
        self.kl_threshold = config.get("kl_threshold", 0.008)
        self.min_lr, self.max_lr = 1e-6, 1e-2

        self.batch_size = self.horizon_length * self.num_actors * self.num_agents
        self.batch_size_envs = self.horizon_length * self.num_actors
        self.minibatch_size = config["minibatch_size"]
        self.num_minibatches = self.batch_size // self.minibatch_size
        # the same three constraints rl_games asserts
        assert self.batch_size % self.minibatch_size == 0, "batch_size (%d) %% minibatch_size (%d) != 0" % (
            self.batch_size, self.minibatch_size)
        assert self.minibatch_size % self.seq_len == 0, "minibatch_size %% seq_len != 0"
        assert self.horizon_length % self.seq_len == 0, "horizon_length %% seq_len != 0"

        self.observation_space = self.env_info["observation_space"]
        self.obs_shape = tuple(self.observation_space.shape)
        self.actions_num = self.env_info["action_space"].shape[0]
        self.actions_low = torch.from_numpy(self.env_info["action_space"].low.copy()).float().to(self.device)
        self.actions_high = torch.from_numpy(self.env_info["action_space"].high.copy()).float().to(self.device)
        self.clip_actions = config.get("clip_actions", True)

        self.model = ModelA2CContinuousLogStd(self.network_params, self.actions_num, self.obs_shape,
                                              self.normalize_value, self.normalize_input).to(self.device)
        self.last_lr = float(config["learning_rate"])
        self.lr = torch.tensor(self.last_lr, device=self.device, dtype=torch.float32)
        self.use_grad_scaler = self.mixed_precision and self.amp_dtype == torch.float16
        self._amp = None            # (loss scale, overflow flag) of the device-side GradScaler: the fp16 fused update only
        if self.use_grad_scaler:
            # fp16 autocast (the reference's mixed_precision: True) needs GradScaler, which drives a torch optimiser
            self._setup_flat_grads()
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self.lr, eps=1e-08,
                                              weight_decay=config.get("weight_decay", 0.0), fused=True)
        else:
            self.optimizer = FlatAdam(self.model.parameters(), self.lr, eps=1e-08,
                                      weight_decay=config.get("weight_decay", 0.0))
            self.flat_grads = self.optimizer.flat_grads
            self.num_params = self.optimizer.num_params
            if self.fused_mixed:
                self.optimizer.enable_lp16_shadow(self.amp_dtype)
                self.model.a2c_network.op_weight_lookup = self.optimizer.shadow_of
                if self.amp_dtype == torch.float16:      # fp16 operands: GradScaler semantics (PY:53 via rl_games)
                    self._amp = self.optimizer.enable_loss_scaling(
                        float(config.get("loss_scale_init", 65536.0)), int(config.get("loss_scale_growth_interval", 2000)))
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.use_grad_scaler)

        self.frame = 0
        self.epoch_num = 0
        self.curr_frames = 0
        self.last_mean_rewards = -100500
        self.mean_rewards = None
        self.algo_observer = algo_observer

        self.experiment_name = config.get("full_experiment_name") or config["name"]
        self.train_dir = config.get("train_dir", "runs")
        self.experiment_dir = os.path.join(self.train_dir, self.experiment_name)
        self.nn_dir = os.path.join(self.experiment_dir, "nn")
        self.summaries_dir = os.path.join(self.experiment_dir, "summaries")
        self.writer = None
        if self.rank == 0 and config.get("write_files", True):
            os.makedirs(self.nn_dir, exist_ok=True)
            self.writer = ScalarLog(self.summaries_dir)
        self._rollout_graph = None
        if self.algo_observer is not None:
            self.algo_observer.after_init(self)

    # ------------------------------------------------------------------ plumbing
    def _setup_flat_grads(self):
        """One contiguous gradient buffer; every ``param.grad`` is a view into it."""
        params = [p for p in self.model.parameters() if p.requires_grad]
        total = sum(p.numel() for p in params)
        self.flat_grads = torch.zeros(total, device=self.device, dtype=torch.float32)
        off = 0
        for p in params:
            p.grad = self.flat_grads[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.num_params = total

    def set_eval(self):
        self.model.eval()

    def set_train(self):
        self.model.train()

    def broadcast_parameters(self):
        """Rank 0's weights and normaliser statistics to every rank (rl_games broadcasts the state dict)."""
        if not self.multi_gpu:
            return
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            dist.broadcast(t.data, 0)
        if self.fused_mixed:
            self.optimizer.refresh_shadow()

    def preprocess_actions(self, actions):
        if self.clip_actions:
            clamped = torch.clamp(actions, -1.0, 1.0)
            d = (self.actions_high - self.actions_low) / 2.0
            m = (self.actions_high + self.actions_low) / 2.0
            return clamped * d + m
        return actions

    def env_step(self, actions):
        obs, rewards, dones, infos = self.vec_env.step(self.preprocess_actions(actions))
        return obs, rewards.unsqueeze(1).to(self.device), dones.to(self.device), infos

    def env_reset(self):
        return self.vec_env.reset()

    def init_tensors(self):
        T, N, dev = self.horizon_length, self.num_actors, self.device
        f32 = dict(device=dev, dtype=torch.float32)
        self.buf = {
            "obses": torch.zeros((T, N) + self.obs_shape, **f32),
            "rewards": torch.zeros((T, N, 1), **f32),
            "values": torch.zeros((T, N, 1), **f32),
            "neglogpacs": torch.zeros((T, N), **f32),
            "dones": torch.zeros((T, N), device=dev, dtype=torch.uint8),
            "actions": torch.zeros((T, N, self.actions_num), **f32),
            "mus": torch.zeros((T, N, self.actions_num), **f32),
            "sigmas": torch.zeros((T, N, self.actions_num), **f32),
        }
        self.current_rewards = torch.zeros((N, 1), **f32)
        self.current_lengths = torch.zeros(N, **f32)
        self.dones = torch.ones(N, device=dev, dtype=torch.uint8)
        self.game_rewards = DeviceAverageMeter(1, self.games_to_track, dev)
        self.game_lengths = DeviceAverageMeter(1, self.games_to_track, dev)
        self.fused_rollout = self.is_cuda and self.use_fused and not self.mixed_precision   # fp32 inference
        if self.fused_rollout:
            # {rew_mean, rew_size, len_mean, len_size, tmp...}: written by vine_rollout_post; the meters are views
            self.meter = torch.zeros(8, **f32)
            self._post_scratch = torch.empty(ROLLOUT_POST_SCRATCH_FLOATS, **f32)
            self.game_rewards.mean, self.game_rewards.current_size = self.meter[0:1], self.meter[1]
            self.game_lengths.mean, self.game_lengths.current_size = self.meter[2:3], self.meter[3]
            self.roll_counter = torch.zeros(1, device=dev, dtype=torch.int64)
            self.head_seed = (int(self.params.get("seed", 0) or 0) + 7919 * (self.rank + 1)) & 0xFFFFFFFFFFFFFFFF
        self.rnn_states = [s.clone() for s in self.model.get_default_rnn_state(N, dev)]
        n_chunks = T // self.seq_len
        # initial LSTM state of every sequence, stored in the DATASET's order ([layer, env, chunk, H] = [1, N * chunks, H]
        # with index env * chunks + chunk): the rollout writes each state once, no permuting copy afterwards
        self.mb_rnn_states = [torch.zeros((1, N, n_chunks, s.shape[-1]), **f32) for s in self.rnn_states]
        self.last_values = torch.zeros((N, 1), **f32)
        self._head_scratch = [torch.zeros((N, self.actions_num), **f32) for _ in range(3)] + [torch.zeros(N, **f32)]
        self._setup_fast_inference()

    # ------------------------------------------------------------------ rollout (R1)
    def get_action_values(self, obs):
        self.model.eval()
        with torch.no_grad():
            return self.model({"is_train": False, "prev_actions": None, "obs": obs, "rnn_states": self.rnn_states})

    def _setup_fast_inference(self):
        """Persistent buffers of the hand-written policy inference of the rollout (``_infer``):
        ``xh`` [N, XW + H] = [MLP output | normalised obs | pad | h] is the operand of ONE gate GEMM against
        ``wcat`` = [w_ih | 0 | w_hh]; operands are bfloat16 in the mixed-precision mode, fp32 otherwise."""
        net = self.model.a2c_network
        self._fast = None
        if not (self.fused_rollout and self.normalize_input and net.activation_is_elu and net.rnn_ln
                and net.rnn_units in (256, 512, 1024) and all(u % 4 == 0 for u in net.units)):
            return
        dev, N, H = self.device, self.num_actors, net.rnn_units
        op = self.amp_dtype if self.rollout_lp16 else torch.float32
        U, F_in = net.units[-1], self.obs_shape[0]
        width = U + (F_in if net.rnn_concat_input else 0)
        XW = (width + 15) // 16 * 16
        f = {"op": op, "U": U, "F": F_in, "XW": XW, "H": H,
             # two copies used alternately: the fused step kernel reads every column of its rows while other
             # workgroups write the new h block, so the h it produces goes to the OTHER buffer
             "xh2": [torch.zeros((N, XW + H), device=dev, dtype=op) for _ in range(2)], "cur": 0,
             "wcat": torch.zeros((4 * H, XW + H), device=dev, dtype=op),
             "acts": [torch.empty((N, u), device=dev, dtype=op) for u in net.units[:-1]],
             "y": torch.empty((N, H), device=dev), "h_tmp": torch.empty((N, H), device=dev),
             "c_tmp": torch.empty((N, H), device=dev)}
        f["x0_sep"] = None if net.rnn_concat_input else torch.empty((N, F_in), device=dev, dtype=op)
        f["ln_in_head"] = H == 256           # vine_policy_head applies the LayerNorm itself (one launch less per step)
        # padded layer-1 weight [units, 32] for the matrix-core kernel (mixed precision, concatenated input)
        f["w1p"] = None
        if (self.rollout_lp16 and net.rnn_concat_input and XW - U == 32
                and fused.linear_elu_mfma_ok(N, net.units[0], 32)):
            f["w1p"] = torch.zeros((net.units[0], 32), device=dev, dtype=op)
        # fp32 matrix-core kernels (vine_mlp3_elu_f32 / vine_lstm_step_f32): the default network at N % 512 == 0
        f["f32_mfma"] = (op == torch.float32 and fused.ROLLOUT_F32_MFMA and net.rnn_concat_input and H == 256 and XW + H == 352
                         and XW - U == 32 and U == 64 and F_in <= 32 and tuple(net.units) == (256, 128, 64) and N % 512 == 0)
        f["f32_split"] = self.rollout_f32_terms if (f["f32_mfma"] and self.rollout_f32_terms in (6, 9)) else 0
        f["wt_f32"] = torch.empty(4 * H * (XW + H), device=dev) if (f["f32_mfma"] and not f["f32_split"]) else None
        # the three bf16 pieces of every recurrent weight, in the split step kernel's fragment order
        f["wt_split"] = (torch.empty(3 * 4 * H * (XW + H), device=dev, dtype=torch.bfloat16) if f["f32_split"] else None)
        f["w1p_f32"] = torch.zeros((net.units[0], 32), device=dev) if f["f32_mfma"] else None     # layer 1, zero-padded
        # the MLP's weights as fragments of bf16 pieces (vine_mlp3_elu_f32_split), rebuilt at every rollout start
        f["mlp_wt_split"] = (torch.empty(288 * 512, device=dev, dtype=torch.bfloat16)
                             if (f["f32_split"] and fused.MLP3_F32_SPLIT and N % 64 == 0) else None)
        f["bias_buf"] = torch.empty(4 * H, device=dev) if f["f32_mfma"] else None                  # b_ih + b_hh of a rollout
        self._fast = f

    def _fast_op_is_fp32(self):
        """True when rollout inference runs on fp32 operands (the reference's rollout precision)."""
        return not self.rollout_lp16

    def _infer_begin(self):
        """Once per rollout: operand copies of the recurrent weights and of h (the update changed the weights)."""
        f, net = self._fast, self.model.a2c_network
        r = net.rnn.rnn
        src = self.optimizer.shadow_of if self.rollout_lp16 else (lambda p: p)
        f["mlp"] = [(src(m.weight), m.bias) for m in net.actor_mlp if isinstance(m, torch.nn.Linear)]
        if f["f32_mfma"] and f["w1p"] is None and f.get("bias_buf") is not None:
            # fp32 rollout (the default): every operand copy of the rollout's start in ONE launch (round 4; they were nine:
            # profiles/r04/iteration_boundary_trace.txt) -- [w_ih | 0 | w_hh], the padded layer-1 weight, b_ih + b_hh, h into
            # the operand buffer, and the step-0 LSTM-state snapshots / slot-0 observation and done flags of the rollout
            cb = fused.CopyBatch()
            cb.add(cb.COPY, f["wcat"][:, :r.weight_ih_l0.shape[1]], r.weight_ih_l0)
            cb.add(cb.COPY, f["wcat"][:, f["XW"]:], r.weight_hh_l0)
            w1 = f["mlp"][0][0]
            cb.add(cb.COPY, f["w1p_f32"][:, :w1.shape[1]], w1)
            cb.add(cb.ADD, f["bias_buf"], r.bias_ih_l0, r.bias_hh_l0)
            f["bias"] = f["bias_buf"]
            f["cur"] = 0
            cb.add(cb.COPY, f["xh2"][0][:, f["XW"]:], self.rnn_states[0][0])
            for extra in getattr(self, "_rollout_start_copies", ()):
                cb.add(cb.COPY, *extra)
            self._rollout_start_copies = ()
            cb.flush(f["wcat"])
            st = torch.cuda.current_stream(self.device).cuda_stream
            if f["f32_split"]:
                fused._check(fused._lib().vine_lstm_tile_weights_split(
                    f["H"], f["XW"] + f["H"], f["wcat"].data_ptr(), f["wcat"].stride(0), f["wt_split"].data_ptr(), st),
                    "vine_lstm_tile_weights_split")
            else:
                fused._check(fused._lib().vine_lstm_tile_weights_f32(
                    f["H"], f["XW"] + f["H"], f["wcat"].data_ptr(), f["wcat"].stride(0), f["wt_f32"].data_ptr(), st),
                    "vine_lstm_tile_weights_f32")
            self._tile_mlp_weights(st)
            return
        f["wcat"][:, :r.weight_ih_l0.shape[1]].copy_(src(r.weight_ih_l0))
        f["wcat"][:, f["XW"]:].copy_(src(r.weight_hh_l0))
        # layer 1 through the matrix-core kernel too: operand = the obs block of xh plus the zero columns behind it
        if f["w1p"] is not None:                      # pad columns stay zero (allocated outside any capture)
            w1 = f["mlp"][0][0]
            f["w1p"][:, :w1.shape[1]].copy_(w1)
        f["bias"] = r.bias_ih_l0 + r.bias_hh_l0
        if f["f32_mfma"]:           # [w_ih | 0 | w_hh] in the step kernel's tile order; layer-1 weight padded to 32 columns
            w1 = f["mlp"][0][0]
            f["w1p_f32"][:, :w1.shape[1]].copy_(w1)
            st = torch.cuda.current_stream(self.device).cuda_stream
            if f["f32_split"]:
                fused._check(fused._lib().vine_lstm_tile_weights_split(
                    f["H"], f["XW"] + f["H"], f["wcat"].data_ptr(), f["wcat"].stride(0), f["wt_split"].data_ptr(), st),
                    "vine_lstm_tile_weights_split")
            else:
                fused._check(fused._lib().vine_lstm_tile_weights_f32(
                    f["H"], f["XW"] + f["H"], f["wcat"].data_ptr(), f["wcat"].stride(0), f["wt_f32"].data_ptr(), st),
                    "vine_lstm_tile_weights_f32")
            self._tile_mlp_weights(st)
        f["cur"] = 0
        f["xh2"][0][:, f["XW"]:].copy_(self.rnn_states[0][0])

    def _tile_mlp_weights(self, st):
        """The MLP weights in the split kernel's fragment order (once per rollout: the update changed them)."""
        f = self._fast
        if f.get("mlp_wt_split") is None or len(f["mlp"]) != 3:
            return
        (W1, _), (W2, _), (W3, _) = f["mlp"]
        fused._check(fused._lib().vine_mlp3_tile_weights_split(W1.data_ptr(), W1.stride(0), W1.shape[1], W2.data_ptr(),
                                                               W2.stride(0), W3.data_ptr(), W3.stride(0),
                                                               f["mlp_wt_split"].data_ptr(), st),
                     "vine_mlp3_tile_weights_split")

    def _infer(self, obs, commit=True):
        """Policy trunk for one step, no autograd: normalise -> [GEMM + bias/ELU kernel] x L -> ONE gate GEMM over
        [x | h] -> LSTM pointwise kernel (state updated in place) -> LayerNorm kernel.  ~10 launches.
        ``commit=False`` leaves the LSTM state untouched (the extra forward for the last values)."""
        lib = fused._lib()
        f, m = self._fast, self.model
        net, rms = m.a2c_network, m.running_mean_std
        N, H, XW = self.num_actors, f["H"], f["XW"]
        bf = int(f["op"] != torch.float32)
        st = torch.cuda.current_stream(self.device).cuda_stream
        xh, xh_next = f["xh2"][f["cur"]], f["xh2"][f["cur"] ^ 1]
        x0 = f["x0_sep"] if f["x0_sep"] is not None else xh[:, f["U"]:f["U"] + f["F"]]
        hp_ptr = (xh_next.data_ptr() + xh_next.element_size() * XW) if commit else None
        n_mlp = len(f["mlp"])
        mlp3 = (bf and fused.MLP3 and f["w1p"] is not None and f["x0_sep"] is None and n_mlp == 3 and N % 64 == 0
                and f["U"] == 64 and f["F"] <= 32 and obs.is_contiguous() and obs.dtype == torch.float32
                and tuple(W.shape for W, _ in f["mlp"][1:]) == ((128, 256), (64, 128)) and f["mlp"][0][0].shape[0] == 256)
        f32k = bool(f["f32_mfma"]) and obs.is_contiguous() and obs.dtype == torch.float32 and n_mlp == 3
        if not f32k and getattr(self, "_pending_fin", None) is not None:
            # a deferred meter fold with no fp32 MLP launch to ride on (a caller changed the inference path between the
            # post-step kernel and this forward): run it as the one-workgroup launch it used to be -- never drop it
            fused._check(lib.vine_rollout_finalize(*self._pending_fin, st), "vine_rollout_finalize")
            self._pending_fin = None
        if f32k:
            # fp32 (the reference's rollout precision) on the matrix cores: normalisation + the three layers in one launch
            (W1, b1), (W2, b2), (W3, b3) = f["mlp"]
            fin = getattr(self, "_pending_fin", None)      # the previous step's meter fold rides on workgroup 0 (round 4)
            self._pending_fin = None
            if f.get("mlp_wt_split") is not None:
                # exact products from bf16 pieces, as the LSTM step below (four waves share the rows, split the units)
                fused._check(lib.vine_mlp3_elu_f32_split(N, xh.data_ptr(), xh.stride(0), obs.data_ptr(), f["F"],
                                                         rms.running_mean.data_ptr(), rms.running_var.data_ptr(),
                                                         float(rms.epsilon), 5.0, f["mlp_wt_split"].data_ptr(),
                                                         b1.data_ptr(), b2.data_ptr(), b3.data_ptr(), 1.0,
                                                         f["f32_split"] | (fused.MLP3_F32_SPLIT_RT << 8) | (int(fused.ROLLOUT_F32_DUAL) << 16),
                                                         *(fin if fin is not None else (None, 0.0, None, None, 0)), st),
                             "vine_mlp3_elu_f32_split")
            else:
                fused._check(lib.vine_mlp3_elu_f32_fin(N, xh.data_ptr(), xh.stride(0), obs.data_ptr(), f["F"],
                                                       rms.running_mean.data_ptr(), rms.running_var.data_ptr(),
                                                       float(rms.epsilon), 5.0, f["w1p_f32"].data_ptr(), 32, b1.data_ptr(),
                                                       256, W2.data_ptr(), W2.stride(0), b2.data_ptr(), 128, W3.data_ptr(),
                                                       W3.stride(0), b3.data_ptr(), 64, 1.0,
                                                       *(fin if fin is not None else (None, 0.0, None, None, 0)), st),
                             "vine_mlp3_elu_f32_fin")
        elif mlp3:
            # observation normalisation and the whole MLP in ONE launch: the kernel normalises the raw observations
            # itself, writes them (bf16, zero-padded) into the LSTM operand's observation block and carries the
            # activations through the three layers in registers (the intermediate activations are not needed here)
            (W1, b1), (W2, b2), (W3, b3) = f["mlp"]
            fused._check(lib.vine_mlp3_elu_mfma(N, xh.data_ptr() + 2 * f["U"], xh.stride(0), obs.data_ptr(), f["F"],
                                                rms.running_mean.data_ptr(), rms.running_var.data_ptr(),
                                                float(rms.epsilon), 5.0, f["w1p"].data_ptr(), b1.data_ptr(), 256,
                                                W2.data_ptr(), W2.stride(0), b2.data_ptr(), 128, W3.data_ptr(), W3.stride(0),
                                                b3.data_ptr(), 64, 1.0, None, None, xh.data_ptr(), xh.stride(0), st),
                         "vine_mlp3_elu_mfma")
        elif not f32k:
            fused._check(lib.vine_normalize_obs(N, f["F"], obs.data_ptr(), rms.running_mean.data_ptr(),
                                                rms.running_var.data_ptr(), float(rms.epsilon), 5.0, x0.data_ptr(),
                                                x0.stride(0), bf, st), "vine_normalize_obs")
        x = x0
        for i, (W, b) in enumerate(f["mlp"] if not (mlp3 or f32k) else ()):
            out = xh if i == n_mlp - 1 else f["acts"][i]
            if i == 0 and f["w1p"] is not None:
                fused._check(lib.vine_linear_elu_mfma(N, W.shape[0], 32, xh.data_ptr() + 2 * f["U"], xh.stride(0),
                                                      f["w1p"].data_ptr(), 32, b.data_ptr(), 1.0, out.data_ptr(),
                                                      out.stride(0), st), "vine_linear_elu_mfma")
            elif bf and fused.linear_elu_mfma_ok(N, W.shape[0], W.shape[1]):
                fused._check(lib.vine_linear_elu_mfma(N, W.shape[0], W.shape[1], x.data_ptr(), x.stride(0), W.data_ptr(),
                                                      W.stride(0), b.data_ptr(), 1.0, out.data_ptr(), out.stride(0), st),
                             "vine_linear_elu_mfma")
            else:
                z = fused._mm(x, W.t())
                fused._check(lib.vine_bias_elu(N, z.shape[1], z.data_ptr(), b.data_ptr(), 1.0, out.data_ptr(),
                                               out.stride(0), bf, st), "vine_bias_elu")
            x = out
        h32, c = self.rnn_states[0][0], self.rnn_states[1][0]
        h_out, c_out = (h32, c) if commit else (f["h_tmp"], f["c_tmp"])
        Kx = XW + H
        if f32k and f["f32_split"]:
            # gate GEMM over [x | h] + the cell update: fp32 operands, every product exact from bf16 pieces (9 pairs)
            fused._check(lib.vine_lstm_step_f32_split(N, H, Kx, xh.data_ptr(), xh.stride(0), f["wt_split"].data_ptr(),
                                                      f["bias"].data_ptr(), c.data_ptr(), h_out.data_ptr(), H,
                                                      c_out.data_ptr(), hp_ptr, xh_next.stride(0),
                                                      f["f32_split"] | (int(fused.rollout_f32_nsplit(N)) << 16), st),
                         "vine_lstm_step_f32_split")
            gates = None
        elif f32k:
            # gate GEMM over [x | h] + the cell update, fp32 on the matrix cores
            fused._check(lib.vine_lstm_step_f32(N, H, Kx, xh.data_ptr(), xh.stride(0), f["wt_f32"].data_ptr(),
                                                f["bias"].data_ptr(), c.data_ptr(), h_out.data_ptr(), H, c_out.data_ptr(),
                                                hp_ptr, xh_next.stride(0), st), "vine_lstm_step_f32")
            gates = None
        elif bf and N % 64 == 0 and Kx in (128, 256, 288, 320, 352, 384, 512) and H % 16 == 0:
            # gate GEMM over [x | h] fused with the pointwise update on the matrix cores
            fused._check(lib.vine_lstm_step_mfma(
                N, H, Kx, xh.data_ptr(), xh.stride(0), None, 0, 0, f["wcat"].data_ptr(), f["wcat"].stride(0), None, 4 * H,
                f["bias"].data_ptr(), c.data_ptr(), None, 0, h_out.data_ptr(), H, c_out.data_ptr(), None,
                hp_ptr, None, 0, Kx, st), "vine_lstm_step_mfma")
            gates = None
        else:
            gates = fused._mm(xh, f["wcat"].t())
        if gates is not None:
            fused._check(lib.vine_lstm_cell_forward(
                N, H, gates.data_ptr(), 4 * H, None, f["bias"].data_ptr(), c.data_ptr(), None, 0, h_out.data_ptr(), H,
                c_out.data_ptr(), None, hp_ptr, None, 0, bf, XW + H, st), "vine_lstm_cell_forward")
        if commit:
            f["cur"] ^= 1
        if f["ln_in_head"]:              # the policy-head kernel normalises (reads the state before rollout_post clears it)
            return h_out
        y = f["y"]
        fused._check(lib.vine_layernorm_forward(N, H, h_out.data_ptr(), net.layer_norm.weight.data_ptr(),
                                                net.layer_norm.bias.data_ptr(), float(net.layer_norm.eps), y.data_ptr(),
                                                None, None, st), "vine_layernorm_forward")
        return y

    def _rollout_body_fused(self):
        """Same sequence as ``_rollout_body`` with the pointwise work in three hand-written kernels per step:
        policy head (mu/value GEMV + sampling + neglogp + value un-normalisation, written straight into the rollout
        buffers), the fused env step, and the post-step bookkeeping (reward shaping, bootstrap, dones, episode
        accumulators, windowed means, LSTM-state zeroing).  ~30 launches per step instead of ~125."""
        lib = fused._lib()
        buf, m = self.buf, self.model
        net = m.a2c_network
        env = getattr(self.vec_env, "env", self.vec_env)
        N, A, H = self.num_actors, self.actions_num, net.rnn_units
        st = torch.cuda.current_stream(self.device).cuda_stream
        vms = m.value_mean_std if self.normalize_value else None
        # the head kernel reads the value normaliser's float64 statistics itself (vine_policy_head_rms: the module's own
        # mean.float(), sqrt(var.float() + eps)); the two-float form stays for other callers of vine_policy_head
        head_rms = (vms is not None and vms.running_mean.numel() == 1 and vms.running_mean.dtype == torch.float64
                    and os.environ.get("VINE_POLICY_HEAD_RMS", "1") != "0")
        if vms is not None and not head_rms:
            vmean = vms.running_mean.float()
            vstd = torch.sqrt(vms.running_var.float() + vms.epsilon)
        else:
            vmean = vstd = None
        gamma_b = float(self.gamma) if self.value_bootstrap else 0.0
        obs = self.obs

        def head(x, n_slot, value_out, mu_out, sigma_out, act_out, nlp_out):
            # x: the LayerNorm output, or (fused inference with H == 256) the raw LSTM output: the head kernel then
            # applies the LayerNorm itself
            ln = net.layer_norm if (fast and self._fast["ln_in_head"]) else None
            if head_rms:
                fused._check(lib.vine_policy_head_rms(
                    N, A, H, x.data_ptr(), net.mu.weight.data_ptr(), net.mu.bias.data_ptr(), net.value.weight.data_ptr(),
                    net.value.bias.data_ptr(), net.sigma.data_ptr(), vms.running_mean.data_ptr(), vms.running_var.data_ptr(),
                    float(vms.epsilon), self.head_seed, self.roll_counter.data_ptr(), mu_out.data_ptr(), sigma_out.data_ptr(),
                    value_out.data_ptr(), act_out.data_ptr(), nlp_out.data_ptr(),
                    ln.weight.data_ptr() if ln is not None else None, ln.bias.data_ptr() if ln is not None else None,
                    float(ln.eps) if ln is not None else 0.0, st), "vine_policy_head_rms")
                return
            fused._check(lib.vine_policy_head(
                N, A, H, x.data_ptr(), net.mu.weight.data_ptr(), net.mu.bias.data_ptr(), net.value.weight.data_ptr(),
                net.value.bias.data_ptr(), net.sigma.data_ptr(), vmean.data_ptr() if vmean is not None else None,
                vstd.data_ptr() if vstd is not None else None, int(self.normalize_value), self.head_seed,
                self.roll_counter.data_ptr(), mu_out.data_ptr(), sigma_out.data_ptr(), value_out.data_ptr(),
                act_out.data_ptr(), nlp_out.data_ptr(), ln.weight.data_ptr() if ln is not None else None,
                ln.bias.data_ptr() if ln is not None else None, float(ln.eps) if ln is not None else 0.0, st),
                "vine_policy_head")

        def trunk(o):
            x = m.norm_obs(o)
            y = net.actor_mlp(x)
            if net.rnn_concat_input:
                y = torch.cat([y, x], dim=1)
            y, states = net.rnn.forward_flat(y, self.rnn_states, None, 1)
            if net.rnn_ln:
                y = net.layer_norm(y)
            return y.contiguous(), states

        fast = getattr(self, "_fast", None) is not None
        batched = fast and self._fast["f32_mfma"] and self._fast["w1p"] is None and os.environ.get("VINE_ROLLOUT_COPYBATCH", "1") != "0"
        if fast:
            if batched:
                # the step-0 snapshots of the LSTM state ride in _infer_begin's one launch
                self._rollout_start_copies = tuple((mb_s[0, :, 0], s_[0]) for s_, mb_s in zip(self.rnn_states, self.mb_rnn_states))
                self._slot0_batched = (buf["obses"].dtype == torch.float32 and obs.dtype == torch.float32 and obs.is_contiguous()
                                       and obs.dim() == 2 and buf["dones"].dtype == torch.uint8 and self.dones.dtype == torch.uint8
                                       and N % 16 == 0)
                if self._slot0_batched:      # slot 0 of the observation / done-flag buffers too (flags moved as 4-byte words)
                    self._rollout_start_copies += ((buf["obses"][0], obs),
                                                   (buf["dones"][0].view(torch.float32), self.dones.view(torch.float32)))
            self._infer_begin()
        # the env writes the next observation, and the post-step kernel the next done flags, straight into the rollout
        # buffers (30 small copy nodes fewer per iteration) when the task offers step_into on this device
        direct = (hasattr(env, "step_into") and buf["obses"].dtype == torch.float32 and buf["obses"][0].is_contiguous()
                  and buf["dones"].dtype == torch.uint8 and self.dones.dtype == torch.uint8
                  and buf["obses"].device == env.rew_buf.device and tuple(buf["obses"].shape[1:]) == tuple(obs.shape))
        if direct and getattr(self, "_obs_last", None) is None:
            self._obs_last = torch.empty_like(obs)
        h_op_stride = self._fast["XW"] + H if fast else 0
        h_op_bf16 = int(fast and self._fast["op"] != torch.float32)
        defer_fin = bool(fast and self._fast["f32_mfma"] and len(self._fast["mlp"]) == 3 and obs.is_contiguous()
                         and obs.dtype == torch.float32 and os.environ.get("VINE_ROLLOUT_FIN_RIDE", "1") != "0")
        post_blocks = int(lib.vine_rollout_post_blocks(N)) if defer_fin else 0
        self._pending_fin = None
        # Round 5: policy head + env step + bookkeeping as ONE launch (vine_step_rollout: a rollout step is 3 launches instead
        # of 5) where the task runs the four-lanes-per-env kernel and the head has the default shape; VINE_ROLLOUT_STEP_FUSED=0
        # / `rollout_step_fused: False` = the three launches
        roll_blocks = int(env.rollout_step_blocks()) if hasattr(env, "rollout_step_blocks") else 0
        step_fused = bool(defer_fin and direct and head_rms and fast and self._fast["ln_in_head"] and A == 2 and H == 256
                          and h_op_bf16 == 0 and roll_blocks > 0 and roll_blocks * 3 <= self._post_scratch.numel()
                          and self.config.get("rollout_step_fused", True)
                          and os.environ.get("VINE_ROLLOUT_STEP_FUSED", "1") != "0")
        if step_fused:
            if getattr(self, "_roll_head", None) is None:
                self._roll_head = (torch.empty(3 * H, device=self.device), torch.empty(3, device=self.device))
                self._roll_args = []
            hw, hc = self._roll_head
            fused._check(lib.vine_rollout_head_prep(net.layer_norm.weight.data_ptr(), net.layer_norm.bias.data_ptr(),
                                                    net.mu.weight.data_ptr(), net.mu.bias.data_ptr(), net.value.weight.data_ptr(),
                                                    net.value.bias.data_ptr(), hw.data_ptr(), hc.data_ptr(), st),
                         "vine_rollout_head_prep")
            self._roll_args.clear()
        self.rollout_step_launches = 3 if step_fused else (5 if defer_fin else 6)
        for n in range(self.horizon_length):
            if n % self.seq_len == 0 and not (batched and n == 0):
                if batched:          # both states in one launch
                    cb = fused.CopyBatch()
                    for s_, mb_s in zip(self.rnn_states, self.mb_rnn_states):
                        cb.add(cb.COPY, mb_s[0, :, n // self.seq_len], s_[0])
                    cb.flush(self.rnn_states[0])
                else:
                    for s_, mb_s in zip(self.rnn_states, self.mb_rnn_states):
                        mb_s[:, :, n // self.seq_len].copy_(s_)
            if fast:
                y = self._infer(obs)
            else:
                y, states = trunk(obs)
                self.rnn_states = [states[0].contiguous(), states[1].contiguous()]
            if (not direct or n == 0) and not (batched and n == 0 and self._slot0_batched):
                # (direct: the env and the post-step kernel wrote slot n themselves, one step ago)
                buf["obses"][n].copy_(obs)
                buf["dones"][n].copy_(self.dones)
            last = n + 1 == self.horizon_length
            if step_fused:
                ra = abi.RolloutArgs()
                ra.y, ra.hw, ra.hc, ra.logstd = y.data_ptr(), hw.data_ptr(), hc.data_ptr(), net.sigma.data_ptr()
                ra.value_mean, ra.value_var = vms.running_mean.data_ptr(), vms.running_var.data_ptr()
                ra.ln_eps, ra.value_eps = float(net.layer_norm.eps), float(vms.epsilon)
                ra.seed, ra.counter = int(self.head_seed), self.roll_counter.data_ptr()
                ra.mu_out, ra.sigma_out, ra.value_out = buf["mus"][n].data_ptr(), buf["sigmas"][n].data_ptr(), buf["values"][n].data_ptr()
                ra.action_out, ra.neglogp_out = buf["actions"][n].data_ptr(), buf["neglogpacs"][n].data_ptr()
                ra.reward_shift, ra.reward_scale, ra.gamma_bootstrap = float(self.reward_shift), float(self.reward_scale), gamma_b
                ra.shaped_out = buf["rewards"][n].data_ptr()
                ra.dones_out = (self.dones if last else buf["dones"][n + 1]).data_ptr()
                ra.cur_rewards, ra.cur_lengths = self.current_rewards.data_ptr(), self.current_lengths.data_ptr()
                ra.h_state, ra.c_state = self.rnn_states[0].data_ptr(), self.rnn_states[1].data_ptr()
                # the operand copy of h that the NEXT step reads (the buffer _infer just switched to)
                ra.h_op = self._fast["xh2"][self._fast["cur"]].data_ptr() + 4 * self._fast["XW"]
                ra.h_op_stride = h_op_stride
                ra.partial = self._post_scratch.data_ptr()
                self._roll_args.append(ra)          # (the launch copies the struct; kept for the record only)
                obs = env.step_rollout_into(ra, self._obs_last if last else buf["obses"][n + 1])
                self._pending_fin = (self.meter.data_ptr(), float(self.games_to_track), self.roll_counter.data_ptr(),
                                     self._post_scratch.data_ptr(), roll_blocks)
                continue
            head(y, n, buf["values"][n], buf["mus"][n], buf["sigmas"][n], buf["actions"][n], buf["neglogpacs"][n])
            # the env kernel clamps to +-clipActions itself (vec_task.py:333); with the [-1, 1] action space
            # rl_games' preprocess_actions (clamp + affine rescale) is the identity on top of that
            if direct:
                # next observation and next done flags straight into their rollout-buffer slots (the last ones into
                # the tensors the next iteration starts from)
                obs = env.step_into(buf["actions"][n], self._obs_last if last else buf["obses"][n + 1])
                dones_dst = self.dones if last else buf["dones"][n + 1]
            else:
                obs_d, rewards, dones, infos = self.vec_env.step(buf["actions"][n])
                obs = obs_d["obs"]
                dones_dst = self.dones
            h_op_ptr = ((self._fast["xh2"][self._fast["cur"]].data_ptr()
                         + self._fast["xh2"][0].element_size() * self._fast["XW"]) if fast else None)
            if defer_fin:
                # per-env pass only; the one-workgroup fold of its episode sums (meters, rollout counter) rides in the next
                # inference's MLP launch -- always one more follows: the next step's, or the last-values forward below
                fused._check(lib.vine_rollout_post_defer(
                    N, H, env.rew_buf.data_ptr(), env.reset_buf.data_ptr(), env.timeout_buf.data_ptr(),
                    buf["values"][n].data_ptr(), float(self.reward_shift), float(self.reward_scale), gamma_b,
                    buf["rewards"][n].data_ptr(), dones_dst.data_ptr(), self.current_rewards.data_ptr(),
                    self.current_lengths.data_ptr(), self.rnn_states[0].data_ptr(), self.rnn_states[1].data_ptr(),
                    h_op_ptr, h_op_stride, h_op_bf16, self._post_scratch.data_ptr(), st), "vine_rollout_post_defer")
                self._pending_fin = (self.meter.data_ptr(), float(self.games_to_track), self.roll_counter.data_ptr(),
                                     self._post_scratch.data_ptr(), post_blocks)
                continue
            fused._check(lib.vine_rollout_post(
                N, H, env.rew_buf.data_ptr(), env.reset_buf.data_ptr(), env.timeout_buf.data_ptr(),
                buf["values"][n].data_ptr(), float(self.reward_shift), float(self.reward_scale), gamma_b,
                buf["rewards"][n].data_ptr(), dones_dst.data_ptr(), self.current_rewards.data_ptr(),
                self.current_lengths.data_ptr(), self.rnn_states[0].data_ptr(), self.rnn_states[1].data_ptr(),
                self.meter.data_ptr(), float(self.games_to_track), self.roll_counter.data_ptr(),
                # the operand copy of h that the NEXT step reads (the buffer _infer just switched to)
                h_op_ptr, h_op_stride, h_op_bf16, self._post_scratch.data_ptr(), st), "vine_rollout_post")
        self.obs = obs
        y = self._infer(obs, commit=False) if fast else trunk(obs)[0]
        scratch = self._head_scratch
        head(y, 0, self.last_values, scratch[0], scratch[1], scratch[2], scratch[3])

    def _rollout_body(self):
        """``play_steps_rnn``: horizon x {store; policy forward (eval); env step; shape reward; bootstrap;
        zero the LSTM state of finished envs}; then last values.  Fixed shapes, no host sync."""
        buf = self.buf
        obs = self.obs
        for n in range(self.horizon_length):
            if n % self.seq_len == 0:
                for s, mb_s in zip(self.rnn_states, self.mb_rnn_states):
                    mb_s[:, :, n // self.seq_len].copy_(s)
            res = self.get_action_values(obs)
            self.rnn_states = list(res["rnn_states"])
            buf["obses"][n].copy_(obs)
            buf["dones"][n].copy_(self.dones)
            buf["actions"][n].copy_(res["actions"])
            buf["neglogpacs"][n].copy_(res["neglogpacs"])
            buf["values"][n].copy_(res["values"])
            buf["mus"][n].copy_(res["mus"])
            buf["sigmas"][n].copy_(res["sigmas"])
            obs_d, rewards, dones, infos = self.env_step(res["actions"])
            obs = obs_d["obs"]
            shaped = (rewards + self.reward_shift) * self.reward_scale
            if self.value_bootstrap and "time_outs" in infos:
                shaped = shaped + self.gamma * res["values"] * infos["time_outs"].to(self.device).unsqueeze(1).float()
            buf["rewards"][n].copy_(shaped)
            self.dones = dones.to(torch.uint8)
            not_dones = self._episode_bookkeeping(rewards, dones)
            self.rnn_states = [s * not_dones.view(1, -1, 1) for s in self.rnn_states]
        self.obs = obs
        self.last_values.copy_(self.get_action_values(obs)["values"])

    def _episode_bookkeeping(self, rewards, dones):
        """The running episode return / length accumulators of one rollout step (play_steps, common_agent.py:293-306 in the
        reference's in-tree text; golden F10): add the step, hand the finished episodes to the two meters, clear them.
        Returns ``not_dones`` [N].  (``vine_rollout_post`` is the same on the device.)"""
        self.current_rewards += rewards
        self.current_lengths += 1
        done_f = dones.float()
        self.game_rewards.update(self.current_rewards, done_f)
        self.game_lengths.update(self.current_lengths.unsqueeze(1), done_f)
        not_dones = 1.0 - done_f
        self.current_rewards = self.current_rewards * not_dones.unsqueeze(1)
        self.current_lengths = self.current_lengths * not_dones
        return not_dones

    def play_steps_rnn(self):
        body = self._rollout_body_fused if self._can_fuse_rollout() else self._rollout_body
        if self.use_graphs and not getattr(getattr(self.vec_env, "env", self.vec_env), "graph_capturable", True):
            self.use_graphs = False              # e.g. MAT_FILE replay: host-indexed state writes every step
        if self.use_graphs:
            try:
                self._play_graphed(body)
                self.graph_status["rollout"] = "graph"
            except RuntimeError as err:          # capture refused (driver/runtime state): run eagerly from now on
                self.graph_status["rollout"] = "eager (capture refused)"
                if self._rollout_graph is not None and getattr(self, "_graph_replayed", False):
                    raise
                print("hipGraph capture of the rollout failed (%s); continuing with eager launches" % str(err)[:200])
                self.use_graphs = False
                self._rollout_graph = None
                torch.cuda.synchronize(self.device)
                if getattr(self, "_g_in", None) is not None:
                    self._load_live(self._g_in)      # the state the capture started from (restored after warm-up)
                body()
        else:
            body()
        buf = self.buf
        batch = self._assemble_dataset_fused()
        if batch is not None:
            return batch
        if (self.is_cuda and self.use_fused and buf["dones"].dtype == torch.uint8 and self.dones.dtype == torch.uint8
                and buf["values"].is_contiguous() and buf["rewards"].is_contiguous()):
            # GAE as one kernel (one env per lane, reverse scan in registers) instead of 16 x 6 small launches
            mb_advs, mb_returns = torch.empty_like(buf["rewards"]), torch.empty_like(buf["rewards"])
            fused._check(fused._lib().vine_gae(
                self.horizon_length, self.num_actors, buf["rewards"].data_ptr(), buf["values"].data_ptr(),
                buf["dones"].data_ptr(), self.last_values.data_ptr(), self.dones.data_ptr(), float(self.gamma),
                float(self.tau), mb_advs.data_ptr(), mb_returns.data_ptr(),
                torch.cuda.current_stream(self.device).cuda_stream), "vine_gae")
        else:
            fdones = self.dones.float()
            mb_fdones = buf["dones"].float()
            mb_advs = discount_values(self.gamma, self.tau, fdones, self.last_values, mb_fdones, buf["values"],
                                      buf["rewards"])
            mb_returns = mb_advs + buf["values"]
        # [T, N, ...] -> [N * T, ...]: straight into the update graphs' static dataset tensors when they exist (one
        # transposing copy per tensor instead of a transposing copy + a copy into the static storage)
        st = getattr(self, "_ds_static", None) if (self.use_graphs and self.is_cuda) else None
        ds_key = {"obses": "obs", "actions": "actions", "neglogpacs": "old_logp_actions", "mus": "mu", "sigmas": "sigma",
                  "dones": "dones"}

        def flat(k):
            src = buf[k]
            dst = st.get(ds_key.get(k)) if st is not None else None
            sz = src.size()
            if dst is not None and dst.is_contiguous() and dst.dtype == src.dtype and \
                    tuple(dst.shape) == (sz[0] * sz[1], *sz[2:]):
                dst.view(sz[1], sz[0], *sz[2:]).copy_(src.transpose(0, 1))
                return dst
            return swap_and_flatten01(src)
        batch = {k: flat(k) for k in ("obses", "actions", "neglogpacs", "values", "mus", "sigmas", "dones")}
        batch["returns"] = swap_and_flatten01(mb_returns)
        batch["played_frames"] = self.batch_size
        states = []
        for mb_s in self.mb_rnn_states:          # already in dataset order: a view, no copy
            states.append(mb_s.view(mb_s.size()[0], mb_s.size()[1] * mb_s.size()[2], mb_s.size()[3]))
        batch["rnn_states"] = states
        return batch

    def _assemble_dataset_fused(self):
        """Rollout buffers -> dataset in three launches (``vine_dataset_assemble``: GAE, the value normaliser's two updates
        and normalisations, the advantage normalisation and every [T, N, .] -> [N T, .] transposition), written straight into
        the persistent dataset tensors the optimiser steps read.  Returns the ``batch`` marker ``prepare_dataset`` accepts,
        or None when the shapes / dtypes are not covered (CPU, stock path, N % 64)."""
        buf, T, N = self.buf, self.horizon_length, self.num_actors
        if not (self.is_cuda and self.use_fused and os.environ.get("VINE_DATASET_FUSED", "1") != "0" and N % 64 == 0
                and buf["dones"].dtype == torch.uint8 and self.dones.dtype == torch.uint8
                and all(buf[k].dtype == torch.float32 and buf[k].is_contiguous()
                        for k in ("obses", "actions", "neglogpacs", "values", "mus", "sigmas", "rewards"))
                and buf["dones"].is_contiguous() and buf["values"].shape[-1] == 1):
            return None
        vms = self.model.value_mean_std if self.normalize_value else None
        st = getattr(self, "_ds_static", None)
        if st is None or not st.get("_assembled", False):
            n = N * T
            dev = self.device
            f32 = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
            st = {"old_values": f32(n, 1), "returns": f32(n, 1), "advantages": f32(n),
                  "old_logp_actions": f32(n), "actions": f32(n, self.actions_num), "obs": f32(n, *buf["obses"].shape[2:]),
                  "dones": torch.empty(n, device=dev, dtype=torch.uint8), "mu": f32(n, self.actions_num),
                  "sigma": f32(n, self.actions_num),
                  "rnn_states": [mb_s.view(mb_s.size()[0], mb_s.size()[1] * mb_s.size()[2], mb_s.size()[3])
                                 for mb_s in self.mb_rnn_states],     # already in dataset order: views, no copy
                  "_assembled": True}
            if st["obs"].dim() != 2:
                return None
            self._ds_static = st
            self._ds_scratch = torch.empty((N + 255) // 256 * 6 + 4, device=dev, dtype=torch.float64)
            self._vms_pending = torch.zeros(3, device=dev, dtype=torch.float64)
            if hasattr(self, "_upd_graphs"):
                self._upd_graphs.clear()      # (captured steps read the dataset at fixed addresses)
        jobs = [(buf["obses"], st["obs"], st["obs"].shape[1], 4), (buf["actions"], st["actions"], self.actions_num, 4),
                (buf["neglogpacs"], st["old_logp_actions"], 1, 4), (buf["mus"], st["mu"], self.actions_num, 4),
                (buf["sigmas"], st["sigma"], self.actions_num, 4), (buf["dones"], st["dones"], 1, 1)]
        import ctypes as C
        k = len(jobs)
        rc = fused._lib().vine_dataset_assemble(
            T, N, buf["rewards"].data_ptr(), buf["values"].data_ptr(), buf["dones"].data_ptr(), self.last_values.data_ptr(),
            self.dones.data_ptr(), float(self.gamma), float(self.tau),
            vms.running_mean.data_ptr() if vms is not None else None, vms.running_var.data_ptr() if vms is not None else None,
            vms.count.data_ptr() if vms is not None else None, float(vms.epsilon) if vms is not None else 0.0,
            int(vms is not None), int(bool(self.normalize_advantage)), st["old_values"].data_ptr(), st["returns"].data_ptr(),
            st["advantages"].data_ptr(), k, (C.c_void_p * k)(*[j[0].data_ptr() for j in jobs]),
            (C.c_void_p * k)(*[j[1].data_ptr() for j in jobs]), (C.c_int32 * k)(*[int(j[2]) for j in jobs]),
            (C.c_int32 * k)(*[j[3] for j in jobs]), self._ds_scratch.data_ptr(), self._vms_pending.data_ptr(),
            torch.cuda.current_stream(self.device).cuda_stream)
        if rc == abi.ERR_UNSUPPORTED:
            return None
        fused._check(rc, "vine_dataset_assemble")
        # the finished dataset (normalised series included): ``prepare_dataset`` adopts it as it is -- also on another agent
        batch = {k: v for k, v in st.items() if not k.startswith("_")}
        batch.update(assembled=True, played_frames=self.batch_size)
        if vms is not None:
            batch["vms_pending"] = self._vms_pending      # committed by prepare_dataset, where rl_games updates the module
        return batch

    def _can_fuse_rollout(self):
        if not getattr(self, "fused_rollout", False):
            return False
        env = getattr(self.vec_env, "env", self.vec_env)
        lo, hi = self.env_info["action_space"].low, self.env_info["action_space"].high
        return (hasattr(env, "timeout_buf") and env.rew_buf.is_cuda and float(lo.min()) == -1.0 and float(hi.max()) == 1.0
                and float(getattr(env, "clip_actions", 1.0)) <= 1.0)

    def _play_graphed(self, body):
        """Capture the whole rollout once, then replay it.  Live state that the captured code rebinds
        (obs, dones, LSTM state, episode accumulators) is kept in static tensors copied in and out."""
        if self._rollout_graph is None:
            fused_mode = body == self._rollout_body_fused
            keep = (lambda t: t) if fused_mode else (lambda t: t.clone())   # meters are views of self.meter when fused
            self._g_in = [self.obs.clone(), self.dones.clone(), [s.clone() for s in self.rnn_states],
                          self.current_rewards.clone(), self.current_lengths.clone(),
                          keep(self.game_rewards.mean), keep(self.game_rewards.current_size),
                          keep(self.game_lengths.mean), keep(self.game_lengths.current_size)]
            flat = [self._g_in[0], self._g_in[1]] + self._g_in[2] + self._g_in[3:5]
            backup = [t.clone() for t in flat]
            snap = self._snapshot_env()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):     # warm-up outside capture (lazy inits of libraries, autotuning)
                self._load_live(self._g_in)
                body()
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._restore_env(snap)
            for t, b in zip(flat, backup):
                t.copy_(b)
            torch.cuda.synchronize(self.device)
            self._rollout_graph = torch.cuda.CUDAGraph()
            self._load_live(self._g_in)
            # thread_local: RCCL's watchdog thread may touch the HIP runtime while this thread captures
            with torch.cuda.graph(self._rollout_graph, capture_error_mode="thread_local"):
                self._load_live(self._g_in)
                body()
                self._g_out = [self.obs, self.dones, self.rnn_states, self.current_rewards, self.current_lengths,
                               self.game_rewards.mean, self.game_rewards.current_size, self.game_lengths.mean,
                               self.game_lengths.current_size]
        with _Range("rollout_graph_replay"):
            self._rollout_graph.replay()
        self._graph_replayed = True
        # carry the outputs over to the static inputs of the next replay
        o = self._g_out
        pairs = [(self._g_in[0], o[0]), (self._g_in[1], o[1])] + list(zip(self._g_in[2], o[2]))
        pairs += [(self._g_in[i], o[i]) for i in range(3, 9)]
        for a, b in pairs:
            if a.data_ptr() != b.data_ptr():      # quantities the kernels update in place need no carry-over
                a.copy_(b)
        self._load_live(self._g_in)

    def _load_live(self, g):
        self.obs, self.dones = g[0], g[1]
        self.rnn_states = list(g[2])
        self.current_rewards, self.current_lengths = g[3], g[4]
        self.game_rewards.mean, self.game_rewards.current_size = g[5], g[6]
        self.game_lengths.mean, self.game_lengths.current_size = g[7], g[8]

    def _snapshot_env(self):
        env = getattr(self.vec_env, "env", self.vec_env)
        snap = {"state": env.state.clone(), "reset": env.reset_buf.clone(), "progress": env.progress_buf.clone(),
                "step": env.step_count, "rng": torch.cuda.get_rng_state(self.device)}
        if getattr(self, "fused_rollout", False):
            snap["agent"] = [t.clone() for t in (self.meter, self.roll_counter)]
        return snap

    def _restore_env(self, snap):
        env = getattr(self.vec_env, "env", self.vec_env)
        torch.cuda.synchronize(self.device)
        env.state.copy_(snap["state"]); env.reset_buf.copy_(snap["reset"]); env.progress_buf.copy_(snap["progress"])
        env.step_count = snap["step"]
        torch.cuda.set_rng_state(snap["rng"], self.device)
        if "agent" in snap:
            self.meter.copy_(snap["agent"][0])
            self.roll_counter.copy_(snap["agent"][1])

    # ------------------------------------------------------------------ dataset (R5)
    def prepare_dataset(self, batch):
        assembled = bool(batch.get("assembled", False))
        if assembled:
            # _assemble_dataset_fused: GAE, value / advantage normalisation and the transpositions are done (and
            # value_mean_std has taken its two updates); the tensors are this agent's persistent ones, or another agent's
            ds = {k: batch[k] for k in ("old_values", "old_logp_actions", "advantages", "returns", "actions", "obs", "dones",
                                        "rnn_states", "mu", "sigma")}
            pend = batch.get("vms_pending")
            if pend is not None and self.normalize_value:
                vms = self.model.value_mean_std       # value_mean_std(values); value_mean_std(returns): one launch
                torch._foreach_copy_([vms.running_mean, vms.running_var, vms.count],
                                     [pend[0:1].view_as(vms.running_mean), pend[1:2].view_as(vms.running_var), pend[2].view_as(vms.count)])
        else:
            returns, values = batch["returns"], batch["values"]
            advantages = returns - values
            if self.normalize_value:
                self.model.value_mean_std.train()
                values = self.model.value_mean_std(values)
                returns = self.model.value_mean_std(returns)
                self.model.value_mean_std.eval()
            advantages = torch.sum(advantages, dim=1)
            if self.normalize_advantage:
                advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
            ds = {"old_values": values, "old_logp_actions": batch["neglogpacs"], "advantages": advantages,
                  "returns": returns, "actions": batch["actions"], "obs": batch["obses"],
                  "dones": batch["dones"], "rnn_states": batch["rnn_states"],
                  "mu": batch["mus"], "sigma": batch["sigmas"]}
        if (self.use_graphs and self.is_cuda) or assembled:
            # persistent storage: the captured optimiser steps read their minibatch slices at fixed addresses
            st = getattr(self, "_ds_static", None)
            if st is None:
                # (the per-sequence LSTM states are persistent rollout buffers in dataset order already: used in place)
                st = self._ds_static = {k: (list(v) if isinstance(v, list) else v.clone()) for k, v in ds.items()}
            else:
                for k, v in ds.items():
                    if isinstance(v, list):
                        for dst, src in zip(st[k], v):
                            if dst.data_ptr() != src.data_ptr():     # (else: written in place by play_steps_rnn)
                                dst.copy_(src)
                    elif st[k].data_ptr() != v.data_ptr():
                        st[k].copy_(v)
            self.dataset = {k: v for k, v in st.items() if not k.startswith("_")}
        else:
            ds["mu"], ds["sigma"] = ds["mu"].clone(), ds["sigma"].clone()
            self.dataset = ds

    def get_minibatch(self, idx):
        """Contiguous slices, no shuffling (rl_games PPODataset._get_item_rnn)."""
        games = self.minibatch_size // self.seq_len
        gstart, gend = idx * games, (idx + 1) * games
        start, end = gstart * self.seq_len, gend * self.seq_len
        mb = {k: v[start:end] for k, v in self.dataset.items() if k != "rnn_states"}
        mb["rnn_states"] = [s[:, gstart:gend, :].contiguous() for s in self.dataset["rnn_states"]]
        mb["range"] = (start, end)
        return mb

    # ------------------------------------------------------------------ update (R3, R4, R6)
    def _fused_grad_half(self, mb, obs_n=None, stats_out=None, norm_stats=None):
        """Forward, ONE kernel for the whole PPO loss and its gradient w.r.t. the head outputs, hand-written backward
        into the flat gradient block; the minibatch KL is parked next to the gradients so that one all-reduce
        averages both.  -> (stats[8], mu, logstd)"""
        self._amp_covered = False
        out = self._fused_grad_half_body(mb, obs_n, stats_out, norm_stats)
        if self._amp is not None:
            # loss-scaled fp16 backward: unless every parameter gradient ended in the overflow-checked column-sum launch
            # the optimiser looks for non-finite gradients itself before it steps
            self.optimizer.check_grads = not self._amp_covered
        return out

    def _fused_grad_half_body(self, mb, obs_n, stats_out, norm_stats=None):
        net = self.model.a2c_network
        hb = (net.mu.bias.grad, net.value.bias.grad)
        ext = all(g is not None and g.is_cuda for g in hb)     # head-bias gradients straight from the loss kernel
        batch_dict = {"obs": mb["obs"] if obs_n is None else obs_n, "obs_is_normalized": obs_n is not None,
                      "rnn_states": mb["rnn_states"], "seq_length": self.seq_len, "dones": mb["dones"],
                      "head_bias_external": ext}
        if norm_stats is not None:
            batch_dict["obs_norm_stats"] = norm_stats
        pack = None
        if ext and fused.HEADS_LOSS and self.fused_mixed:
            # LayerNorm + heads + loss + their backward inside the trunk node (one launch instead of three)
            pack = fused.ppo_loss_pack(net.sigma, mb["actions"], mb["old_logp_actions"], mb["advantages"], mb["old_values"],
                                       mb["returns"], mb["mu"], mb["sigma"], self.e_clip, self.clip_value, self.critic_coef,
                                       self.entropy_coef, self.bounds_loss_coef or 0.0, hb, kl_out=self.optimizer.aux[0:1],
                                       logstd_grad=net.sigma.grad, update_old=True, stats_out=stats_out, amp=self._amp)
            batch_dict["loss_pack"] = pack
        mu, value, logstd, _, heads = self.model.forward_raw(batch_dict)
        if pack is not None and heads is not None and heads.grad_fn is not None and getattr(heads.grad_fn, "loss_fused", None):
            torch.autograd.backward([heads], [heads.detach()])      # the gradient handed over is ignored by the node
            self._amp_covered = bool(pack.get("amp_covered", False))
            return pack["stats"], mu.detach(), logstd.detach()
        g_mu, g_val, g_ls, stats = fused.ppo_loss_fused(
            mu, logstd, value, mb["actions"], mb["old_logp_actions"], mb["advantages"], mb["old_values"], mb["returns"],
            mb["mu"], mb["sigma"], self.e_clip, self.clip_value, self.critic_coef, self.entropy_coef,
            self.bounds_loss_coef or 0.0, heads=heads, head_bias_grads=hb if (ext and heads is not None) else None,
            # KL next to the gradients (it rides in the all-reduce), log-sigma gradient into its slot and the dataset's
            # mu / sigma refreshed in place: all by the loss kernels themselves
            kl_out=self.optimizer.aux[0:1], logstd_grad=net.sigma.grad, update_old=True, stats_out=stats_out, amp=self._amp)
        # the gradient block was left zeroed by the last Adam step
        if heads is not None:
            torch.autograd.backward([heads], [g_mu])            # g_mu is the [n, A+1] gradient of [mu | value]
        else:
            torch.autograd.backward([mu, value], [g_mu, g_val])
        return stats, mu.detach(), logstd.detach()

    def calc_gradients_fused(self, mb):
        """GPU path (fp32 or bf16-operand GEMMs): same arithmetic as ``calc_gradients``."""
        stats, mu_d, logstd_d = self._fused_grad_half(mb)
        sigma_d = torch.exp(logstd_d).expand_as(mu_d)
        kl = stats[4]
        if self.multi_gpu and not self.use_grad_scaler:
            self._kl_in_comm = True
            kl = self.optimizer.aux[0]
        self.truncate_gradients_and_step()
        return stats[0], stats[1], stats[3], kl, stats[2], mu_d, sigma_d

    def calc_gradients(self, mb):
        if self.is_cuda and not self.mixed_precision and self.use_fused:
            return self.calc_gradients_fused(mb)
        batch_dict = {"is_train": True, "prev_actions": mb["actions"], "obs": mb["obs"], "rnn_states": mb["rnn_states"],
                      "seq_length": self.seq_len, "dones": mb["dones"]}
        with torch.autocast(device_type=self.device.type, dtype=self.amp_dtype, enabled=self.mixed_precision):
            res = self.model(batch_dict)
            action_log_probs, values, entropy = res["prev_neglogp"], res["values"], res["entropy"]
            mu, sigma = res["mus"], res["sigmas"]
            a_loss = actor_loss(mb["old_logp_actions"], action_log_probs, mb["advantages"], self.e_clip).mean()
            c_loss = critic_loss(mb["old_values"], values, self.e_clip, mb["returns"], self.clip_value).mean()
            b_loss = bound_loss(mu).mean() if self.bounds_loss_coef is not None else torch.zeros((), device=self.device)
            entropy = entropy.mean()
            loss = (a_loss + 0.5 * c_loss * self.critic_coef - entropy * self.entropy_coef
                    + b_loss * (self.bounds_loss_coef or 0.0))
        self.flat_grads.zero_()
        self.scaler.scale(loss).backward()
        self.truncate_gradients_and_step()
        with torch.no_grad():
            kl = policy_kl(mu.detach().float(), sigma.detach().float(), mb["mu"], mb["sigma"])
        return a_loss.detach(), c_loss.detach(), entropy.detach(), kl, b_loss.detach(), mu.detach(), sigma.detach()

    def truncate_gradients_and_step(self):
        if self.multi_gpu:
            buf = self.optimizer.comm_buffer if not self.use_grad_scaler else self.flat_grads
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)                 # RCCL over xGMI, in place, 1.63 MB
        if not self.use_grad_scaler:
            if self.truncate_grads:
                if self.multi_gpu:
                    self.flat_grads.div_(self.rank_size)
                # the fp16 fused update leaves loss-scaled gradients here (unscaled inside the Adam kernel): clip against
                # the unscaled norm, as rl_games does after scaler.unscale_
                self.optimizer.clip_grad_norm_(self.grad_norm)
                self.optimizer.step()
            else:
                self.optimizer.step(grad_scale=1.0 / self.rank_size)   # /world folded into the Adam kernel
            return
        if self.multi_gpu:
            self.flat_grads.div_(self.rank_size)
        if self.truncate_grads:
            self.scaler.unscale_(self.optimizer)
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_norm)
        self.scaler.step(self.optimizer)
        self.scaler.update()

    def _lr_schedule_args(self):
        """Arguments of the AdaptiveScheduler folded into the Adam launch (graphed update: `legacy` schedule, KL parked
        next to the gradients by the loss kernel, summed over the ranks by the gradient all-reduce)."""
        return (self.optimizer.aux[0:1], 1.0 / self.rank_size, self.kl_threshold, self.min_lr, self.max_lr)

    def update_lr_from_kl(self, kl):
        """AdaptiveScheduler of rl_games on a device scalar (common_agent.py:217-221 shows the call site)."""
        if self.multi_gpu and not getattr(self, "_kl_in_comm", False):
            kl = kl.clone()
            dist.all_reduce(kl, op=dist.ReduceOp.SUM)
        self._kl_in_comm = False
        if self.is_cuda:
            rc = fused._lib().vine_adaptive_lr(self.lr.data_ptr(), kl.contiguous().data_ptr(), 1.0 / self.rank_size,
                                               self.kl_threshold, self.min_lr, self.max_lr,
                                               torch.cuda.current_stream(self.device).cuda_stream)
            assert rc == 0
            return
        kl = kl / self.rank_size
        lr = self.lr
        down = torch.clamp(lr / 1.5, min=self.min_lr)
        up = torch.clamp(lr * 1.5, max=self.max_lr)
        new = torch.where(kl > 2.0 * self.kl_threshold, down, torch.where(kl < 0.5 * self.kl_threshold, up, lr))
        self.lr.copy_(new)

    def train_epoch(self):
        self._epochs_run = getattr(self, "_epochs_run", 0) + 1
        t_play = time.time()
        # On the GPU the two phases are timed with events on the stream, NOT with a host synchronisation between them: a
        # sync after the rollout left the device idle while the host issued the ~40 small launches that turn the rollout
        # buffers into the dataset (0.4 ms of an 11.7 ms iteration); now the host queues them while the rollout graph runs.
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if self.is_cuda else None
        if ev:
            ev[0].record(torch.cuda.current_stream(self.device))
        self.set_eval()
        with torch.no_grad(), _Range("rollout"):
            batch = self.play_steps_rnn()
        if ev:
            ev[1].record(torch.cuda.current_stream(self.device))
        play_time = time.time() - t_play
        t_upd = time.time()
        upd_range = _Range("update")
        upd_range.__enter__()
        self.set_train()
        if self.fused_mixed:
            # the Adam kernel keeps the bf16 operand copies current; this catches every other writer of the
            # parameters (checkpoint restore, broadcast, user code) at the cost of one 0.8 MB copy per iteration
            self.optimizer.refresh_shadow()
        self.curr_frames = batch.pop("played_frames")
        self.prepare_dataset(batch)
        # loss statistics of every optimiser step on the device, in the layout of the loss kernel's statistics:
        # rows of [a_loss, c_loss, b_loss, entropy, kl, (loss, 0, 0)]
        n_steps = self.mini_epochs_num * self.num_minibatches
        if getattr(self, "_stat_rows", None) is None or self._stat_rows.shape[0] != n_steps:
            self._stat_rows = torch.zeros((n_steps, 8), device=self.device, dtype=torch.float32)
        rows = self._stat_rows
        graphed = self._update_graphs_usable()
        kl_global = False
        # one rank: nothing has to happen between the optimiser steps of a mini-epoch, so all of them replay as ONE
        # hipGraph (4 graph launches per iteration instead of 64: the launch gaps were ~6 % of the update)
        # several ranks: the same, with the gradient all-reduce of every step captured INSIDE the graph, when RCCL accepts
        # a capture on this stack (probed once, collectively: ``_collective_capture_ok``); otherwise one graph per step with
        # the collective issued eagerly between them
        scope = os.environ.get("VINE_UPD_GRAPH", "all")
        whole_epoch = (graphed and scope in ("all", "epoch")
                       and (not self.multi_gpu or self._collective_capture_ok()))
        # round 5: ALL mini-epochs of the update as one graph (one replay per iteration instead of four: the device sat
        # idle for ~9 us in front of every replay, and each was followed by a copy of its loss statistics); the
        # learning-rate schedule and the step counter live in the Adam launch, so nothing happens on the host in between
        all_done = whole_epoch and scope == "all" and self._update_epoch_graphed(rows, n_epochs=self.mini_epochs_num)
        for mini_ep in range(0 if not all_done else self.mini_epochs_num, self.mini_epochs_num):
            nb = self.num_minibatches
            if whole_epoch and self._update_epoch_graphed(rows[mini_ep * nb:(mini_ep + 1) * nb]):
                if self.normalize_input:
                    self.model.running_mean_std.eval()
                continue
            whole_epoch = False            # capture refused: single rank: the flag set by the failure sends the steps below
                                           # eager; several ranks: they fall back to one graph per step (eager collective)
            graphed = graphed and not getattr(self, "_update_graphs_failed", False)
            step_graphed = 0
            for i in range(self.num_minibatches):
                row = mini_ep * self.num_minibatches + i
                if graphed and self._update_step_graphed(i):
                    step_graphed += 1
                    continue
                self._flush_pending_adam()     # (a capture was refused mid-update: the previous step's Adam is still owed)
                if step_graphed:
                    rows[row - step_graphed:row].copy_(self._step_stats[i - step_graphed:i])
                    step_graphed = 0
                mb = self.get_minibatch(i)
                a_loss, c_loss, entropy, kl, b_loss, cmu, csigma = self.calc_gradients(mb)
                start, end = mb["range"]
                self.dataset["mu"][start:end] = cmu.float()          # dataset.update_mu_sigma
                self.dataset["sigma"][start:end] = csigma.float()
                in_comm = getattr(self, "_kl_in_comm", False)
                kl_global = kl_global or in_comm
                rows[row, :5].copy_(torch.stack([a_loss, c_loss, b_loss, entropy, kl / self.rank_size if in_comm else kl]))
                if self.is_adaptive_lr and self.schedule_type == "legacy":
                    self.update_lr_from_kl(kl)
            if step_graphed:      # the graphs of this mini-epoch left their statistics in _step_stats: one copy
                rows[(mini_ep + 1) * nb - step_graphed:(mini_ep + 1) * nb].copy_(self._step_stats[nb - step_graphed:nb])
            if self.is_adaptive_lr and self.schedule_type == "standard":
                ep = rows[mini_ep * self.num_minibatches:(mini_ep + 1) * self.num_minibatches, 4].mean()
                self._kl_in_comm = kl_global
                self.update_lr_from_kl(ep * self.rank_size if kl_global else ep)
            if self.normalize_input:
                self.model.running_mean_std.eval()   # statistics are updated during the first mini-epoch only
        self._flush_pending_adam()                   # the last step's Adam (per-step graphs defer it into the next graph)
        if ev:
            ev[2].record(torch.cuda.current_stream(self.device))
            # Round 4: the host does not wait for THIS iteration here -- it waits for the previous one (whose events it then
            # reads), so that the next iteration's rollout graph is already queued when this update ends.  Waiting here
            # left the device idle for ~125 us per iteration (the sync, the host's bookkeeping, the next graph launch:
            # profiles/r04/iteration_boundary_trace.txt).  The host never runs more than one iteration ahead; the first
            # iteration, and `sync_each_iteration: True` / VINE_SYNC_EACH_ITER=1, wait as before.
            prev = getattr(self, "_ev_prev", None)
            if prev is None or self.config.get("sync_each_iteration", False) or os.environ.get("VINE_SYNC_EACH_ITER") == "1":
                torch.cuda.synchronize(self.device)
                timed = ev
            else:
                prev[2].synchronize()
                timed = prev
            self._ev_prev = ev
        upd_range.__exit__()
        update_time = time.time() - t_upd
        if ev:
            gp, gu = timed[0].elapsed_time(timed[1]) * 1e-3, timed[1].elapsed_time(timed[2]) * 1e-3
            if timed is ev:
                # (the split of the iteration as the device saw it; the sum is the host's wall time of the iteration)
                total = play_time + update_time
                play_time = total * gp / max(gp + gu, 1e-12)
                update_time = total - play_time
            else:
                # deferred wait: the host's time inside this call says nothing (it only queues work) -- report the DEVICE
                # times of the previous iteration (rollout start -> rollout end -> update end: the same quantity in steady state)
                play_time, update_time = gp, gu
        m = rows.mean(0)
        stats = {"a_loss": m[0], "c_loss": m[1], "entropy": m[3], "kl": m[4], "b_loss": m[2]}
        return play_time, update_time, stats

    # ------------------------------------------------------------------ update as hipGraphs
    def _update_graphs_usable(self):
        """The optimiser step is ~105 launches of 5-30 us each: with bf16 GEMM operands (and nearly so in fp32) the
        host cannot issue them as fast as the GPU retires them.  From the second iteration on (the first one runs
        eagerly and warms every lazily initialised handle) the steps are replayed from hipGraphs.  One rank: a whole
        mini-epoch per graph.  Several ranks: the RCCL all-reduce of every step is issued outside any capture, and ONE
        graph runs from behind one all-reduce to the next -- [Adam + learning-rate schedule of step s - 1] + [normalisation
        (+ running-statistics update in the first mini-epoch) + forward + loss + backward of step s] -- so an iteration is
        32 graph replays + 32 collectives + one trailing Adam graph (round 2: two graphs per step, 64 + 32)."""
        return (self.use_graphs and self.is_cuda and self.use_fused and not self.mixed_precision
                and not self.truncate_grads and self.is_adaptive_lr and self.schedule_type == "legacy"
                and getattr(self, "_epochs_run", 0) > 1 and not getattr(self, "_update_graphs_failed", False))

    def _capture_agreed(self, ok):
        """Whether EVERY rank captured its graph: the ranks must take the same path through the optimiser step (graph
        replays around one eager all-reduce, or eager launches), so a refusal on one rank sends all of them eager.
        One MIN all-reduce, only at the (rare) moments a capture is attempted -- all ranks attempt it at the same
        optimiser step, so the collective is matched."""
        if not self.multi_gpu:
            return ok
        flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    def _flush_pending_adam(self):
        """Per-step graphs defer a step's Adam launch into the NEXT step's graph (it then sits behind the all-reduce with
        no graph boundary of its own); whoever leaves that chain -- the end of the update, a refused capture -- runs it."""
        if not getattr(self, "_adam_pending", False):
            return
        self._adam_pending = False
        rec = self._upd_graphs.get("tail") if hasattr(self, "_upd_graphs") else None
        if rec is None and not getattr(self, "_update_graphs_failed", False):
            try:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize(self.device)
                with torch.cuda.graph(g, pool=self._upd_pool, capture_error_mode="thread_local"):
                    self.optimizer.step(grad_scale=1.0 / self.rank_size, lr_schedule=self._lr_schedule_args())
                rec = self._upd_graphs["tail"] = {"G": g}
            except RuntimeError:
                rec = None
        if rec is not None:
            with _Range("update_graph_tail_adam_lr"):
                rec["G"].replay()
        else:
            self.optimizer.step(grad_scale=1.0 / self.rank_size, lr_schedule=self._lr_schedule_args())

    def _update_step_graphed(self, i):
        # the running-statistics update of the observation normaliser (first mini-epoch only) is part of the graph:
        # its kernels use fixed-order two-stage sums and no memsets, so they replay faithfully
        if getattr(self, "_update_graphs_failed", False):
            return False
        lead = bool(getattr(self, "_adam_pending", False))      # does the graph start with the previous step's Adam?
        key = (i, bool(self.normalize_input and self.model.running_mean_std.training), lead)
        rec = self._upd_graphs.get(key) if hasattr(self, "_upd_graphs") else None
        if rec is None:
            err = None
            try:
                rec = self._capture_update_step(i, key, lead)
            except RuntimeError as e:        # capture refused: nothing has executed
                err = e
            if not self._capture_agreed(rec is not None):
                # this and every later step run eagerly, on every rank
                print("hipGraph capture of the optimiser step failed on %s (%s); continuing with eager launches"
                      % ("this rank" if err is not None else "another rank", str(err)[:200]))
                if rec is not None:
                    self._upd_graphs.pop(key, None)
                self._update_graphs_failed = True
                self._kl_in_comm = False
                self.graph_status["update"] = "eager (capture refused)"
                torch.cuda.synchronize(self.device)
                return False
        with _Range("update_graph_adam_forward_loss_backward"):
            rec["G"].replay()
        if self.multi_gpu:
            with _Range("grad_all_reduce"):
                dist.all_reduce(self.optimizer.comm_buffer, op=dist.ReduceOp.SUM)      # gradients + KL, RCCL over xGMI
        self._adam_pending = True
        self.graph_status["update"] = "graph (per optimiser step: Adam of the previous step + forward / backward; all-reduce between graphs)"
        return True

    def _agree_bounded(self, ok, timeout_s):
        """``_capture_agreed`` with a deadline: True / False = the MIN over the ranks, None = no answer in time."""
        if not self.multi_gpu:
            return bool(ok)
        flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
        work = dist.all_reduce(flag, op=dist.ReduceOp.MIN, async_op=True)
        if self.is_cuda:
            # (the work handle of an NCCL collective completes when its kernel has run: poll it, and the stream behind it)
            ev = torch.cuda.Event()
            work.wait()                                   # orders the current stream behind the collective; does not block the host
            ev.record(torch.cuda.current_stream(self.device))
            if not wait_bounded(ev.query, timeout_s):
                return None
        elif not wait_bounded(work.is_completed, timeout_s):
            return None
        return bool(flag.item() > 0.5)

    def _abort_job(self, message):
        """Last resort of the collective probe: say why and leave with a non-zero code WITHOUT running destructors that
        would wait for the stuck stream (never an exec: the process has initialised the GPU)."""
        print("FATAL (rank %d): %s" % (self.rank, message), flush=True)
        os._exit(PROBE_EXIT_CODE)

    def _collective_capture_ok(self):
        """May the RCCL gradient all-reduce live INSIDE a captured graph?  Decided once per agent, by all ranks together
        (``collective_probe_protocol``): a throw-away graph holding one ``all_reduce`` of a small tensor on this rank's
        stream is captured; only if EVERY rank captured it is it replayed (twice, each wait bounded by
        ``collective_probe_timeout_s`` / VINE_COLLECTIVE_PROBE_TIMEOUT, default 10 s) and its result checked (SUM over the
        ranks).  Any refusal -- config ``collective_in_graph: False``, env ``VINE_COLLECTIVE_IN_GRAPH=0``, a capture error,
        a wrong sum, a timeout, on ANY rank -- keeps the collective eager between per-step graphs.
        ``self.collective_capture`` records the outcome (bench line: ``collective_in_graph``)."""
        state = getattr(self, "_coll_capture", None)
        if state is not None:
            return state
        want = bool(self.config.get("collective_in_graph", True)) and os.environ.get("VINE_COLLECTIVE_IN_GRAPH", "1") != "0"
        timeout_s = float(os.environ.get("VINE_COLLECTIVE_PROBE_TIMEOUT", self.config.get("collective_probe_timeout_s", 10.0)))
        expect = float(self.rank_size * (self.rank_size + 1) // 2)

        def capture():
            if not want:
                return False, "disabled", None
            if not self.is_cuda:
                return False, "no device", None
            if dist.get_backend() != "nccl":
                return False, "backend %s cannot be captured" % dist.get_backend(), None     # (gloo rehearsals)
            try:
                probe = torch.full((256,), float(dist.get_rank() + 1), device=self.device)
                keep = probe.clone()
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    dist.all_reduce(probe, op=dist.ReduceOp.SUM)
                return True, "captured", (g, probe, keep)
            except RuntimeError as err:
                torch.cuda.synchronize(self.device)
                return False, "capture refused: %s" % str(err)[:160], None

        def replay(handle):
            g, probe, keep = handle
            stream = torch.cuda.current_stream(self.device)
            for _ in range(2):
                probe.copy_(keep)
                g.replay()
                ev = torch.cuda.Event()
                ev.record(stream)
                if not wait_bounded(ev.query, timeout_s):
                    return "timeout"
                if not bool((probe == expect).all()):
                    return "wrong sum"
            return "ok"

        ok, why = collective_probe_protocol(capture, replay, lambda v: self._agree_bounded(v, timeout_s), self._abort_job)
        self._coll_capture = ok
        self.collective_capture = {"in_graph": ok, "probe": why, "timeout_s": timeout_s}
        return ok

    def _update_epoch_graphed(self, rows_out, n_epochs=1):
        """All optimiser steps of one mini-epoch (``n_epochs`` > 1: of that many, i.e. of the whole update) as one graph
        replay.  Several ranks: the RCCL all-reduce of every step is
        a node of that graph (``_collective_capture_ok``); if the capture is refused on any rank all of them fall back to
        one graph per step with the collective between the graphs (``_update_step_graphed``)."""
        key = ("epoch" if n_epochs == 1 else "all", bool(self.normalize_input and self.model.running_mean_std.training))
        rec = self._upd_graphs.get(key) if hasattr(self, "_upd_graphs") else None
        if rec is None:
            err = None
            try:
                rec = self._capture_update_epoch(key, n_epochs, rows_out)
            except RuntimeError as e:
                err = e
            if not self._capture_agreed(rec is not None):
                if rec is not None:
                    self._upd_graphs.pop(key, None)
                torch.cuda.synchronize(self.device)
                if self.multi_gpu:
                    print("hipGraph capture of the mini-epoch with its all-reduces failed on %s (%s); one graph per optimiser "
                          "step, collective between the graphs" % ("this rank" if err is not None else "another rank", str(err)[:200]))
                    self._coll_capture = False
                    self.collective_capture = {"in_graph": False, "probe": "mini-epoch capture refused: %s" % str(err)[:160]}
                    return False
                print("hipGraph capture of the mini-epoch failed (%s); continuing with eager launches" % str(err)[:200])
                self._update_graphs_failed = True
                self._kl_in_comm = False
                self.graph_status["update"] = "eager (capture refused)"
                return False
        with _Range("update_graph_mini_epoch"):
            rec["G"].replay()
        if self.multi_gpu and not rec.get("replayed"):
            # the first replay of a graph that holds real collectives: bounded wait (a hang inside a replayed RCCL node
            # would otherwise be silent); afterwards the graph has proven itself and replays are not waited for
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            t = float((getattr(self, "collective_capture", None) or {}).get("timeout_s", 10.0))
            if not wait_bounded(ev.query, t):
                self._abort_job("the first replay of the mini-epoch graph (gradient all-reduces captured) did not finish "
                                "within %.0f s; relaunch with VINE_COLLECTIVE_IN_GRAPH=0 (all-reduce between per-step graphs)" % t)
            rec["replayed"] = True
        if rec["stats"].data_ptr() != rows_out.data_ptr():      # (the whole-update graph writes the iteration's rows itself)
            rows_out.copy_(rec["stats"])
        if n_epochs > 1 and self.normalize_input:
            self.model.running_mean_std.eval()                   # (as the per-mini-epoch loop leaves the module)
        per = "iteration" if n_epochs > 1 else "mini-epoch"
        self.graph_status["update"] = ("graph (1 per %s, all-reduces captured)" % per if self.multi_gpu
                                       else "graph (1 per %s)" % per)
        return True

    def _rms_updates_ahead(self, nb):
        """The observation normaliser's training-mode updates of all ``nb`` optimiser steps of the first mini-epoch in three
        launches at its head (``vine_rms_update_multi``) instead of two in front of every step: a step's update depends on
        the minibatch's observations and on the statistics after the previous step only, never on the weights.  Returns
        (mean [nb, F], var [nb, F]) -- step i normalises with row i, bit-identical to updating in front of it -- or None
        when the per-step path has to run (``VINE_RMS_AHEAD=0``, shapes the kernels do not cover)."""
        if os.environ.get("VINE_RMS_AHEAD", "1") == "0" or not self.normalize_input:
            return None
        rms, obs = self.model.running_mean_std, self.dataset["obs"]
        net = self.model.a2c_network
        first = obs[:self.minibatch_size]
        if not (obs.shape[0] == nb * self.minibatch_size and obs.is_contiguous() and rms.training and rms._use_kernels(first)
                and net.op_weight_lookup is not None and net.trunk_supported(first, self.seq_len)):
            return None
        return rms.update_kernels_multi(obs, nb)

    def _capture_update_epoch(self, key, n_epochs=1, rows_out=None):
        if not hasattr(self, "_upd_graphs"):
            self._upd_graphs = {}
            self._upd_pool = torch.cuda.graph_pool_handle()
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        pool = None if os.environ.get("VINE_UPD_POOL") == "separate" else self._upd_pool
        nb = self.num_minibatches
        # one mini-epoch: a buffer of the graph's own, copied into the iteration's rows after every replay; the whole update:
        # the iteration's rows themselves (a persistent buffer of the agent: self._stat_rows)
        stats_all = (rows_out if n_epochs > 1 else
                     torch.zeros((nb, 8), device=self.device, dtype=torch.float32))     # (never allocate zeros in capture)
        keep = []
        rms_training = bool(key[1])
        try:
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                for ep in range(n_epochs):
                    ahead = self._rms_updates_ahead(nb) if (rms_training and ep == 0) else None
                    for i in range(nb):
                        mb = self.get_minibatch(i)
                        # statistics straight into their row, step counter and learning-rate schedule inside the Adam launch
                        stats, mu_d, _logstd_d = self._fused_grad_half(
                            mb, stats_out=stats_all[ep * nb + i],
                            norm_stats=None if ahead is None else (ahead[0][i], ahead[1][i]))
                        if self.multi_gpu:
                            # gradients + KL + overflow flag, SUM over the ranks (RCCL over xGMI): a node of the graph
                            dist.all_reduce(self.optimizer.comm_buffer, op=dist.ReduceOp.SUM)
                        self.optimizer.step(grad_scale=1.0 / self.rank_size, lr_schedule=self._lr_schedule_args())
                        keep.append((mb, stats, mu_d))
                    keep.append(ahead)
                    if n_epochs > 1 and self.normalize_input:
                        self.model.running_mean_std.eval()      # statistics are updated during the first mini-epoch only
        finally:
            if n_epochs > 1 and self.normalize_input and rms_training:
                self.model.running_mean_std.train()              # (the caller switches it after the replay)
        rec = {"G": g, "stats": stats_all, "keep": keep}
        self._upd_graphs[key] = rec
        return rec

    def _capture_update_step(self, i, key, lead):
        if not hasattr(self, "_upd_graphs"):
            self._upd_graphs = {}
            self._upd_pool = torch.cuda.graph_pool_handle()
        if getattr(self, "_step_stats", None) is None or self._step_stats.shape[0] != self.num_minibatches:
            # this rank's [a_loss, c_loss, b_loss, entropy, kl, loss, 0, 0] of every step of a mini-epoch: written by the
            # graphs, copied into the iteration's rows once per mini-epoch
            self._step_stats = torch.zeros((self.num_minibatches, 8), device=self.device, dtype=torch.float32)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        pool = None if os.environ.get("VINE_UPD_POOL") == "separate" else self._upd_pool
        with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
            if lead:
                # the step behind the previous all-reduce: the KL next to the gradients is the sum over the ranks by
                # now, the schedule runs inside the Adam launch (the dataset's mu / sigma slices were refreshed by the
                # loss kernel of that step)
                self.optimizer.step(grad_scale=1.0 / self.rank_size, lr_schedule=self._lr_schedule_args())
            mb = self.get_minibatch(i)
            stats, mu_d, _logstd_d = self._fused_grad_half(mb, stats_out=self._step_stats[i])
        rec = {"G": g, "stats": stats, "keep": (mb, stats, mu_d)}
        self._upd_graphs[key] = rec
        return rec


    # ------------------------------------------------------------------ checkpoint (aux: resume)
    def get_full_state_weights(self):
        return {"model": self.model.state_dict(), "epoch": self.epoch_num, "optimizer": self.optimizer.state_dict(),
                "frame": self.frame, "last_mean_rewards": self.last_mean_rewards, "last_lr": float(self.lr)}

    def save(self, fn):
        state = self.get_full_state_weights()
        torch.save(state, fn + ".pth")
        return fn + ".pth"

    def restore(self, fn):
        ckpt = torch.load(fn, map_location=self.device, weights_only=False)
        self.model.load_state_dict(ckpt["model"])
        self.epoch_num = ckpt.get("epoch", 0)
        self.frame = ckpt.get("frame", 0)
        self.last_mean_rewards = ckpt.get("last_mean_rewards", -100500)
        if "optimizer" in ckpt:
            self.optimizer.load_state_dict(ckpt["optimizer"])
            for group in self.optimizer.param_groups:
                group["lr"] = self.lr
        if "last_lr" in ckpt:
            self.lr.fill_(ckpt["last_lr"])
        if self.fused_mixed:
            self.optimizer.refresh_shadow()

    # ------------------------------------------------------------------ main loop
    def train(self):
        self.init_tensors()
        self.obs = self.env_reset()["obs"].to(self.device)
        self.broadcast_parameters()
        total_time = 0.0
        t_last = time.perf_counter()
        while True:
            self.epoch_num += 1
            epoch_num = self.epoch_num
            play_time, update_time, stats = self.train_epoch()
            # rl_games logs the host's wall clock.  With the deferred wait train_epoch hands back the DEVICE's split of the
            # previous iteration; the scalars this loop reads below synchronise every iteration anyway, so the wall time
            # between two returns is one whole iteration including the host gap: the split is rescaled to it (as
            # bench_support does), and `fps total`, performance/* and total_time (the x-axis of rewards/time) are wall-clock
            # figures again
            t_now = time.perf_counter()
            wall, t_last = t_now - t_last, t_now
            if play_time + update_time > 0:
                k = wall / (play_time + update_time)
                play_time, update_time = play_time * k, update_time * k
            sum_time = play_time + update_time
            total_time += sum_time
            curr_frames = self.curr_frames * self.rank_size
            self.frame += curr_frames
            should_exit = False
            if self.rank == 0:
                self.last_lr = float(self.lr)
                if self.print_stats:   # the reference's console line (common_agent.py:145-148)
                    print(f"fps step and policy inference: {curr_frames / play_time:.0f} "
                          f"fps total: {curr_frames / sum_time:.0f} epoch: {epoch_num}/{self.max_epochs}")
                self.write_stats(total_time, epoch_num, play_time, update_time, stats, curr_frames)
                if self.algo_observer is not None:
                    env = getattr(self.vec_env, "env", self.vec_env)
                    self.algo_observer.process_infos({k: v for k, v in getattr(env, "extras", {}).items()}, None)
                    self.algo_observer.after_print_stats(self.frame, epoch_num, total_time)
                if float(self.game_rewards.current_size) > 0:
                    mean_rewards = float(self.game_rewards.get_mean()[0])
                    mean_lengths = float(self.game_lengths.get_mean()[0])
                    self.mean_rewards = mean_rewards
                    if self.writer:
                        for suffix, x in (("step", self.frame), ("iter", epoch_num), ("time", total_time)):
                            self.writer.add_scalar("rewards/" + suffix, mean_rewards, x)
                            self.writer.add_scalar("episode_lengths/" + suffix, mean_lengths, x)
                    checkpoint_name = self.config["name"] + "_ep_" + str(epoch_num) + "_rew_" + str(mean_rewards)
                    if self.writer and self.save_freq > 0 and epoch_num % self.save_freq == 0 \
                            and mean_rewards <= self.last_mean_rewards:
                        self.save(os.path.join(self.nn_dir, "last_" + checkpoint_name))
                    if mean_rewards > self.last_mean_rewards and epoch_num >= self.save_best_after:
                        print("saving next best rewards: ", mean_rewards)
                        self.last_mean_rewards = mean_rewards
                        if self.writer:
                            self.save(os.path.join(self.nn_dir, self.config["name"]))
                        if self.last_mean_rewards > self.config.get("score_to_win", float("inf")):
                            print("Network won!")
                            should_exit = True
                if epoch_num >= self.max_epochs:
                    if self.writer:
                        self.save(os.path.join(self.nn_dir, "last_" + self.config["name"] + "ep" + str(epoch_num)
                                               + "rew" + str(self.mean_rewards)))
                    print("MAX EPOCHS NUM!")
                    should_exit = True
                if self.writer:
                    self.writer.flush()
            if self.multi_gpu:
                flag = torch.tensor(float(should_exit), device=self.device)
                dist.broadcast(flag, 0)
                should_exit = bool(flag.item())
            if should_exit:
                return self.last_mean_rewards, epoch_num

    def write_stats(self, total_time, epoch_num, play_time, update_time, stats, curr_frames):
        if not self.writer:
            return
        frame = self.frame
        w = self.writer
        w.add_scalar("performance/step_inference_rl_update_fps", curr_frames / (play_time + update_time), frame)
        w.add_scalar("performance/step_inference_fps", curr_frames / play_time, frame)
        w.add_scalar("performance/rl_update_time", update_time, frame)
        w.add_scalar("performance/step_inference_time", play_time, frame)
        w.add_scalar("losses/a_loss", stats["a_loss"], frame)
        w.add_scalar("losses/c_loss", stats["c_loss"], frame)
        w.add_scalar("losses/entropy", stats["entropy"], frame)
        w.add_scalar("losses/bounds_loss", stats["b_loss"], frame)
        w.add_scalar("info/last_lr", self.last_lr, frame)
        if getattr(self.optimizer, "amp_state", None) is not None:      # device-side GradScaler (fp16 update)
            w.add_scalar("info/loss_scale", self.optimizer.loss_scale, frame)
        w.add_scalar("info/e_clip", self.e_clip, frame)
        w.add_scalar("info/kl", stats["kl"], frame)
        w.add_scalar("info/epochs", epoch_num, frame)


class Runner:
    """The slice of ``rl_games.torch_runner.Runner`` train.py uses (train.py:139-171): load / reset / run."""

    def __init__(self, algo_observer=None):
        self.algo_observer = algo_observer

    def load(self, yaml_conf):
        self.default_config = yaml_conf["params"]
        self.params = copy.deepcopy(self.default_config)
        self.seed = self.params.get("seed", None)
        self.algo_name = self.params["algo"]["name"]
        if self.algo_name != "a2c_continuous":
            raise NotImplementedError("algo %r: only a2c_continuous is built" % self.algo_name)
        self.exp_config = None

    def reset(self):
        pass

    def run(self, args):
        if args.get("train", True):
            agent = A2CAgent("run", copy.deepcopy(self.params), vec_env=args.get("vec_env"),
                             algo_observer=self.algo_observer)
            ckpt = args.get("checkpoint")
            if ckpt:
                agent.restore(ckpt)
            return agent.train()
        from .player import PpoPlayerContinuous
        player = PpoPlayerContinuous(copy.deepcopy(self.params), vec_env=args.get("vec_env"))
        if args.get("checkpoint"):
            player.restore(args["checkpoint"])
        return player.run()
