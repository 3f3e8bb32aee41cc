"""Actor-critic network selected by cfg/train/Vine5LinkMovingBasePPO.yaml:10-40:
``separate: False``; obs -> MLP[256,128,64] (ELU) -> concat with obs (``concat_input``) -> LSTM(256)
(``before_mlp: False``) -> LayerNorm -> mu (Linear, no activation), value (Linear); log-sigma is a free
parameter initialised to 0 (``fixed_sigma``, ``const_initializer 0``).  State-dict key names follow
rl_games (``a2c_network.actor_mlp.0.weight`` ... ``a2c_network.rnn.rnn.weight_ih_l0``) so checkpoints
written by either side load in the other.

The LSTM is stepped explicitly (one fused gate GEMM + ``torch._VF.lstm_cell`` pointwise per time step): the
rollout needs single steps, the update needs length-``seq_len`` sequences with the hidden state zeroed
wherever the stored ``dones`` flag is set, which a per-step loop expresses exactly.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused
from .running_mean_std import RunningMeanStd

_ACTIVATIONS = {"elu": nn.ELU, "relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "selu": nn.SELU,
                "None": nn.Identity, None: nn.Identity}


class _LSTM(nn.Module):
    """Parameter container with ``nn.LSTM``'s names/shapes/initialisation (1 layer, batch-second)."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / math.sqrt(hidden_size)
        self.weight_ih_l0 = nn.Parameter(torch.empty(4 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh_l0 = nn.Parameter(torch.empty(4 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih_l0 = nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k))
        self.bias_hh_l0 = nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k))

    def cell(self, x, h, c):
        return torch._VF.lstm_cell(x, (h, c), self.weight_ih_l0, self.weight_hh_l0, self.bias_ih_l0, self.bias_hh_l0)


class _RnnWrap(nn.Module):
    """rl_games wraps the torch RNN in a module called ``rnn`` (keys ``rnn.rnn.weight_ih_l0``)."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.rnn = _LSTM(input_size, hidden_size)

    def forward(self, x, states, dones=None):
        """x [T, B, F]; states (h, c) each [1, B, H]; dones [T, B] or None: state is zeroed before step t
        where dones[t] != 0 (the env finished an episode at step t-1)."""
        h, c = states[0][0], states[1][0]
        outs = []
        for t in range(x.shape[0]):
            if dones is not None:
                keep = (1.0 - dones[t].to(h.dtype)).unsqueeze(-1)
                h, c = h * keep, c * keep
            h, c = self.rnn.cell(x[t], h, c)
            outs.append(h)
        return torch.stack(outs, 0), (h.unsqueeze(0), c.unsqueeze(0))

    def forward_flat(self, x, states, dones, seq_length):
        """Sequence-major rows (index = seq * seq_length + t): the layout of the rollout buffers and minibatches.
        On the GPU this is the fused path (one input GEMM for all steps + hand-written pointwise kernels)."""
        r = self.rnn
        out, h, c = fused.lstm_sequence(x, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0,
                                        states[0][0], states[1][0], dones, seq_length)
        return out, (h.unsqueeze(0), c.unsqueeze(0))


class A2CNetwork(nn.Module):
    def __init__(self, params, actions_num, input_shape):
        super().__init__()
        mlp, rnn = params["mlp"], params.get("rnn")
        self.units = list(mlp["units"])
        act = _ACTIVATIONS[mlp.get("activation", "elu")]
        if params.get("separate", False):
            raise NotImplementedError("separate actor/critic networks are not used by Vine5LinkMovingBasePPO")
        if rnn is None or rnn.get("name") != "lstm" or rnn.get("layers", 1) != 1 or rnn.get("before_mlp", False):
            raise NotImplementedError("only the configuration of Vine5LinkMovingBasePPO.yaml is built: "
                                      "1-layer LSTM after the MLP")
        num_inputs = int(input_shape[0])
        layers, in_size = [], num_inputs
        for u in self.units:
            layers += [fused.SplitKLinear(in_size, u), act()]
            in_size = u
        self.actor_mlp = nn.Sequential(*layers)
        self.activation_is_elu = act is nn.ELU
        # set by the agent for `mixed_precision: True`: parameter -> its bfloat16 shadow (learning/flat_adam.py)
        self.op_weight_lookup = None
        self.rnn_units = int(rnn["units"])
        self.rnn_concat_input = bool(rnn.get("concat_input", False))
        rnn_in = in_size + (num_inputs if self.rnn_concat_input else 0)
        self.rnn = _RnnWrap(rnn_in, self.rnn_units)
        self.rnn_ln = bool(rnn.get("layer_norm", False))
        if self.rnn_ln:
            self.layer_norm = nn.LayerNorm(self.rnn_units)
        self.value = fused.SplitKLinear(self.rnn_units, 1)
        self.mu = fused.SplitKLinear(self.rnn_units, actions_num)
        space = params["space"]["continuous"]
        if not space.get("fixed_sigma", True):
            raise NotImplementedError("fixed_sigma: False")
        self.sigma = nn.Parameter(torch.zeros(actions_num, dtype=torch.float32), requires_grad=True)
        # rl_games `default` initialiser = identity on weights, zeros on Linear biases
        for m in self.modules():
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)
        nn.init.constant_(self.sigma, float(space.get("sigma_init", {}).get("val", 0)))

    def get_default_rnn_state(self, batch, device=None):
        z = torch.zeros((1, batch, self.rnn_units), device=device)
        return (z, z.clone())

    def forward_heads(self, obs, states, seq_length, dones, head_bias_external=False, norm=None, loss_pack=None):
        """Training forward on the MI355X as one fused autograd node (learning/fused.py:_Trunk):
        -> (heads [n, A+1] = [mu | value], states).  Caller checks ``fused.trunk_supported`` first."""
        r = self.rnn.rnn
        mlp = [(m.weight, m.bias) for m in self.actor_mlp if isinstance(m, nn.Linear)]
        op_weights = None
        if self.op_weight_lookup is not None:      # mixed precision: bf16 shadows kept by the optimiser
            op_weights = [self.op_weight_lookup(w) for w, _ in mlp] + [self.op_weight_lookup(r.weight_ih_l0),
                                                                       self.op_weight_lookup(r.weight_hh_l0)]
        heads, h, c = fused.trunk(obs, states[0][0], states[1][0], dones, seq_length, self.rnn_concat_input, mlp,
                                  (r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0),
                                  (self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps),
                                  (self.mu.weight, self.mu.bias, self.value.weight, self.value.bias),
                                  op_weights=op_weights, head_bias_external=head_bias_external, norm=norm,
                                  loss_pack=loss_pack)
        return heads, (h.unsqueeze(0), c.unsqueeze(0))

    def trunk_supported(self, obs, seq_length):
        if torch.is_autocast_enabled():
            return False
        return fused.trunk_supported(obs, self.units, self.activation_is_elu, self.rnn_units, self.rnn_ln, seq_length)

    def forward(self, obs, states, seq_length=1, dones=None):
        """obs [B, num_obs] ordered sequence-major (index = seq * seq_length + t)."""
        out = self.actor_mlp(obs)
        if self.rnn_concat_input:
            out = torch.cat([out, obs], dim=1)
        out, states = self.rnn.forward_flat(out, states, dones, seq_length)
        if self.rnn_ln:
            out = self.layer_norm(out)
        value = self.value(out)
        mu = self.mu(out)
        return mu, mu * 0.0 + self.sigma, value, states


class ModelA2CContinuousLogStd(nn.Module):
    """rl_games ``continuous_a2c_logstd`` model: normalisers + network + Gaussian head."""

    def __init__(self, network_params, actions_num, obs_shape, normalize_value, normalize_input, value_size=1):
        super().__init__()
        self.a2c_network = A2CNetwork(network_params, actions_num, obs_shape)
        self.normalize_value, self.normalize_input = normalize_value, normalize_input
        if normalize_value:
            self.value_mean_std = RunningMeanStd((value_size,))
        if normalize_input:
            self.running_mean_std = RunningMeanStd(tuple(obs_shape))

    def is_rnn(self):
        return True

    def get_default_rnn_state(self, batch, device=None):
        return self.a2c_network.get_default_rnn_state(batch, device)

    def norm_obs(self, obs):
        return self.running_mean_std(obs) if self.normalize_input else obs

    def unnorm_value(self, value):
        return self.value_mean_std(value, unnorm=True) if self.normalize_value else value

    @staticmethod
    def neglogp(x, mean, std, logstd):
        return (0.5 * (((x - mean) / std) ** 2).sum(dim=-1) + 0.5 * np.log(2.0 * np.pi) * x.size()[-1]
                + logstd.sum(dim=-1))

    def forward_raw(self, input_dict):
        """Training forward for the fused loss: (mu [n,A], value [n,1], logstd parameter [A], rnn states, heads).
        ``heads`` is the [n, A+1] = [mu | value] block when the fused trunk ran (mu/value are views of it), else None.
        ``obs_is_normalized``: ``obs`` already went through ``norm_obs`` (the captured optimiser step keeps the
        running-statistics update outside the hipGraph); ``head_bias_external``: see ``fused.trunk``."""
        net = self.a2c_network
        T = input_dict.get("seq_length", 1)
        raw, norm = input_dict["obs"], None
        rms = self.running_mean_std if self.normalize_input else None
        if (rms is not None and not input_dict.get("obs_is_normalized", False) and net.op_weight_lookup is not None
                and rms._use_kernels(raw) and net.trunk_supported(raw, T)):
            # mixed-precision trunk: update the running statistics here (training mode), let the trunk normalise the
            # raw observations itself (one launch for normalisation + MLP)
            ahead = input_dict.get("obs_norm_stats")
            if ahead is not None:
                # (the mini-epoch graph ran the statistics updates of all its steps up front -- vine_rms_update_multi --
                # and hands each step the moments its own update would have left)
                obs, norm = raw, (ahead[0], ahead[1], rms.epsilon)
            else:
                rms.update_kernels(raw)
                obs, norm = raw, (rms.running_mean, rms.running_var, rms.epsilon)
        else:
            obs = raw if input_dict.get("obs_is_normalized", False) else self.norm_obs(raw)
        if net.trunk_supported(obs, T):
            heads, states = net.forward_heads(obs, input_dict["rnn_states"], T, input_dict.get("dones", None),
                                              input_dict.get("head_bias_external", False), norm=norm,
                                              loss_pack=input_dict.get("loss_pack"))
            A = net.mu.weight.shape[0]
            return heads[:, :A], heads[:, A:], net.sigma, states, heads
        mu, _logstd, value, states = net(obs, input_dict["rnn_states"], T, input_dict.get("dones", None))
        return mu, value, net.sigma, states, None

    def forward(self, input_dict):
        is_train = input_dict.get("is_train", True)
        prev_actions = input_dict.get("prev_actions", None)
        obs = self.norm_obs(input_dict["obs"])
        mu, logstd, value, states = self.a2c_network(obs, input_dict["rnn_states"], input_dict.get("seq_length", 1),
                                                    input_dict.get("dones", None))
        sigma = torch.exp(logstd)
        if is_train:
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd).sum(dim=-1)
            prev_neglogp = self.neglogp(prev_actions, mu, sigma, logstd)
            return {"prev_neglogp": prev_neglogp, "values": value, "entropy": entropy, "rnn_states": states,
                    "mus": mu, "sigmas": sigma}
        selected_action = mu + sigma * torch.randn_like(mu)
        neglogp = self.neglogp(selected_action, mu, sigma, logstd)
        return {"neglogpacs": neglogp, "values": self.unnorm_value(value), "actions": selected_action,
                "rnn_states": states, "mus": mu, "sigmas": sigma}
