"""Deployment-side policy loader: what ``isaacgymenvs/vine_robot_test_model.py:143-177`` does with rl_games'
``PpoPlayerContinuous`` -- rebuild the policy from the run's config pickle (written by train.py next to
``config.yaml``) and a ``.pth`` checkpoint, then map observations to actions without any simulator.

    policy = VinePolicy.load("runs/Vine5LinkMovingBase/<time>_rlg_config_dict.pkl", "runs/.../nn/Vine5LinkMovingBase.pth")
    action = policy.get_action(obs)          # obs: [num_obs] or [B, num_obs], already scaled like the task's obs_buf

The observation layout is the task's (V5:1339-1390): e.g. for TIP_AND_CART_AND_OBJ_INFO
``cat[cart_y, fd_cart_v, tip_pos(3), fd_tip_vel(3), target_pos(3), target_vel(3), smoothed_u_fpam, prev_u_rail, obj_info(2)] / obs_scaling``.
"""
import pickle

import numpy as np
import torch

from .learning.network import ModelA2CContinuousLogStd


class VinePolicy:
    def __init__(self, params, num_obs, num_actions=2, device="cpu"):
        config = params["config"]
        self.device = torch.device(device)
        self.model = ModelA2CContinuousLogStd(params["network"], num_actions, (num_obs,),
                                              config.get("normalize_value", False), config["normalize_input"]).to(self.device)
        self.model.eval()
        self.states = None

    @classmethod
    def load(cls, config_pickle, checkpoint, device="cpu"):
        with open(config_pickle, "rb") as f:
            rlg = pickle.load(f)
        ckpt = torch.load(checkpoint, map_location=device, weights_only=False)
        num_obs = ckpt["model"]["running_mean_std.running_mean"].shape[0] if "running_mean_std.running_mean" in ckpt["model"] \
            else ckpt["model"]["a2c_network.actor_mlp.0.weight"].shape[1]
        num_actions = ckpt["model"]["a2c_network.mu.weight"].shape[0]
        self = cls(rlg["params"], num_obs, num_actions, device)
        self.model.load_state_dict(ckpt["model"])
        return self

    def reset(self):
        """Forget the LSTM state (call at the start of an episode)."""
        self.states = None

    @torch.no_grad()
    def get_action(self, obs, deterministic=True):
        x = torch.as_tensor(np.asarray(obs, dtype=np.float32) if not torch.is_tensor(obs) else obs, device=self.device)
        single = x.dim() == 1
        if single:
            x = x.unsqueeze(0)
        if self.states is None or self.states[0].shape[1] != x.shape[0]:
            self.states = [s.clone() for s in self.model.get_default_rnn_state(x.shape[0], self.device)]
        res = self.model({"is_train": False, "prev_actions": None, "obs": x, "rnn_states": self.states})
        self.states = list(res["rnn_states"])
        a = torch.clamp(res["mus"] if deterministic else res["actions"], -1.0, 1.0)
        return a[0] if single else a


class VineRobotControlModel(torch.nn.Module):
    """The reference's robot-side entry point (``isaacgymenvs/vine_robot_test_model.py:143-177``), same constructor and
    methods, without rl_games:

        model = VineRobotControlModel(config_path, checkpoint_path, x_range=(-10.0, 10.0), u_range=(-0.1, 3.0)).to(device)
        action = model.get_action(q, qd, tip_pos, tip_vel, target_pos)      # (rail command, FPAM pressure u)

    ``config_path``: the ``*_rlg_config_dict.pkl`` train.py writes into ``runs/<name>/`` (train.py:160-163);
    ``checkpoint_path``: a ``nn/*.pth`` of the same run.  ``get_action`` concatenates its arguments into ONE observation
    row in the order given (vine_robot_test_model.py:162; their total width must be the ``num_obs`` the checkpoint was
    trained with -- the robot-side caller assembles, and scales, exactly the columns its policy was trained on), runs the
    policy with its LSTM state carried from call to call, clamps the action to [-1, 1] and maps it to the physical
    ranges: element 0 to ``x_range`` (the rail command), element 1 -- or the only element of a 1-action policy -- to
    ``u_range`` (vine_robot_test_model.py:165-170).  As in the reference's player the action is SAMPLED
    (``is_determenistic=False`` there); ``deterministic=True`` returns the mean instead."""

    def __init__(self, config_path, checkpoint_path, x_range, u_range, deterministic=False):
        super().__init__()
        self.config_path = config_path
        self.checkpoint_path = checkpoint_path
        self.rail_force_min, self.rail_force_max = x_range
        self.u_min, self.u_max = u_range
        self.deterministic = bool(deterministic)
        with open(config_path, "rb") as f:
            self.cfg = pickle.load(f)
        self.policy = VinePolicy.load(config_path, checkpoint_path, device="cpu")
        self.model = self.policy.model                    # registered submodule: .to(device) moves the weights
        self.num_obs = int(self.model.a2c_network.actor_mlp[0].weight.shape[1]) if hasattr(
            self.model.a2c_network, "actor_mlp") else None

    def _apply(self, fn, *a, **k):                        # keep the policy's device / LSTM state in step with .to()
        out = super()._apply(fn, *a, **k)
        self.policy.device = next(self.model.parameters()).device
        self.policy.states = None
        return out

    def reset(self):
        self.policy.reset()

    def get_action(self, q, qd, tip_pos, tip_vel, target_pos):
        obs = torch.cat([q, qd, tip_pos, tip_vel, target_pos])[None, ...].to(q.device)      # (1, sum(xi))
        if self.num_obs is not None and obs.shape[1] != self.num_obs:
            raise ValueError("get_action: the concatenated observation has %d columns, the checkpoint was trained on %d"
                             % (obs.shape[1], self.num_obs))
        action = self.forward(obs)[0].clone()                                              # (action_dim,)
        if torch.numel(action) == 1:
            return self.rescale(action, self.u_min, self.u_max)
        elif torch.numel(action) == 2:
            action[0] = self.rescale(action[0], self.rail_force_min, self.rail_force_max)
            action[1] = self.rescale(action[1], self.u_min, self.u_max)
        return action

    def forward(self, obs):
        return self.policy.get_action(obs, deterministic=self.deterministic)

    def rescale(self, x, low, high):
        return (x + 1) * (high - low) / 2 + low
