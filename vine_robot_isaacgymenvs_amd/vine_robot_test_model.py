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
