"""CPU stand-in for the task class, backed by the ORACLE (test infrastructure only).

Gives the host-side code paths that do not need a GPU (PPO agent, adapter, multi-process gloo tests) an
environment with the VecTask interface.  Never imported by the product package.
"""
import ctypes as C

import numpy as np
import torch

from oracle import vine_oracle as vo
from vine_robot_isaacgymenvs_amd import native  # noqa: F401  (kept: shares abi constants; never loaded here)
from vine_robot_isaacgymenvs_amd.tasks.base import spaces
from vine_robot_isaacgymenvs_amd.tasks.vine5link_moving_base import num_observations, ObservationType, vine_config_from_cfg


class OracleVecTask:
    def __init__(self, cfg, precision="f32", seed=42, omp=False):
        lib = vo.load(precision, omp)
        self.cfg = cfg
        self.vcfg = vine_config_from_cfg(cfg, lib, seed=seed)
        self.env = vo.OracleEnv(self.vcfg, precision, omp)
        self.num_envs = self.vcfg.num_envs
        self.num_obs = num_observations(ObservationType[cfg["env"]["OBSERVATION_TYPE"]])
        self.num_acts, self.num_states, self.num_agents = 2, 0, 1
        self.device = self.rl_device = "cpu"
        self.observation_space = spaces.Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self.action_space = spaces.Box(np.ones(2) * -1.0, np.ones(2) * 1.0)
        self.max_episode_length = cfg["env"]["maxEpisodeLength"]
        self.obs_dict = {}
        self.extras = {}

    @property
    def reset_buf(self):
        return torch.from_numpy(self.env.reset_buf)

    @property
    def progress_buf(self):
        return torch.from_numpy(self.env.progress)

    def reset(self):
        self.obs_dict["obs"] = torch.from_numpy(self.env.obs.copy())
        return self.obs_dict

    def step(self, actions):
        obs, rew, rst, to = self.env.step(actions.detach().cpu().numpy())
        self.obs_dict["obs"] = torch.from_numpy(obs.copy())
        self.extras["time_outs"] = torch.from_numpy(to.astype(bool))
        return self.obs_dict, torch.from_numpy(rew.copy()), torch.from_numpy(rst.copy()), self.extras

    def get_number_of_agents(self):
        return 1

    def get_env_info(self):
        return {"action_space": self.action_space, "observation_space": self.observation_space}
