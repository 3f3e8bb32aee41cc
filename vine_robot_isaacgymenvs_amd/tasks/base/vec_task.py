"""``Env`` / ``VecTask``: the environment-side contract of the reference, kept signature-for-signature
(isaacgymenvs/tasks/base/vec_task.py:60-162, 165-223, 260-283, 319-427) with the Isaac Gym layer
replaced by one fused HIP launch per ``step`` behind the C ABI of ``include/vine.h``.

What is kept: constructor signature, ``step/reset/reset_done/reset_idx/zero_actions/get_state``,
buffer names and dtypes, ``observation_space/action_space/num_envs/num_acts/num_obs``.
What is dropped (out of scope): viewer, rendering, generic domain randomisation.
"""
import abc
from typing import Any, Dict, Tuple

import numpy as np
import torch

from . import spaces


class Env(abc.ABC):
    def __init__(self, config: Dict[str, Any], rl_device: str, sim_device: str, graphics_device_id: int,
                 headless: bool):
        """Mirrors vec_task.py:61-108 (device parsing, sizes, spaces, clip values)."""
        split_device = sim_device.split(":")
        self.device_type = split_device[0]
        self.device_id = int(split_device[1]) if len(split_device) > 1 else 0

        if self.device_type.lower() not in ("cuda", "gpu"):
            # The reference falls back to a CPU pipeline here (vec_task.py:75-81).  This build is the
            # MI355X path only; a silent CPU route would void every parity/perf claim.
            raise RuntimeError(
                "sim_device=%r: vine_robot_isaacgymenvs_amd runs on MI355X only (HIP kernels, no CPU pipeline)"
                % sim_device)
        self.device = "cuda:" + str(self.device_id)
        config["sim"]["use_gpu_pipeline"] = True
        self.rl_device = rl_device

        self.headless = headless
        self.graphics_device_id = -1  # no renderer on the target

        self.num_environments = config["env"]["numEnvs"]
        self.num_agents = config["env"].get("numAgents", 1)
        self.num_observations = config["env"]["numObservations"]
        self.num_states = config["env"].get("numStates", 0)
        self.num_actions = config["env"]["numActions"]

        self.control_freq_inv = config["env"].get("controlFrequencyInv", 1)

        self.obs_space = spaces.Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self.state_space = spaces.Box(np.ones(self.num_states) * -np.inf, np.ones(self.num_states) * np.inf)
        self.act_space = spaces.Box(np.ones(self.num_actions) * -1., np.ones(self.num_actions) * 1.)

        self.clip_obs = config["env"].get("clipObservations", np.inf)
        self.clip_actions = config["env"].get("clipActions", np.inf)

    @abc.abstractmethod
    def allocate_buffers(self):
        """Create torch buffers for observations, rewards, actions dones and any additional data."""

    @abc.abstractmethod
    def step(self, actions: torch.Tensor) -> Tuple[Dict[str, torch.Tensor], torch.Tensor, torch.Tensor, Dict[str, Any]]:
        """Step the physics of the environment."""

    @abc.abstractmethod
    def reset(self) -> Dict[str, torch.Tensor]:
        """Return the current observation dictionary."""

    @abc.abstractmethod
    def reset_idx(self, env_ids: torch.Tensor):
        """Reset environments having the provided indices."""

    @property
    def observation_space(self):
        return self.obs_space

    @property
    def action_space(self):
        return self.act_space

    @property
    def num_envs(self) -> int:
        return self.num_environments

    @property
    def num_acts(self) -> int:
        return self.num_actions

    @property
    def num_obs(self) -> int:
        return self.num_observations


class VecTask(Env):
    metadata = {"render.modes": [], "video.frames_per_second": 24}

    def __init__(self, config, rl_device, sim_device, graphics_device_id, headless,
                 virtual_screen_capture: bool = False, force_render: bool = False):
        super().__init__(config, rl_device, sim_device, graphics_device_id, headless)
        self.virtual_screen_capture = False   # accepted and ignored: no display on the target
        self.force_render = False
        if self.cfg["physics_engine"] != "physx":
            # same error as vec_task.py:194-196 for an unknown backend; "flex" has no analogue here
            raise ValueError(f"Invalid physics engine backend: {self.cfg['physics_engine']}")
        self.viewer = None
        self.sim_initialized = False
        self.create_sim()
        self.sim_initialized = True
        self.allocate_buffers()
        self.obs_dict = {}

    def allocate_buffers(self):
        """Same names, shapes and dtypes as vec_task.py:260-283 (reset_buf starts at ones)."""
        n, dev = self.num_envs, self.device
        # two observation buffers, used alternately, so the dict returned by step k stays valid during step k+1
        self._obs_ring = [torch.zeros((n, self.num_obs), device=dev, dtype=torch.float) for _ in range(2)]
        self._obs_slot = 0
        self.obs_buf = self._obs_ring[0]
        self.states_buf = torch.zeros((n, self.num_states), device=dev, dtype=torch.float)
        self.rew_buf = torch.zeros(n, device=dev, dtype=torch.float)
        self.reset_buf = torch.ones(n, device=dev, dtype=torch.long)
        self.timeout_buf = torch.zeros(n, device=dev, dtype=torch.bool)
        self.progress_buf = torch.zeros(n, device=dev, dtype=torch.long)
        self.randomize_buf = torch.zeros(n, device=dev, dtype=torch.long)
        self.extras = {}

    @abc.abstractmethod
    def create_sim(self):
        """Create the native environment handle."""

    @abc.abstractmethod
    def _native_step(self, actions: torch.Tensor, obs_out: torch.Tensor):
        """One fused launch: writes obs_out, rew_buf, reset_buf, progress_buf, timeout_buf."""

    def get_state(self):
        return torch.clamp(self.states_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    def step(self, actions: torch.Tensor):
        """vec_task.py:319-380.  Action clamp, pre/4x(actuation+simulate)/post, time-outs and the
        observation clamp all happen inside the kernel; this method only marshals tensors."""
        a = actions
        if a.device != self.rew_buf.device or a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        self._obs_slot ^= 1
        self.obs_buf = self._obs_ring[self._obs_slot]
        self._native_step(a, self.obs_buf)
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        self.obs_dict["obs"] = self.obs_buf.to(self.rl_device)
        if self.num_states > 0:
            self.obs_dict["states"] = self.get_state()
        return self.obs_dict, self.rew_buf.to(self.rl_device), self.reset_buf.to(self.rl_device), self.extras

    def step_into(self, actions: torch.Tensor, obs_out: torch.Tensor):
        """``step`` with the observation written straight into the caller's buffer (e.g. the next slot of a rollout
        buffer: saves the copy a trainer would make) -- an extension of the reference API.  ``obs_buf`` (and
        ``obs_dict["obs"]``, ``extras["time_outs"]``) are re-bound to the caller's tensor as views, no copy, so that
        ``reset()`` / ``reset_done()`` / a player on the same env keep returning the CURRENT observation (the reference
        re-binds obs_buf every step too, V5:1385).  A trainer that replays captured steps from a hipGraph must make the
        LAST step of the captured sequence write to a tensor it keeps (the agent's ``_obs_last``): the binding made at
        capture time then stays the current observation after every replay.
        obs_out: [num_envs, num_obs] float32, contiguous, on the sim device.  Returns obs_out."""
        if (obs_out.shape != self.obs_buf.shape or obs_out.dtype != torch.float32 or not obs_out.is_contiguous()
                or obs_out.device != self.obs_buf.device):
            raise ValueError("step_into: obs_out must be a contiguous float32 [num_envs, num_obs] tensor on the sim device")
        a = actions
        if a.device != self.rew_buf.device or a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        self._native_step(a, obs_out)
        self.obs_buf = obs_out
        self.obs_dict["obs"] = obs_out.to(self.rl_device)          # same device: the tensor itself
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        return obs_out

    def zero_actions(self) -> torch.Tensor:
        return torch.zeros([self.num_envs, self.num_actions], dtype=torch.float32, device=self.rl_device)

    def reset_idx(self, env_idx):
        pass

    def reset(self):
        """Called once at start; returns the current (zero) buffer without simulating (vec_task.py:398-410)."""
        self.obs_dict["obs"] = torch.clamp(self.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)
        if self.num_states > 0:
            self.obs_dict["states"] = self.get_state()
        return self.obs_dict

    def reset_done(self):
        """vec_task.py:412-427."""
        done_env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        if len(done_env_ids) > 0:
            self.reset_idx(done_env_ids)
        self.obs_dict["obs"] = torch.clamp(self.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)
        if self.num_states > 0:
            self.obs_dict["states"] = self.get_state()
        return self.obs_dict, done_env_ids

    def get_number_of_agents(self):
        return self.num_agents
