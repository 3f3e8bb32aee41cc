"""Adapter layer between the RL runner and the task (isaacgymenvs/utils/rlgames_utils.py:41-180).

rl_games is not a dependency: ``RLGPUEnv`` keeps the ``IVecEnv`` method set rl_games calls
(step/reset/reset_done/get_number_of_agents/get_env_info) so that either the built-in PPO
(``learning/``) or an installed rl_games can drive it.
"""
import os
from typing import Callable

from ..tasks import isaacgym_task_map

env_configurations = {}   # name -> {'env_creator': thunk, 'vecenv_type': str}  (rl_games.common.env_configurations)


def register_env(name, config):
    env_configurations[name] = config


def get_rlgames_env_creator(seed: int, task_config: dict, task_name: str, sim_device: str, rl_device: str,
                            graphics_device_id: int, headless: bool, multi_gpu: bool = False,
                            post_create_hook: Callable = None, virtual_screen_capture: bool = False,
                            force_render: bool = False):
    """rlgames_utils.py:41-92: returns a thunk that builds the task from ``isaacgym_task_map``."""

    def create_rlgpu_env():
        task_config.setdefault("seed", seed)
        env = isaacgym_task_map[task_name](
            cfg=task_config, rl_device=rl_device, sim_device=sim_device, graphics_device_id=graphics_device_id,
            headless=headless, virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        if post_create_hook is not None:
            post_create_hook()
        return env

    return create_rlgpu_env


class RLGPUAlgoObserver:
    """rlgames_utils.py:95-148: lets the env log scalars next to the algorithm's statistics.  Scalars found in the
    ``infos`` dict returned by ``env.step`` are written as ``<key>/frame|iter|time`` to the agent's writer."""

    def __init__(self, env_stats_every=10):
        self.algo = None
        self.direct_info = {}
        self.env_stats_every = env_stats_every      # epochs between dumps of the task's wandb_dict scalars
        self.histogram_rows = []

    def after_init(self, algo):
        self.algo = algo
        self.writer = getattr(algo, "writer", None)
        self.env_stats_every = int(getattr(algo, "config", {}).get("env_stats_every", self.env_stats_every))
        env = self._task()
        if self.writer is not None and self.env_stats_every > 0 and hasattr(env, "bind_reward_matrix"):
            env.bind_reward_matrix()             # per-term reward entries of the dashboard (V5:1272-1284)

    def _task(self):
        vec_env = getattr(self.algo, "vec_env", None)
        return getattr(vec_env, "env", vec_env)

    def process_infos(self, infos, done_indices=None):
        assert isinstance(infos, dict), "RLGPUAlgoObserver expects dict info"
        self.direct_info = {}
        for k, v in infos.items():
            if isinstance(v, (float, int)) or (hasattr(v, "shape") and len(v.shape) == 0):
                self.direct_info[k] = v

    def after_print_stats(self, frame, epoch_num, total_time):
        if self.writer is None:
            return
        for k, v in self.direct_info.items():
            self.writer.add_scalar(f"{k}/frame", float(v), frame)
            self.writer.add_scalar(f"{k}/iter", float(v), epoch_num)
            self.writer.add_scalar(f"{k}/time", float(v), total_time)
        env = self._task()
        if env is None:
            return
        # the task's dashboard scalars (the reference wandb.log()s them from the step; here one device pass + one
        # device->host copy every `env_stats_every` epochs), same key names under "env/"
        if self.env_stats_every > 0 and epoch_num % self.env_stats_every == 0 and hasattr(env, "collect_stats"):
            for k, v in env.collect_stats().items():
                self.writer.add_scalar("env/" + k, v, frame)
        # CREATE_HISTOGRAMS_PERIODICALLY (V5:1392-1452): observations of env `index_to_view`, 100 consecutive control
        # steps per histogram set; taken from the rollout buffer so the step path stays free of host work
        cfg_env = getattr(env, "cfg", {}).get("env", {}) if hasattr(env, "cfg") else {}
        buf = getattr(self.algo, "buf", None)
        if cfg_env.get("CREATE_HISTOGRAMS_PERIODICALLY", False) and buf is not None and hasattr(env, "write_histograms"):
            self.histogram_rows += buf["obses"][:, env.index_to_view].detach().cpu().tolist()
            if len(self.histogram_rows) >= 100:
                env.write_histograms(self.histogram_rows[:100], os.path.join(os.path.dirname(self.writer.path), "histograms"))
                self.histogram_rows = []


class RLGPUEnv:
    """rlgames_utils.py:151-180."""

    def __init__(self, config_name, num_actors, **kwargs):
        self.env = env_configurations[config_name]["env_creator"](**kwargs)

    @classmethod
    def wrap(cls, env):
        """Adapter around an already built task object."""
        self = cls.__new__(cls)
        self.env = env
        return self

    def step(self, actions):
        return self.env.step(actions)

    def reset(self):
        return self.env.reset()

    def reset_done(self):
        return self.env.reset_done()

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()

    def get_env_info(self):
        info = {"action_space": self.env.action_space, "observation_space": self.env.observation_space}
        if self.env.num_states > 0:
            info["state_space"] = self.env.state_space
            print(info["action_space"], info["observation_space"], info["state_space"])
        else:
            print(info["action_space"], info["observation_space"])
        return info
