"""vine_robot_isaacgymenvs_amd — MI355X-native drop-in for the Vine5LinkMovingBase hot path of
tylerlum/Vine_Robot_IsaacGymEnvs (env step + PPO rollout/update).

Public surface mirrors ``isaacgymenvs/__init__.py:15-56``: ``make(seed, task, num_envs, sim_device, rl_device, ...)``.
"""
from .utils.config import RESOLVERS, load_config, load_task_config  # noqa: F401  (resolvers: __init__.py:8-12)

__all__ = ["make", "load_config", "load_task_config"]


def make(seed: int, task: str, num_envs: int, sim_device: str, rl_device: str, graphics_device_id: int = -1,
         headless: bool = False, multi_gpu: bool = False, virtual_screen_capture: bool = False,
         force_render: bool = True, cfg=None):
    """Create the vectorised task (isaacgymenvs/__init__.py:15-56).  ``cfg`` is the composed root config dict
    (what train.py passes); when None the packaged YAMLs are composed for ``task`` and ``num_envs`` is applied."""
    from .utils.rlgames_utils import get_rlgames_env_creator
    if cfg is None:
        cfg_dict = load_task_config(task)
        cfg_dict["env"]["numEnvs"] = num_envs
    else:
        cfg_dict = cfg["task"]
    cfg_dict["seed"] = seed
    create_rlgpu_env = get_rlgames_env_creator(
        seed=seed, task_config=cfg_dict, task_name=cfg_dict["name"], sim_device=sim_device, rl_device=rl_device,
        graphics_device_id=graphics_device_id, headless=headless, multi_gpu=multi_gpu,
        virtual_screen_capture=virtual_screen_capture, force_render=force_render)
    return create_rlgpu_env()
