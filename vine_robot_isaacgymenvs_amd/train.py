"""``python -m vine_robot_isaacgymenvs_amd.train task=Vine5LinkMovingBase key=value ...`` — the entry point
with the semantics of the reference's ``isaacgymenvs/train.py:35-183``: compose the config, rank -> device,
seed (+rank twice), register the env creator, dump ``runs/<name>/config.yaml`` and the config pickle, run."""
import datetime
import os
import pickle
import sys


def launch(cfg):
    from . import make
    from .learning.a2c_continuous import Runner
    from .utils.config import to_yaml
    from .utils.rlgames_utils import RLGPUAlgoObserver, RLGPUEnv, register_env  # noqa: F401
    from .utils.utils import set_np_formatting, set_seed

    time_str = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
    if cfg["checkpoint"]:
        cfg["checkpoint"] = os.path.abspath(cfg["checkpoint"])          # train.py:60-63
        cfg["train"]["params"]["load_path"] = cfg["checkpoint"]
    set_np_formatting()

    rank = int(os.getenv("LOCAL_RANK", "0"))
    if cfg["multi_gpu"]:                                                 # train.py:71-75
        cfg["sim_device"] = f"cuda:{rank}"
        cfg["rl_device"] = f"cuda:{rank}"
        cfg["train"]["params"]["config"]["device"] = cfg["rl_device"]
    if str(cfg["rl_device"]).startswith("cuda"):
        import torch
        if torch.cuda.is_available():        # the rank's GPU is the current device of the process from here on
            torch.cuda.set_device(torch.device(cfg["rl_device"]))
    cfg["seed"] += rank                                                  # train.py:78 (and set_seed adds it again)
    cfg["seed"] = set_seed(cfg["seed"], torch_deterministic=cfg["torch_deterministic"], rank=rank)
    cfg["train"]["params"]["seed"] = cfg["seed"]

    def create_env_thunk(**kwargs):
        return make(cfg["seed"], cfg["task_name"], cfg["task"]["env"]["numEnvs"], cfg["sim_device"], cfg["rl_device"],
                    cfg["graphics_device_id"], cfg["headless"], cfg["multi_gpu"], cfg["capture_video"],
                    cfg["force_render"], cfg, **kwargs)

    register_env("rlgpu", {"vecenv_type": "RLGPU", "env_creator": create_env_thunk})   # train.py:122-127

    rlg_config_dict = cfg["train"]
    runner = Runner(RLGPUAlgoObserver())                               # train.py:139-141
    runner.load(rlg_config_dict)
    runner.reset()

    if rank == 0:                                                        # train.py:147-163
        experiment_dir = os.path.join("runs", cfg["train"]["params"]["config"]["name"])
        os.makedirs(experiment_dir, exist_ok=True)
        with open(os.path.join(experiment_dir, "config.yaml"), "w") as f:
            f.write(to_yaml(cfg))
        with open(os.path.join(experiment_dir, f"{time_str}_rlg_config_dict.pkl"), "wb") as f:
            pickle.dump(rlg_config_dict, f)

    return runner.run({"train": not cfg["test"], "play": cfg["test"], "checkpoint": cfg["checkpoint"], "sigma": None})


def main(argv=None):
    from .utils.config import load_config
    argv = sys.argv[1:] if argv is None else argv
    cfg = load_config("config", overrides=argv)
    return launch(cfg)


if __name__ == "__main__":
    main()
