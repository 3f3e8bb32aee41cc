"""Config surface: the OmegaConf/Hydra subset used by the reference's YAMLs."""
import os

import pytest

from vine_robot_isaacgymenvs_amd.utils import config as cf

REF_CFG = "/root/reference/isaacgymenvs/cfg"


def test_defaults_compose_and_resolve():
    c = cf.load_config()
    env, sim = c["task"]["env"], c["task"]["sim"]
    assert c["task_name"] == "Vine5LinkMovingBase" and c["task"]["physics_engine"] == "physx"
    assert env["numEnvs"] == 4096 and env["controlFrequencyInv"] == 4
    assert isinstance(env["DAMPING"], float) and env["DAMPING"] == 0.02          # '2e-2' is a float, not a str
    assert env["RANDOM_INIT_CART_MIN_Y"] == pytest.approx(-0.03) and env["RANDOM_INIT_CART_MAX_Y"] == 0.3
    assert env["OBSERVATION_TYPE"] == "POS_AND_FD_VEL_AND_OBJ_INFO"
    assert sim["use_gpu_pipeline"] is True and sim["physx"]["use_gpu"] is True and sim["physx"]["num_threads"] == 4
    assert c["task"]["task"]["vine_randomize"] is True
    p = c["train"]["params"]
    assert p["seed"] == 42 and p["load_checkpoint"] is False and p["load_path"] == ""
    k = p["config"]
    assert k["name"] == "Vine5LinkMovingBase" and k["full_experiment_name"] == "Vine5LinkMovingBase"
    assert k["learning_rate"] == 3e-4 and isinstance(k["learning_rate"], float)
    assert k["num_actors"] == 4096 and k["horizon_length"] == 16 and k["minibatch_size"] == 32768
    assert k["max_epochs"] == 500 and k["score_to_win"] == 20_000_000_000 and k["device"] == "cuda:0"
    assert p["network"]["mlp"]["units"] == [256, 128, 64] and p["network"]["rnn"]["units"] == 256
    assert c["wandb_name"] == "Vine5LinkMovingBase"


def test_cli_overrides_like_the_readme():
    """README.md:63 style command line."""
    ov = ["task=Vine5LinkMovingBase", "num_envs=512", "max_iterations=600", "vine_randomize=False",
          "task.env.CREATE_SHELF=False", "task.env.OBSERVATION_TYPE=TIP_AND_CART_AND_OBJ_INFO",
          "task.env.maxEpisodeLength=100", "task.env.SUCCESS_DIST=0.04", "task.env.MIN_TARGET_Y=-0.4",
          "RAIL_SOFT_LIMIT=0.25", "RAIL_P_GAIN=30", "RAIL_ACCELERATION=6", "experiment=run7", "checkpoint=runs/a.pth",
          "sim_device=cuda:3", "pipeline=cpu", "train.params.config.gamma=0.95", "horizon_length=32"]
    c = cf.load_config(overrides=ov)
    env = c["task"]["env"]
    assert env["numEnvs"] == 512 and c["train"]["params"]["config"]["num_actors"] == 512
    assert env["OBSERVATION_TYPE"] == "TIP_AND_CART_AND_OBJ_INFO" and env["maxEpisodeLength"] == 100
    assert env["RAIL_SOFT_LIMIT"] == 0.25 and env["RANDOM_INIT_CART_MAX_Y"] == 0.25
    assert env["RANDOM_INIT_CART_MIN_Y"] == pytest.approx(-0.025)
    assert env["RAIL_P_GAIN"] == 30 and env["RAIL_ACCELERATION"] == 6
    assert c["task"]["task"]["vine_randomize"] is False
    assert c["task"]["sim"]["use_gpu_pipeline"] is False
    k = c["train"]["params"]["config"]
    assert k["max_epochs"] == 600 and k["name"] == "run7" and k["gamma"] == 0.95 and k["horizon_length"] == 32
    assert c["train"]["params"]["load_checkpoint"] is True and c["train"]["params"]["load_path"] == "runs/a.pth"


def test_unknown_keys_are_rejected_with_a_clear_message():
    """README.md:63 passes ACCEL_TARGET_SCALING_MIN, a key the task YAML does not have: Hydra refuses it too."""
    with pytest.raises(cf.ConfigError, match=r"\+task.env.ACCEL_TARGET_SCALING_MIN"):
        cf.load_config(overrides=["task.env.ACCEL_TARGET_SCALING_MIN=0.5"])
    c = cf.load_config(overrides=["+task.env.ACCEL_TARGET_SCALING_MIN=0.5"])
    assert c["task"]["env"]["ACCEL_TARGET_SCALING_MIN"] == 0.5
    with pytest.raises(cf.ConfigError):
        cf.load_config(overrides=["task=NoSuchTask"])
    with pytest.raises(cf.ConfigError):
        cf.load_config(overrides=["novalue"])


def test_resolver_grammar():
    cfg = {"a": {"b": 3, "c": "${.b}", "d": "${..top}", "e": "${eval:'2 * ${.b} + ${..n.m}'}", "s": "x_${.b}_${top}"},
           "top": "T", "n": {"m": 0.5}, "q": '${eq:${top},"t"}', "r": '${contains:"cu",${dev}}', "dev": "CUDA:1",
           "i": "${if:${flag},yes,no}", "flag": "", "j": "${resolve_default:7,${num}}", "num": "",
           "k": "${resolve_default:7,${num2}}", "num2": 9, "l": ["${top}", {"z": "${a.b}"}]}
    r = cf.resolve(cfg)
    assert r["a"] == {"b": 3, "c": 3, "d": "T", "e": 6.5, "s": "x_3_T"}
    assert r["q"] is True and r["r"] is True and r["i"] == "no" and r["j"] == 7 and r["k"] == 9
    assert r["l"] == ["T", {"z": 3}]
    with pytest.raises(cf.ConfigError):
        cf.resolve({"a": "${b}", "b": "${a}"})
    with pytest.raises(cf.ConfigError):
        cf.resolve({"a": "${nope:1}"})
    with pytest.raises(cf.ConfigError):
        cf.resolve({"a": "${missing.key}"})


def test_scalar_typing():
    assert cf.parse_scalar("2e-2") == 0.02 and cf.parse_scalar("3") == 3 and cf.parse_scalar("True") is True
    assert cf.parse_scalar("abc") == "abc" and cf.parse_scalar("'1'") == "1" and cf.parse_scalar("") == ""
    assert cf.parse_scalar("[1, 2]") == [1, 2] and cf.parse_scalar("null") is None


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference checkout not present (GPU box)")
def test_loader_reads_the_reference_yaml_files_unchanged():
    """The same loader composes the reference's own cfg directory; every reference key exists in ours with the
    same resolved value, except the documented deviations."""
    ref = cf.load_config(overrides=["task=Vine5LinkMovingBase"], config_dir=REF_CFG)
    ours = cf.load_config()
    deviations = {("task", "env", "CAPTURE_VIDEO")}

    def walk(a, b, path):
        for k, v in a.items():
            assert k in b, "missing key %s" % ".".join(path + (k,))
            if isinstance(v, dict):
                walk(v, b[k], path + (k,))
            elif path + (k,) not in deviations:
                assert b[k] == v, ".".join(path + (k,))

    walk(ref["task"], ours["task"], ("task",))
    walk(ref["train"], ours["train"], ("train",))
    for k, v in ref.items():
        if k not in ("task", "train"):
            assert k in ours and ours[k] == v, k


def test_dump_yaml_roundtrip(tmp_path):
    """The packaged defaults written out as a Hydra-style directory compose to the same config."""
    d = cf.dump_default_yaml(str(tmp_path / "cfg"))
    a = cf.load_config(overrides=["num_envs=77"])
    b = cf.load_config(overrides=["num_envs=77"], config_dir=d)
    assert a == b
