"""CPU-side checks of the drop-in boundary: the product library loads, exports every symbol that
include/vine.h declares, agrees with the oracle on defaults, and refuses to run without a GPU."""
import ctypes as C
import os
import re

import pytest

from oracle import vine_oracle as vo
from vine_robot_isaacgymenvs_amd import abi, native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(header="vine.h"):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vine_[a-z_0-9]+)\s*\(", text)))


def test_header_and_ctypes_mirror_agree():
    assert header_functions() == sorted(abi.PROTOTYPES)


@pytest.fixture(scope="module")
def hip_lib():
    native.build()
    return native.load()


def test_product_library_exports_every_symbol(hip_lib):
    for name in header_functions():
        assert hasattr(hip_lib, name), name
    assert hip_lib.vine_backend_name() == b"hip-gfx950"


def test_ppo_header_symbols_exported_and_mirrored(hip_lib):
    """include/vine_ppo.h (fused PPO-update ops): every declared function is exported and has a ctypes prototype."""
    names = header_functions("vine_ppo.h")
    assert names == sorted(abi.PPO_PROTOTYPES)
    for name in names:
        assert hasattr(hip_lib, name), name
    text = open(os.path.join(REPO, "include", "vine_ppo.h")).read()
    assert int(re.search(r"#define VINE_PPO_PARTIAL_BLOCKS (\d+)", text).group(1)) == abi.PPO_PARTIAL_BLOCKS
    assert hip_lib.vine_trunk_args_size() == C.sizeof(abi.TrunkArgs)      # VineTrunkArgs <-> its ctypes mirror
    assert hip_lib.vine_step_rollout_args_size() == C.sizeof(abi.RolloutArgs)      # VineRolloutArgs likewise
    m = re.search(r"#define VINE_ROLLOUT_POST_SCRATCH_FLOATS \((\d+) \* (\d+)\)", text)
    assert int(m.group(1)) * int(m.group(2)) == abi.ROLLOUT_POST_SCRATCH_FLOATS


def test_oracle_exports_every_symbol():
    for prec in ("f32", "f64"):
        lib = vo.load(prec)
        for name in header_functions():
            assert hasattr(lib, name), name


def test_struct_layout_and_defaults_match_oracle(hip_lib):
    """Same VineConfig bytes from both implementations of vine_config_default / set_obs_type."""
    a, b = abi.VineConfig(), abi.VineConfig()
    assert hip_lib.vine_config_default(C.byref(a)) == 0
    assert vo.load("f64").vine_config_default(C.byref(b)) == 0
    assert bytes(a) == bytes(b)
    for obs_type in (0, 1):
        for scale in (0, 1):
            assert hip_lib.vine_config_set_obs_type(C.byref(a), obs_type, scale) == 0
            assert vo.load("f64").vine_config_set_obs_type(C.byref(b), obs_type, scale) == 0
            assert bytes(a) == bytes(b)
            assert hip_lib.vine_num_obs(C.byref(a)) == (28 if obs_type == 0 else 18)
    assert hip_lib.vine_config_set_obs_type(C.byref(a), 7, 1) == abi.ERR_INVALID_ARG


def test_field_enum_matches_header():
    text = open(os.path.join(REPO, "include", "vine.h")).read()
    for name, val in re.findall(r"\b(VF_[A-Z_0-9]+)\s*=\s*(\d+)\s*[,/]", text):
        assert getattr(abi, name) == int(val), name
    assert abi.VF_PIPE_Y == 42 + 2 * abi.MAX_DELAY and abi.VF_COUNT == 44 + 2 * abi.MAX_DELAY
    for name, shift in re.findall(r"\bVINE_(FLAG_[A-Z_]+)\s*=\s*1u << (\d+)", text):
        assert getattr(abi, name) == 1 << int(shift), name


def test_invalid_configs_are_rejected(hip_lib):
    c = abi.VineConfig()
    hip_lib.vine_config_default(C.byref(c))
    h = C.c_void_p()
    c.num_envs = 0
    assert hip_lib.vine_create(C.byref(c), 0, None, C.byref(h)) == abi.ERR_INVALID_ARG
    c.num_envs = 8
    c.action_delay = abi.MAX_DELAY + 1
    assert hip_lib.vine_create(C.byref(c), 0, None, C.byref(h)) == abi.ERR_INVALID_ARG
    c.action_delay = 1
    c.abi_version = 99
    assert hip_lib.vine_create(C.byref(c), 0, None, C.byref(h)) == abi.ERR_INVALID_ARG
    assert b"abi_version" in hip_lib.vine_last_error()


def test_observation_types_and_scaling_rules(hip_lib):
    """All six ObservationType members (V5:67-73) with their column counts (V5:152-170); scaling exists only for
    the two *_OBJ_INFO layouts (V5:267-268) -- product library, oracle and the Python front end agree."""
    from oracle import vine_oracle as vo
    from vine_robot_isaacgymenvs_amd.tasks.vine5link_moving_base import (ObservationType, num_observations,
                                                                         vine_config_from_cfg)
    from vine_robot_isaacgymenvs_amd.utils.config import load_task_config
    for lib in (hip_lib, vo.load("f64")):
        for t in ObservationType:
            code = abi.OBS_TYPE_BY_NAME[t.value]
            c = abi.VineConfig()
            lib.vine_config_default(C.byref(c))
            assert lib.vine_config_set_obs_type(C.byref(c), code, 0) == 0
            assert lib.vine_num_obs(C.byref(c)) == num_observations(t)
            assert all(c.obs_scaling[i] == 1.0 for i in range(abi.MAX_OBS))
            rc = lib.vine_config_set_obs_type(C.byref(c), code, 1)
            assert rc == (0 if code in abi.SCALABLE_OBS_TYPES else abi.ERR_UNSUPPORTED)
        assert lib.vine_config_set_obs_type(C.byref(c), 6, 0) == abi.ERR_INVALID_ARG
    cfg = load_task_config(overrides=["OBSERVATION_TYPE=POS_AND_VEL"])
    with pytest.raises(NotImplementedError, match="Observation scaling not implemented"):
        vine_config_from_cfg(cfg, hip_lib)
    cfg["env"]["SCALE_OBSERVATIONS"] = False
    c = vine_config_from_cfg(cfg, hip_lib)
    assert c.obs_type == abi.OBS_POS_AND_VEL and hip_lib.vine_num_obs(C.byref(c)) == 26


def test_product_fails_loudly_without_gpu(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    c = abi.VineConfig()
    hip_lib.vine_config_default(C.byref(c))
    c.num_envs = 8
    h = C.c_void_p()
    rc = hip_lib.vine_create(C.byref(c), 0, None, C.byref(h))
    assert rc == abi.ERR_NO_DEVICE
    assert b"no CPU path" in hip_lib.vine_last_error()
    with pytest.raises(RuntimeError):
        native.check(rc, hip_lib)


def test_task_class_refuses_cpu_device():
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    from vine_robot_isaacgymenvs_amd.utils.config import load_task_config
    cfg = load_task_config("Vine5LinkMovingBase", overrides=["num_envs=8"])
    with pytest.raises(RuntimeError, match="MI355X only"):
        isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg, rl_device="cpu", sim_device="cpu", graphics_device_id=-1,
                                                headless=True, virtual_screen_capture=False, force_render=False)
