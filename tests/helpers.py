"""Shared helpers for the parity tests: VineConfig equivalents of the fixture configurations."""
import ctypes as C

import numpy as np

from oracle import vine_oracle as vo
from vine_robot_isaacgymenvs_amd import abi


def base_cfg(num_envs, obs_type=abi.OBS_POS_AND_FD_VEL_AND_OBJ_INFO, randomize=False, **over):
    """TY defaults (reference task YAML) with vine_randomize off unless asked."""
    cfg = vo.default_config(num_envs=num_envs)
    # the four non-scalable observation types only exist unscaled (V5:267-268)
    rc = vo.load().vine_config_set_obs_type(C.byref(cfg), obs_type, int(obs_type in abi.SCALABLE_OBS_TYPES))
    assert rc == 0, rc
    cfg.set_flag(abi.FLAG_VINE_RANDOMIZE, randomize)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


F6_CASES = [("delay1", 1, 0, False), ("delay0_tipobs", 0, 1, False), ("delay2", 2, 0, False),
            ("held_damping008", 1, 0, "damping008"), ("held_efflim03", 1, 0, "efflim03"),
            ("shelf_delay1", 1, 0, "shelf"), ("shelf_contact_reset", 1, 0, "shelf_contact_reset"), ("pipe_delay1", 1, 0, "pipe")]
F6_OBSTACLE_CASES = [c for c in F6_CASES if c[0].startswith(("shelf", "pipe"))]


def f6_cfg(num_envs, action_delay, obs_type, held=False):
    """The configuration tests/golden/make_golden.py used for the F6 trajectories (env_overrides array inside the
    fixture): the task YAML's own DAMPING (0.02) in the product's default physics mode, or -- ``held`` -- the reference's
    literal actuation (efforts of V5:1062 incl. C_j*qd_j held over the sim step) at DAMPING 0.08 ("damping008") or at the
    YAML's DAMPING with a 0.3 N m joint effort clamp ("efflim03"); "shelf" / "shelf_contact_reset" / "pipe" = the default
    mode with that obstacle (BASELINE configs[4]'s step sequencing; the task YAML's default obstacle)."""
    cfg = base_cfg(num_envs, obs_type)
    variant = held or ""
    if variant in ("damping008", "efflim03"):
        cfg.set_flag(abi.FLAG_FPAM_DAMPING_HELD, True)
        if variant == "efflim03":
            cfg.effort_limit = 0.3            # at the YAML's own DAMPING (0.02)
        else:
            cfg.damping = 0.08
    cfg.max_episode_length = 20
    cfg.success_dist = 0.12
    cfg.rail_soft_limit = 0.2
    cfg.min_target_y, cfg.max_target_y = -0.3, -0.1
    cfg.min_target_z, cfg.max_target_z = 0.53, 0.6
    cfg.random_init_cart_min_y, cfg.random_init_cart_max_y = -0.02, 0.2
    cfg.action_delay = action_delay
    if variant.startswith("shelf"):
        cfg.set_flag(abi.FLAG_CREATE_SHELF, True)
        cfg.set_flag(abi.FLAG_USE_NONZERO_CONTACT_FORCE_RESET, variant == "shelf_contact_reset")
        cfg.min_target_y, cfg.max_target_y = -0.12, -0.02
        cfg.min_target_z, cfg.max_target_z = 0.56, 0.66
        cfg.min_target_depth, cfg.max_target_depth = 0.0, 0.1
    if variant == "pipe":
        cfg.set_flag(abi.FLAG_CREATE_PIPE, True)
        cfg.min_target_y, cfg.max_target_y = -0.3, -0.2
        cfg.min_target_z, cfg.max_target_z = 0.58, 0.67
    return cfg


def random_state(rng, n, cfg=None):
    """A plausible mid-episode state block [VF_COUNT, n] (float64)."""
    st = np.zeros((abi.VF_COUNT, n))
    st[abi.VF_Q0] = rng.uniform(-0.25, 0.25, n)
    st[abi.VF_Q0 + 1:abi.VF_Q0 + 6] = rng.uniform(-0.4, 0.4, (5, n))
    st[abi.VF_QD0] = rng.uniform(-1, 1, n)
    st[abi.VF_QD0 + 1:abi.VF_QD0 + 6] = rng.uniform(-3, 3, (5, n))
    c = cfg or vo.default_config()
    for e in range(n):
        t = vo.tip(c, st[abi.VF_Q0:abi.VF_Q0 + 6, e], st[abi.VF_QD0:abi.VF_QD0 + 6, e])
        st[abi.VF_TIP_Y, e], st[abi.VF_TIP_Z, e], st[abi.VF_TIP_VY, e], st[abi.VF_TIP_VZ, e] = t
    st[abi.VF_CART_Y] = st[abi.VF_Q0]
    st[abi.VF_CART_VY] = st[abi.VF_QD0]
    st[abi.VF_TARGET_Y] = rng.uniform(-0.48, -0.4, n)
    st[abi.VF_TARGET_Z] = rng.uniform(0.58, 0.67, n)
    st[abi.VF_SMOOTHED_U] = rng.uniform(-0.1, 3.0, n)
    st[abi.VF_PREV_CART_VEL] = st[abi.VF_QD0] + rng.uniform(-0.05, 0.05, n)
    st[abi.VF_PREV_CART_VEL_ERR] = rng.uniform(-0.5, 0.5, n)
    st[abi.VF_AGG_REW] = rng.uniform(-5, 5, n)
    st[abi.VF_FIFO0] = rng.uniform(-1, 1, n)
    st[abi.VF_FIFO0 + 1] = rng.uniform(-0.1, 3.0, n)
    st[abi.VF_SHELF_Y] = 0.2
    return st
