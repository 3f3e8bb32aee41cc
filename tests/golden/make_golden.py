#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    python tests/golden/make_golden.py

What it does: registers stand-ins for the third-party packages the reference imports at module
top but that are not installed (isaacgym, gym, wandb, rl_games-free paths only), loads the
reference's ``vec_task.py`` and ``Vine5LinkMovingBase.py`` *from where they lie* under
/root/reference, builds the real task object through its real ``__init__`` on top of a
``FakeGym`` tensor API, and records inputs/outputs of the reference functions as small ``.npz``
files.  Nothing of the reference's source is copied; the fixtures are data.

PhysX is not available: for the step-sequencing fixture (F6) ``FakeGym.simulate`` integrates
the articulation with this repo's CPU oracle (float32 build).  F6 therefore pins the
reference's ORDER of operations (reset-next-step, stale tip, 4x actuation, FIFO, timeouts),
not PhysX numerics.
"""
import importlib.util
import logging
import math
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VINE_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, REPO)

from oracle import vine_oracle as vo  # noqa: E402
from vine_robot_isaacgymenvs_amd import abi  # noqa: E402

N_VINE_BODIES = 8   # slider, cart, link_0..4, tip
VINE_BODY = {"slider": 0, "cart": 1, "link_0": 2, "link_1": 3, "link_2": 4, "link_3": 5, "link_4": 6, "tip": 7}
N_SHELF_BODIES = 2  # shelf, shelf_link
SHELF_BODY = {"shelf": 0, "shelf_link": 1}
DOF_NAMES = ["slider_to_cart", "cart_to_link_0", "link_0_to_link_1", "link_1_to_link_2", "link_2_to_link_3",
             "link_3_to_link_4"]
FLT_MAX = np.finfo(np.float32).max


# --------------------------------------------------------------------------- stand-ins
class _Vec3:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def __add__(self, o):
        return _Vec3(self.x + o.x, self.y + o.y, self.z + o.z)


class _Bag:
    """Attribute bag: any attribute can be set; unknown reads give another bag."""

    def __init__(self, *a, **k):
        self.__dict__.update(k)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        b = _Bag()
        self.__dict__[name] = b
        return b


class _Transform(_Bag):
    def __init__(self, p=None, r=None):
        super().__init__()
        self.p = p or _Vec3()
        self.r = r


class _CameraProperties:
    def __init__(self):
        self.width, self.height = 1600, 900


class _DofType:
    DOF_ROTATION = 1
    DOF_TRANSLATION = 2


def install_stubs():
    if not hasattr(np, "Inf"):          # the reference predates NumPy 2.0 (vec_task.py:102 uses np.Inf)
        np.Inf = np.inf
    gymapi = types.ModuleType("isaacgym.gymapi")
    gymapi.Vec3 = _Vec3
    gymapi.Quat = lambda *a: tuple(a)
    gymapi.Transform = _Transform
    gymapi.SimParams = _Bag
    gymapi.PlaneParams = _Bag
    gymapi.AssetOptions = _Bag
    gymapi.CameraProperties = _CameraProperties
    gymapi.ContactCollection = lambda x: x
    gymapi.DofType = _DofType
    gymapi.DOF_MODE_EFFORT = 3
    gymapi.SIM_PHYSX, gymapi.SIM_FLEX = 1, 0
    gymapi.UP_AXIS_Z, gymapi.UP_AXIS_Y = 1, 0
    gymapi.DOMAIN_ENV, gymapi.DOMAIN_SIM = 1, 0
    gymapi.IMAGE_COLOR = 0
    for k in ["R", "P", "D", "K", "J", "LEFT", "RIGHT", "UP", "DOWN", "H", "C", "ESCAPE", "V"]:
        setattr(gymapi, "KEY_" + k, "KEY_" + k)
    gymapi.acquire_gym = lambda: FakeGym.instance
    gymtorch = types.ModuleType("isaacgym.gymtorch")
    gymtorch.wrap_tensor = lambda t: t
    gymtorch.unwrap_tensor = lambda t: t
    gymutil = types.ModuleType("isaacgym.gymutil")
    torch_utils = types.ModuleType("isaacgym.torch_utils")
    torch_utils.to_torch = lambda x, dtype=torch.float, device="cpu", requires_grad=False: torch.tensor(
        x, dtype=dtype, device=device, requires_grad=requires_grad)

    def _quat_from_angle_axis(angle, axis):      # stand-in for isaacgym.torch_utils (unit axis assumed, xyzw)
        xyz = axis * torch.sin(angle / 2).unsqueeze(-1)
        return torch.cat([xyz, torch.cos(angle / 2).unsqueeze(-1)], dim=-1)

    torch_utils.quat_from_angle_axis = _quat_from_angle_axis
    isaacgym = types.ModuleType("isaacgym")
    isaacgym.gymapi, isaacgym.gymtorch, isaacgym.gymutil, isaacgym.torch_utils = gymapi, gymtorch, gymutil, torch_utils
    sys.modules.update({"isaacgym": isaacgym, "isaacgym.gymapi": gymapi, "isaacgym.gymtorch": gymtorch,
                        "isaacgym.gymutil": gymutil, "isaacgym.torch_utils": torch_utils})

    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")

    class Box:
        def __init__(self, low, high):
            self.low, self.high, self.shape = np.asarray(low), np.asarray(high), np.asarray(low).shape

    spaces.Box = Box
    gym.spaces = spaces
    gym.Space = object
    sys.modules.update({"gym": gym, "gym.spaces": spaces})

    wandb = types.ModuleType("wandb")
    errors = types.ModuleType("wandb.errors")

    class Error(Exception):
        pass

    errors.Error = Error

    def _log(*a, **k):
        raise Error("wandb not initialised")

    wandb.errors, wandb.log, wandb.save = errors, _log, _log
    sys.modules.update({"wandb": wandb, "wandb.errors": errors})

    # synthetic package so that the reference's relative import `.base.vec_task` resolves
    for name in ["isaacgymenvs", "isaacgymenvs.utils", "isaacgymenvs.tasks", "isaacgymenvs.tasks.base"]:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    dr = types.ModuleType("isaacgymenvs.utils.dr_utils")
    for f in ["get_property_setter_map", "get_property_getter_map", "get_default_setter_args", "apply_random_samples",
              "check_buckets", "generate_random_samples"]:
        setattr(dr, f, None)
    sys.modules["isaacgymenvs.utils.dr_utils"] = dr


def load_reference():
    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    vt = load("isaacgymenvs.tasks.base.vec_task", "isaacgymenvs/tasks/base/vec_task.py")
    v5 = load("isaacgymenvs.tasks.Vine5LinkMovingBase", "isaacgymenvs/tasks/Vine5LinkMovingBase.py")
    logging.disable(logging.CRITICAL)
    return vt, v5


# --------------------------------------------------------------------------- fake tensor API
class FakeGym:
    """Minimal Isaac Gym tensor API.  Physics = this repo's float32 oracle (see module docstring)."""

    instance = None

    def __init__(self, num_envs, create_shelf, oracle_cfg, create_pipe=False):
        FakeGym.instance = self
        self.n = num_envs
        self.shelf = create_shelf
        self.pipe = create_pipe
        self.ocfg = oracle_cfg
        self.vine_body_off = (N_SHELF_BODIES if create_shelf else 0) + (1 if create_pipe else 0)
        self.nb = N_VINE_BODIES + self.vine_body_off
        self.dof_state = torch.zeros(num_envs * 6, 2)
        self.rb_state = torch.zeros(num_envs * self.nb, 13)
        self.root_state = torch.zeros(num_envs * (1 + int(create_shelf) + int(create_pipe)), 13)
        self.contact = torch.zeros(num_envs * self.nb, 3)
        self.efforts = torch.zeros(num_envs, 6)
        self.calls = []
        self._actors = 0
        self.apn = 1 + int(create_shelf) + int(create_pipe)     # actors per env: [shelf], [pipe], vine (V5:466-500)
        self.sim_root = torch.zeros(num_envs * self.apn, 13)     # what the SIMULATOR holds: changes only through set_*
        self.sim_root[:, 6] = 1.0
        self._refresh_bodies()

    def __getattr__(self, name):  # every call the fixtures do not care about
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None

    # ---- setup
    def create_sim(self, *a, **k):
        return object()

    def load_asset(self, sim, root, file, opts):
        return file

    def get_asset_dof_count(self, asset):
        return 6

    def get_asset_rigid_body_count(self, asset):
        return N_VINE_BODIES

    def get_asset_dof_type(self, asset, i):
        return _DofType.DOF_TRANSLATION if i == 0 else _DofType.DOF_ROTATION

    def get_asset_dof_name(self, asset, i):
        return DOF_NAMES[i]

    def get_asset_dof_names(self, asset):
        return list(DOF_NAMES)

    def get_dof_type_string(self, t):
        return str(t)

    def get_asset_dof_dict(self, asset):
        return {n: i for i, n in enumerate(DOF_NAMES)}

    def get_asset_rigid_body_dict(self, asset):
        return dict(VINE_BODY)

    def get_asset_joint_dict(self, asset):
        return {n: i for i, n in enumerate(DOF_NAMES)}

    def get_asset_dof_properties(self, asset):
        p = np.zeros(6, dtype=[("lower", np.float32), ("upper", np.float32)])
        p["lower"], p["upper"] = -FLT_MAX, FLT_MAX   # the URDF has no <limit> tags (assumption)
        return p

    def create_env(self, *a):
        return len(self.calls)

    def create_actor(self, env, asset, pose, name, group=0, filter=0, segmentationId=0):
        self._actors += 1
        row = self._actors - 1
        if row < self.sim_root.shape[0]:            # initial pose of the actor (identity orientation unless given)
            self.sim_root[row, 0:3] = torch.tensor([pose.p.x, pose.p.y, pose.p.z])
            self.root_state[row] = self.sim_root[row]
        return 0 if name == "shelf" or not self.shelf else 1

    def get_actor_index(self, env, handle, domain):
        return self._actors - 1

    def get_actor_rigid_shape_properties(self, env, handle):
        return [_Bag(friction=1.0)]

    def get_actor_dof_properties(self, env, handle):
        return np.zeros(6, dtype=[("driveMode", np.int32), ("damping", np.float32), ("stiffness", np.float32)])

    def find_actor_rigid_body_index(self, env, handle, name, domain):
        if name in SHELF_BODY and self.shelf and name not in VINE_BODY:
            return SHELF_BODY[name]
        if name == "base_link":                      # the pipe's only body, created after the shelf
            return N_SHELF_BODIES if self.shelf else 0
        return self.vine_body_off + VINE_BODY[name]

    def create_camera_sensor(self, *a):
        return 0

    # ---- tensors
    def acquire_dof_state_tensor(self, sim):
        return self.dof_state

    def acquire_actor_root_state_tensor(self, sim):
        return self.root_state

    def acquire_rigid_body_state_tensor(self, sim):
        return self.rb_state

    def acquire_net_contact_force_tensor(self, sim):
        return self.contact

    def set_dof_actuation_force_tensor(self, sim, t):
        self.efforts = t.clone()
        self.calls.append("set_efforts")

    def set_dof_state_tensor_indexed(self, sim, state, idx, n):
        self.calls.append("set_dof_state")   # dof_state is live; body states stay stale until simulate

    def set_actor_root_state_tensor_indexed(self, sim, state, idx, n):
        self.calls.append("set_root_state")          # the teleport: only the named actors move, contacts are not touched
        rows = idx.long()
        self.sim_root[rows] = state[rows].clone()

    def _obstacles(self, e):
        """(shelf_y, shelf_z, pipe_y, pipe_z, theta') of env e as the simulator holds them.  The pipe's root orientation
        is a rotation about x by theta = theta' + 90 deg (V5:858-861): quaternion (sin(theta/2), 0, 0, cos(theta/2))."""
        ob = np.zeros(5)
        row = e * self.apn
        if self.shelf:
            ob[0], ob[1] = float(self.sim_root[row, 1]), float(self.sim_root[row, 2])
            row += 1
        if self.pipe:
            ob[2], ob[3] = float(self.sim_root[row, 1]), float(self.sim_root[row, 2])
            ob[4] = 2.0 * math.atan2(float(self.sim_root[row, 3]), float(self.sim_root[row, 6])) - math.pi / 2
        return ob

    def _refresh_bodies(self):
        ds = self.dof_state.view(self.n, 6, 2).numpy().astype(np.float64)
        rb = self.rb_state.view(self.n, self.nb, 13)
        for e in range(self.n):
            t = vo.tip(self.ocfg, ds[e, :, 0], ds[e, :, 1], precision="f32")
            tip, cart = self.vine_body_off + VINE_BODY["tip"], self.vine_body_off + VINE_BODY["cart"]
            rb[e, tip, 1], rb[e, tip, 2], rb[e, tip, 8], rb[e, tip, 9] = t[0], t[1], t[2], t[3]
            rb[e, cart, 1], rb[e, cart, 2], rb[e, cart, 8] = ds[e, 0, 0], 0.975, ds[e, 0, 1]

    def simulate(self, sim):
        """``gym.simulate``: this repo's articulation model.  In the product's default mode (DESIGN.md assumption P4) the
        velocity term of the efforts the reference has just set, -C_j * qd_j (V5:1062), is not held over the sim step:
        it is taken back out of the efforts here (qd has not changed since the reference computed them) and C_j joins the
        implicitly integrated DOF damping -- exactly what vine_step does.  ``VINE_FLAG_FPAM_DAMPING_HELD`` in the
        oracle config selects the literal held form instead."""
        self.calls.append("simulate")
        ds = self.dof_state.view(self.n, 6, 2)
        h = float(np.float32(self.ocfg.dt) / np.float32(self.ocfg.substeps))
        held = self.ocfg.has_flag(abi.FLAG_FPAM_DAMPING_HELD)
        C_ = np.array(list(self.ocfg.fpam_C), np.float32)
        for e in range(self.n):
            qd0 = ds[e, :, 1].numpy().astype(np.float32)
            eff = self.efforts[e].numpy().astype(np.float32).copy()
            cj = None
            if self.ocfg.effort_limit > 0:     # the simulator's DOF effort clamp (assumption switch, include/vine.h)
                lim = np.float32(self.ocfg.effort_limit)
                assert held, "effort clamp fixture: literal mode only"
                eff[1:] = np.clip(eff[1:], -lim, lim)
            if not held:
                eff[1:] = eff[1:] + C_ * qd0[1:]
                cj = np.full(6, np.float32(self.ocfg.damping), np.float32)
                cj[1:] += C_
            if self.shelf or self.pipe:
                # obstacle contacts re-evaluated in every substep (the oracle's own simulate loop); the net contact
                # force tensor then holds this sim step's force on `shelf_link` -- row SHELF_BODY["shelf_link"] of the
                # env's bodies -- until the next simulate (a teleport of the shelf does not clear it)
                q, qd, cf = vo.simulate_obstacles(self.ocfg, ds[e, :, 0].numpy(), ds[e, :, 1].numpy(), eff, h,
                                                  self.ocfg.substeps, self.shelf, self.pipe, self._obstacles(e),
                                                  form=vo.FORM_ABS, precision="f32", cj=cj)
                if self.shelf:
                    self.contact[e * self.nb + SHELF_BODY["shelf_link"]] = torch.tensor([0.0, cf, 0.0])
            else:
                q, qd = vo.simulate(self.ocfg, ds[e, :, 0].numpy(), ds[e, :, 1].numpy(), eff, h,
                                    self.ocfg.substeps, form=vo.FORM_ABS, precision="f32", cj=cj)
            ds[e, :, 0] = torch.from_numpy(q.astype(np.float32))
            ds[e, :, 1] = torch.from_numpy(qd.astype(np.float32))
        self._refresh_bodies()


# --------------------------------------------------------------------------- config
def reference_task_cfg(num_envs, **env_overrides):
    """The reference's task YAML with its interpolations resolved to their defaults."""
    with open(os.path.join(REF, "isaacgymenvs/cfg/task/Vine5LinkMovingBase.yaml")) as f:
        cfg = yaml.safe_load(f)
    e, s, t = cfg["env"], cfg["sim"], cfg["task"]
    cfg["physics_engine"] = "physx"
    e["numEnvs"], e["controlFrequencyInv"] = num_envs, 4
    e["CAPTURE_VIDEO"], e["CREATE_PIPE"] = False, False
    e["RAIL_VELOCITY_SCALE"], e["RAIL_SOFT_LIMIT"], e["RAIL_P_GAIN"], e["RAIL_ACCELERATION"] = 1.0, 0.3, 10.0, 8.0
    e["OBSERVATION_TYPE"] = "POS_AND_FD_VEL_AND_OBJ_INFO"
    e["DAMPING"] = float(e["DAMPING"])                      # PyYAML reads 2e-2 as a string; OmegaConf as float
    e["RANDOM_INIT_CART_MIN_Y"], e["RANDOM_INIT_CART_MAX_Y"] = -0.1 * 0.3, 0.3
    s["use_gpu_pipeline"], s["enable_viewer_sync_at_start"] = False, True
    s["physx"].update(num_threads=4, solver_type=1, use_gpu=False, num_subscenes=4)
    t["vine_randomize"] = False
    for k, v in env_overrides.items():
        if k == "vine_randomize":
            t[k] = v
        elif k in t["randomization_parameters"]:
            t["randomization_parameters"][k] = v
        else:
            e[k] = v
    return cfg


def oracle_cfg_from(cfg, held=False, effort_limit=0.0):
    """VineConfig equivalent of a reference cfg dict (FakeGym physics = the product's default mode unless ``held``)."""
    e, rp = cfg["env"], cfg["task"]["randomization_parameters"]
    c = vo.default_config(num_envs=e["numEnvs"])
    c.max_episode_length = e["maxEpisodeLength"]
    c.action_delay = e["ACTION_DELAY"]
    c.damping, c.stiffness = e["DAMPING"], e["STIFFNESS"]
    c.rail_soft_limit, c.rail_p_gain, c.rail_d_gain = e["RAIL_SOFT_LIMIT"], e["RAIL_P_GAIN"], e["RAIL_D_GAIN"]
    c.rail_acceleration, c.rail_velocity_scale = e["RAIL_ACCELERATION"], e["RAIL_VELOCITY_SCALE"]
    c.fpam_min, c.fpam_max = e["FPAM_MIN"], e["FPAM_MAX"]
    c.success_dist = e["SUCCESS_DIST"]
    c.min_target_y, c.max_target_y = e["MIN_TARGET_Y"], e["MAX_TARGET_Y"]
    c.min_target_z, c.max_target_z = e["MIN_TARGET_Z"], e["MAX_TARGET_Z"]
    c.min_target_depth, c.max_target_depth = e["MIN_TARGET_DEPTH_IN_OBSTACLE"], e["MAX_TARGET_DEPTH_IN_OBSTACLE"]
    c.random_init_cart_min_y, c.random_init_cart_max_y = e["RANDOM_INIT_CART_MIN_Y"], e["RANDOM_INIT_CART_MAX_Y"]
    names = ["POSITION", "CONST_NEGATIVE", "POSITION_SUCCESS", "VELOCITY_SUCCESS", "VELOCITY", "U_RAIL_VELOCITY_CONTROL",
             "U_FPAM_CONTROL", "RAIL_VELOCITY_CHANGE", "U_FPAM_CHANGE", "RAIL_LIMIT", "CART_Y", "TIP_Y", "CONTACT_FORCE"]
    for i, nme in enumerate(names):
        c.reward_weights[i] = e[nme + "_REWARD_WEIGHT"]
    vo.load().vine_config_set_obs_type(c, abi.OBS_TYPE_BY_NAME[e["OBSERVATION_TYPE"]], int(e["SCALE_OBSERVATIONS"]))
    c.set_flag(abi.FLAG_CREATE_SHELF, e["CREATE_SHELF"])
    c.set_flag(abi.FLAG_CREATE_PIPE, e.get("CREATE_PIPE", False))
    c.set_flag(abi.FLAG_USE_TARGET_REACHED_RESET, e["USE_TARGET_REACHED_RESET"])
    c.set_flag(abi.FLAG_USE_TIP_LIMIT_HIT_RESET, e["USE_TIP_LIMIT_HIT_RESET"])
    c.set_flag(abi.FLAG_USE_NONZERO_CONTACT_FORCE_RESET, e["USE_NONZERO_CONTACT_FORCE_RESET"])
    c.set_flag(abi.FLAG_USE_SMOOTHED_FPAM, e["USE_SMOOTHED_FPAM"])
    c.set_flag(abi.FLAG_VINE_RANDOMIZE, cfg["task"]["vine_randomize"])
    c.set_flag(abi.FLAG_FPAM_DAMPING_HELD, held)
    c.effort_limit = effort_limit
    c.dyn_scale_min, c.dyn_scale_max = rp["DYNAMICS_SCALING_MIN"], rp["DYNAMICS_SCALING_MAX"]
    return c


def make_task(vt, v5, cfg, held=False, effort_limit=0.0):
    vt.EXISTING_SIM = None
    ocfg = oracle_cfg_from(cfg, held=held, effort_limit=effort_limit)
    FakeGym(cfg["env"]["numEnvs"], cfg["env"]["CREATE_SHELF"], ocfg, cfg["env"].get("CREATE_PIPE", False))
    task = v5.Vine5LinkMovingBase(cfg=cfg, rl_device="cpu", sim_device="cpu", graphics_device_id=-1, headless=True,
                                  virtual_screen_capture=False, force_render=False)
    return task, FakeGym.instance, ocfg


def npf(t):
    return t.detach().cpu().numpy().copy()


# --------------------------------------------------------------------------- fixtures
def f1_actions_and_actuation(vt, v5, out):
    N = 64
    g = torch.Generator().manual_seed(1234)
    # (a) pre_physics_step over a short action sequence: rescale, FIFO delay, EMA smoothing
    for delay in (0, 1, 3):
        cfg = reference_task_cfg(N, ACTION_DELAY=delay)
        task, gym, _ = make_task(vt, v5, cfg)
        T = 6
        acts = torch.rand(T, N, 2, generator=g) * 2.4 - 1.2
        rec = {k: [] for k in ["u_rail", "u_fpam", "smoothed", "prev_u_rail"]}
        for t in range(T):
            a = torch.clamp(acts[t], -1.0, 1.0)
            task.pre_physics_step(a)
            rec["u_rail"].append(npf(task.u_rail_velocity)); rec["u_fpam"].append(npf(task.u_fpam))
            rec["smoothed"].append(npf(task.smoothed_u_fpam)); rec["prev_u_rail"].append(npf(task.prev_u_rail_velocity))
        out["f1_pre_delay%d" % delay] = dict(actions=npf(acts), **{k: np.stack(v) for k, v in rec.items()})
    # (b) actuation: efforts from random states, vine_randomize off and on (scaling captured by re-seeding)
    for randomize in (False, True):
        cfg = reference_task_cfg(N, vine_randomize=randomize, DYNAMICS_SCALING_MIN=0.9, DYNAMICS_SCALING_MAX=1.1)
        task, gym, _ = make_task(vt, v5, cfg)
        ds = gym.dof_state.view(N, 6, 2)
        ds[:, :, 0] = torch.rand(N, 6, generator=g) * 1.0 - 0.5
        ds[:, :, 1] = torch.rand(N, 6, generator=g) * 6.0 - 3.0
        cart_vy = torch.rand(N, generator=g) * 2 - 1
        rb = gym.rb_state.view(N, gym.nb, 13)
        rb[:, VINE_BODY["cart"], 8] = cart_vy
        task.u_rail_velocity = torch.rand(N, 1, generator=g) * 2 - 1
        task.u_rail_velocity[:8, 0] = cart_vy[:8] + torch.linspace(-0.12, 0.12, 8)   # straddle the |e| > 0.1 switch
        task.u_fpam = torch.rand(N, 1, generator=g) * 3.1 - 0.1
        task.smoothed_u_fpam = torch.rand(N, 1, generator=g) * 3.1 - 0.1
        task.prev_cart_vel = torch.rand(N, 1, generator=g) * 2 - 1
        task.prev_cart_vel_error = torch.rand(N, 1, generator=g) * 2 - 1
        rec = dict(q=npf(ds[:, :, 0]), qd=npf(ds[:, :, 1]), cart_vy=npf(cart_vy), u_rail=npf(task.u_rail_velocity),
                   u_fpam=npf(task.u_fpam), smoothed=npf(task.smoothed_u_fpam), prev_cart_vel=npf(task.prev_cart_vel),
                   prev_cart_vel_err=npf(task.prev_cart_vel_error), dt=np.float64(cfg["sim"]["dt"]))
        if randomize:
            torch.manual_seed(777)
            full = torch.FloatTensor(N, 5, 20).uniform_(0.9, 1.1)
            j = torch.arange(5)
            rec["scale"] = np.concatenate([npf(full[:, j, j + 5 * k]) for k in range(4)], axis=1)  # [N,20]: K,C,b,B
            torch.manual_seed(777)
        task.compute_and_set_dof_actuation_force_tensor()
        rec.update(efforts=npf(gym.efforts), prev_cart_vel_out=npf(task.prev_cart_vel),
                   prev_cart_vel_err_out=npf(task.prev_cart_vel_error), rail_force=npf(task.rail_force))
        out["f1_actuation_rand%d" % int(randomize)] = rec


def f2_observations(vt, v5, out):
    N = 64
    g = torch.Generator().manual_seed(99)
    unscaled = ("POS_ONLY", "POS_AND_VEL", "POS_AND_FD_VEL", "POS_AND_PREV_POS")   # V5:267-268: no scaling constants
    for obs_type in ("POS_AND_FD_VEL_AND_OBJ_INFO", "TIP_AND_CART_AND_OBJ_INFO") + unscaled:
        cfg = reference_task_cfg(N, OBSERVATION_TYPE=obs_type, SCALE_OBSERVATIONS=obs_type not in unscaled)
        task, gym, _ = make_task(vt, v5, cfg)
        ds = gym.dof_state.view(N, 6, 2)
        ds[:, :, 0] = torch.rand(N, 6, generator=g) - 0.5
        rb = gym.rb_state.view(N, gym.nb, 13)
        rb[:, VINE_BODY["tip"], 1:3] = torch.rand(N, 2, generator=g) - 0.5
        if obs_type in unscaled:
            ds[:, :, 1] = (torch.rand(N, 6, generator=g) - 0.5) * 4
            rb[:, VINE_BODY["tip"], 8:10] = (torch.rand(N, 2, generator=g) - 0.5) * 3
            rb[:8, VINE_BODY["tip"], 2] = 7.0      # unscaled columns need a larger value to reach the clamp
        rb[:8, VINE_BODY["tip"], 1] = 2.0          # forces the +-5 clamp after scaling
        task.prev_dof_pos = ds[:, :, 0] - (torch.rand(N, 6, generator=g) - 0.5) * 0.1
        task.prev_tip_positions = task.tip_positions - (torch.rand(N, 3, generator=g) - 0.5) * 0.05
        task.prev_tip_positions[:, 0] = 0
        task.target_positions = torch.rand(N, 3, generator=g) - 0.5
        task.target_positions[:, 0] = 0
        task.smoothed_u_fpam = torch.rand(N, 1, generator=g) * 3
        task.prev_u_rail_velocity = torch.rand(N, 1, generator=g) * 2 - 1
        task.object_info = torch.rand(N, 2, generator=g) - 0.5
        task.object_info[:, 1] = 0
        obs = task.compute_observations()
        out["f2_obs_" + obs_type] = dict(
            q=npf(ds[:, :, 0]), qd=npf(ds[:, :, 1]), tip_vel=npf(task.tip_velocities),
            prev_q=npf(task.prev_dof_pos), tip=npf(task.tip_positions),
            prev_tip=npf(task.prev_tip_positions), target=npf(task.target_positions),
            smoothed=npf(task.smoothed_u_fpam), prev_u_rail=npf(task.prev_u_rail_velocity),
            obj_info=npf(task.object_info), obs=npf(obs), obs_clamped=npf(torch.clamp(obs, -task.clip_obs, task.clip_obs)),
            control_dt=np.float64(task.control_dt), obs_scaling=npf(task.obs_scaling))


def f3_reward(vt, v5, out):
    N = 96
    g = torch.Generator().manual_seed(5)
    names = v5.REWARD_NAMES
    for tag, w in (("default", [0, 0, 1.0, 0, 0.1, 0, 0, 0, 0, 1.0, 0, 0, 0.10]),
                   ("allones", [1.0, 0.5, 0.25, 2.0, 0.1, 0.3, 0.7, 1.1, 0.9, 1.0, 0.6, 0.4, 0.2])):
        dist = torch.rand(N, generator=g) * 0.3
        reached = dist < 0.08
        tip_v = torch.rand(N, 3, generator=g) * 2 - 1
        tip_v[:, 0] = 0
        tgt_v = torch.zeros(N, 3)
        u_rail = torch.rand(N, 1, generator=g) * 2 - 1
        u_fpam = torch.rand(N, 1, generator=g) * 3.1 - 0.1
        prev_u_rail = torch.rand(N, 1, generator=g) * 2 - 1
        smoothed = torch.rand(N, 1, generator=g) * 3.1 - 0.1
        cart_y = torch.rand(N, generator=g) * 0.8 - 0.4
        limit_hit = cart_y.abs() > 0.3
        tip_limit = torch.rand(N, generator=g) > 0.5
        contact = torch.rand(N, generator=g) * 2
        contact[::3] = 0
        total, rm, wrm = v5.compute_reward_jit(dist, reached, tip_v, tgt_v, u_rail, u_fpam, prev_u_rail, smoothed, limit_hit,
                                               tip_limit, cart_y, contact, torch.tensor([w]), names)
        out["f3_reward_" + tag] = dict(dist=npf(dist), reached=npf(reached), tip_v=npf(tip_v), u_rail=npf(u_rail),
                                       u_fpam=npf(u_fpam), prev_u_rail=npf(prev_u_rail), smoothed=npf(smoothed),
                                       cart_y=npf(cart_y), limit_hit=npf(limit_hit), tip_limit=npf(tip_limit),
                                       contact=npf(contact), weights=np.array(w, np.float32), total=npf(total),
                                       matrix=npf(rm), weighted=npf(wrm))


def f4_reset(vt, v5, out):
    rows = []
    for reset_in in (0, 1):
        for prog in (0, 498, 499, 500):
            for bits in range(16):
                reached, limit, tip_limit, contact = [(bits >> k) & 1 for k in range(4)]
                for flags in range(8):
                    f_reach, f_tip, f_contact = [(flags >> k) & 1 for k in range(3)]
                    r = v5.compute_reset_jit(torch.tensor([reset_in]), torch.tensor([prog]), 500,
                                             torch.tensor([bool(reached)]), torch.tensor([bool(limit)]),
                                             torch.tensor([bool(tip_limit)]), torch.tensor([bool(contact)]),
                                             bool(f_reach), bool(f_tip), bool(f_contact))
                    rows.append([reset_in, prog, reached, limit, tip_limit, contact, f_reach, f_tip, f_contact, int(r[0])])
    out["f4_reset_table"] = dict(table=np.array(rows, np.int64), max_episode_length=np.int64(500))


def f5_reset_sampling(vt, v5, out):
    N = 1024
    # CREATE_PIPE (the reference's default obstacle): pose + object_info of V5:841-885
    for shelf in (False, True):
        cfg = reference_task_cfg(N, CREATE_SHELF=shelf, CREATE_PIPE=True)
        task, gym, _ = make_task(vt, v5, cfg)
        torch.manual_seed(43)
        task.reset_idx(torch.arange(N))
        out["f5_reset_pipe_shelf%d" % int(shelf)] = dict(
            q=npf(task.dof_pos), target=npf(task.target_positions), obj_info=npf(task.object_info),
            pipe_root=npf(gym.root_state[task.pipe_indices, 0:7]),
            shelf_root=npf(gym.root_state[task.shelf_indices, 0:3]) if shelf else np.zeros((0, 3), np.float32),
            pipe_radius=np.float64(v5.PIPE_RADIUS))
    for shelf in (False, True):
        cfg = reference_task_cfg(N, CREATE_SHELF=shelf)
        task, gym, _ = make_task(vt, v5, cfg)
        torch.manual_seed(42)
        task.reset_idx(torch.arange(N))
        rec = dict(q=npf(task.dof_pos), qd=npf(task.dof_vel), target=npf(task.target_positions),
                   obj_info=npf(task.object_info), prev_q=npf(task.prev_dof_pos), calls=np.array(gym.calls))
        if shelf:
            rec["shelf_root"] = npf(gym.root_state[task.shelf_indices, 0:3])
        rec["ranges"] = np.array([math.radians(10), cfg["env"]["RANDOM_INIT_CART_MIN_Y"], cfg["env"]["RANDOM_INIT_CART_MAX_Y"],
                                  cfg["env"]["MIN_TARGET_Y"], cfg["env"]["MAX_TARGET_Y"], cfg["env"]["MIN_TARGET_Z"],
                                  cfg["env"]["MAX_TARGET_Z"], cfg["env"]["MIN_TARGET_DEPTH_IN_OBSTACLE"],
                                  cfg["env"]["MAX_TARGET_DEPTH_IN_OBSTACLE"]])
        out["f5_reset_shelf%d" % int(shelf)] = rec


def f6_trajectory(vt, v5, out):
    """The real VecTask.step driven for T steps; every reset's drawn values are recorded."""
    N, T = 8, 64
    # "held_damping008": the reference's LITERAL actuation semantics -- the efforts of V5:1062, velocity term C_j*qd_j
    # included, held over the whole sim step -- at the DAMPING for which this articulation model is stable under them
    # (0.08); "held_efflim03": the same literal semantics at the YAML's own DAMPING (0.02), bounded by
    # a simulator-side joint effort clamp of 0.3 N m.  The other three run the product's default mode at the YAML's DAMPING.
    # "shelf_*" / "pipe_delay1" (round 4): BASELINE configs[4]'s step sequencing and the task YAML's default obstacle through
    # the real VecTask.step -- FakeGym.simulate re-evaluates the obstacle contacts per substep and leaves the sim step's
    # force on `shelf_link` in the net-contact-force tensor, set_actor_root_state_tensor_indexed teleports the obstacle:
    # pins that entry 0 of a step's four contact norms is the PREVIOUS step's last sim step (VT:343-351 reads the tensor
    # before each simulate), that a teleport inside reset_idx does not clear it (V5:816-839), the mean (V5:1240-1244), and
    # the contact term of reward / reset.  Targets and depths are chosen so that the vine meets the front-edge strip.
    shelf_over = dict(CREATE_SHELF=True, ACTION_DELAY=1, MIN_TARGET_Y=-0.12, MAX_TARGET_Y=-0.02, MIN_TARGET_Z=0.56,
                      MAX_TARGET_Z=0.66, MIN_TARGET_DEPTH_IN_OBSTACLE=0.0, MAX_TARGET_DEPTH_IN_OBSTACLE=0.1)
    for tag, over in (("delay1", dict(ACTION_DELAY=1)), ("delay0_tipobs", dict(ACTION_DELAY=0, OBSERVATION_TYPE="TIP_AND_CART_AND_OBJ_INFO")),
                      ("delay2", dict(ACTION_DELAY=2)), ("held_damping008", dict(ACTION_DELAY=1, DAMPING=0.08)),
                      ("held_efflim03", dict(ACTION_DELAY=1)),
                      ("shelf_delay1", dict(shelf_over, USE_NONZERO_CONTACT_FORCE_RESET=False)),
                      ("shelf_contact_reset", dict(shelf_over, USE_NONZERO_CONTACT_FORCE_RESET=True)),
                      ("pipe_delay1", dict(CREATE_PIPE=True, ACTION_DELAY=1, MIN_TARGET_Y=-0.3, MAX_TARGET_Y=-0.2,
                                           MIN_TARGET_Z=0.58, MAX_TARGET_Z=0.67))):
        env_over = dict(maxEpisodeLength=20, SUCCESS_DIST=0.12, RAIL_SOFT_LIMIT=0.2, MIN_TARGET_Y=-0.3,
                        MAX_TARGET_Y=-0.1, MIN_TARGET_Z=0.53, MAX_TARGET_Z=0.6, RANDOM_INIT_CART_MIN_Y=-0.02,
                        RANDOM_INIT_CART_MAX_Y=0.2)
        env_over.update(over)
        obstacle = env_over.get("CREATE_SHELF", False) or env_over.get("CREATE_PIPE", False)
        cfg = reference_task_cfg(N, **env_over)
        task, gym, ocfg = make_task(vt, v5, cfg, held=tag.startswith("held"), effort_limit=0.3 if "efflim03" in tag else 0.0)
        torch.manual_seed(42)
        g = torch.Generator().manual_seed(2024)
        first = task.reset()
        rec = {k: [] for k in ["actions", "obs", "rew", "reset", "timeouts", "progress", "reset_values", "did_reset", "q",
                               "qd", "tip", "agg"] + (["contact_norms", "contact_mean", "obstacle_pose"] if obstacle else [])}
        rec_first = npf(first["obs"])
        for t in range(T):
            a = torch.rand(N, 2, generator=g) * 2.6 - 1.3           # exceeds +-1: exercises the action clamp
            if t % 7 < 3:
                a[:, 0] = 1.3 if (t // 7) % 2 == 0 else -1.3        # push the cart towards the soft limit
            will_reset = npf(task.reset_buf).astype(bool)
            obs, rew, rst, extras = task.step(a)
            vals = np.zeros((N, 10), np.float32)
            vals[:, 0:5] = npf(task.dof_pos)[:, 1:6]; vals[:, 5] = npf(task.dof_pos)[:, 0]
            vals[:, 6:9] = npf(task.target_positions); vals[:, 9] = npf(task.object_info)[:, 0]
            if env_over.get("CREATE_PIPE", False):
                vals[:, 6] = npf(task.object_info)[:, 0]            # the pipe's entrance depth (V5:884); slot 6 is target x = 0 otherwise
            # dof_pos is the post-reset draw only for the envs that were reset inside this step
            vals[~will_reset] = 0
            if obstacle:
                # the four norms VT:349-351 collected in this step (zeros without a shelf: V5:1246-1248), their mean as
                # compute_reward forms it (V5:1242-1243), and where the simulator holds the obstacles after the step
                norms = (torch.stack(task.shelf_contact_force_norms, dim=0) if cfg["env"]["CREATE_SHELF"]
                         else torch.zeros(4, N))
                rec["contact_norms"].append(npf(norms)); rec["contact_mean"].append(npf(torch.mean(norms, dim=0)))
                rec["obstacle_pose"].append(np.stack([gym._obstacles(e) for e in range(N)]).astype(np.float32))
            rec["actions"].append(npf(a)); rec["obs"].append(npf(obs["obs"])); rec["rew"].append(npf(rew))
            rec["reset"].append(npf(rst)); rec["timeouts"].append(npf(extras["time_outs"]))
            rec["progress"].append(npf(task.progress_buf)); rec["reset_values"].append(vals)
            rec["did_reset"].append(will_reset); rec["q"].append(npf(task.dof_pos)); rec["qd"].append(npf(task.dof_vel))
            rec["tip"].append(npf(task.tip_positions)); rec["agg"].append(npf(task.aggregated_rew_buf))
        n_sim = gym.calls.count("simulate")
        assert n_sim == 4 * T, n_sim
        d = {k: np.stack(v) for k, v in rec.items()}
        d.update(first_obs=rec_first, env_overrides=np.array(sorted("%s=%r" % kv for kv in env_over.items())))
        out["f6_traj_" + tag] = d
        if tag == "delay1":
            # F7: the dashboard scalars the reference puts into wandb_dict every step (V5:1250-1322): key names and,
            # for the last step of this trajectory, the values computed from the state the fixture already records
            keys = sorted(task.wandb_dict.keys())
            out["f7_wandb_keys"] = dict(keys=np.array(keys),
                                        values=np.array([float(task.wandb_dict[k]) for k in keys], np.float64),
                                        index_to_view=np.int64(task.index_to_view))


def f8_ppo_loss_terms(out):
    """F8: the PPO loss terms as the reference's in-tree text states them (isaacgymenvs/learning/common_agent.py:
    _actor_loss 482-501, _critic_loss 503-516, bound_loss 427-435).  rl_games (the code that actually runs) is not
    installed; these three methods are lifted out of the reference file by name and run on a stand-in ``self``."""
    import ast
    import types
    path = os.path.join(REF, "isaacgymenvs/learning/common_agent.py")
    tree = ast.parse(open(path).read())
    wanted = {"_actor_loss", "_critic_loss", "bound_loss"}
    funcs = [n for cls in tree.body if isinstance(cls, ast.ClassDef) for n in cls.body
             if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {f.name for f in funcs} == wanted
    ns = {"torch": torch}
    exec(compile(ast.Module(body=funcs, type_ignores=[]), path, "exec"), ns)
    agent = types.SimpleNamespace(ppo=True, bounds_loss_coef=0.0001, ppo_device="cpu")
    g = torch.Generator().manual_seed(808)
    n, A = 512, 2
    old_nlp = torch.randn(n, generator=g) * 0.5 + 2.0
    nlp = old_nlp + torch.randn(n, generator=g) * 0.3
    adv = torch.randn(n, generator=g)
    old_v, v, ret = torch.randn(n, 1, generator=g), torch.randn(n, 1, generator=g), torch.randn(n, 1, generator=g)
    v[:64] = old_v[:64] + 0.5                      # beyond the clip range
    mu = torch.randn(n, A, generator=g) * 0.9      # a good share beyond +-1
    e_clip = 0.2
    a = ns["_actor_loss"](agent, old_nlp, nlp, adv, e_clip)
    c = ns["_critic_loss"](agent, old_v, v, e_clip, ret, True)
    c_noclip = ns["_critic_loss"](agent, old_v, v, e_clip, ret, False)
    b = ns["bound_loss"](agent, mu)
    out["f8_ppo_loss_terms"] = dict(old_neglogp=npf(old_nlp), neglogp=npf(nlp), advantage=npf(adv), old_values=npf(old_v),
                                    values=npf(v), returns=npf(ret), mu=npf(mu), e_clip=np.float64(e_clip),
                                    a_loss=npf(a["actor_loss"]), clip_frac=npf(a["actor_clip_frac"]),
                                    c_loss=npf(c["critic_loss"]), c_loss_noclip=npf(c_noclip["critic_loss"]),
                                    b_loss_soft_bound_1=npf(b))


def f9_gae(out):
    """F9: generalised advantage estimation as the reference's in-tree text states it (isaacgymenvs/learning/
    common_agent.py:413-425, ``discount_values``: lifted out of the file by name like F8's loss terms).  It takes the
    terminal flags AFTER each step and next values already masked by them; rl_games' own form (next-value /
    next-nonterminal, the one this build implements) is the same recurrence on dones[t + 1] / values[t + 1]: the fixture
    stores the T + 1 flags and values both forms are fed from."""
    import ast
    import types
    path = os.path.join(REF, "isaacgymenvs/learning/common_agent.py")
    tree = ast.parse(open(path).read())
    funcs = [n for cls in tree.body if isinstance(cls, ast.ClassDef) for n in cls.body
             if isinstance(n, ast.FunctionDef) and n.name == "discount_values"]
    assert len(funcs) == 1
    ns = {"torch": torch}
    exec(compile(ast.Module(body=funcs, type_ignores=[]), path, "exec"), ns)
    T, N = 16, 64
    agent = types.SimpleNamespace(horizon_length=T, gamma=0.99, tau=0.95)
    g = torch.Generator().manual_seed(909)
    rewards = torch.randn(T, N, 1, generator=g)
    values = torch.randn(T + 1, N, 1, generator=g)                    # V_0 .. V_T (V_T = the bootstrap "last values")
    dones = (torch.rand(T + 1, N, generator=g) < 0.15).float()        # dones[t]: the episode ended at step t - 1
    dones[:, :4] = 0.0                                                # some envs never finish
    dones[5, 4:8] = 1.0
    fdones = dones[1:]                                                # flags after step t
    next_values = values[1:] * (1.0 - fdones).unsqueeze(2)
    advs = ns["discount_values"](agent, fdones, values[:T], rewards, next_values)
    out["f9_gae"] = dict(rewards=npf(rewards), values=npf(values), dones=npf(dones), advs=npf(advs),
                         gamma=np.float64(0.99), tau=np.float64(0.95))


def f10_rollout_bookkeeping(out):
    """F10: the per-step bookkeeping of the rollout as the reference's in-tree text states it (isaacgymenvs/learning/
    common_agent.py:257-316, ``play_steps``, lifted by name and run for T steps on a stand-in ``self`` whose env, policy
    and critic replay prepared tensors): shaped rewards into the buffer, done flags, the running episode return / length
    accumulators and their masking by ``not_dones``, and WHICH finished episodes' returns / lengths are handed to the
    ``game_rewards`` / ``game_lengths`` meters at every step (the stand-in meters record the arguments of ``update``).
    rl_games' own ``play_steps_rnn`` (absent) keeps the same books; the windowed mean inside its AverageMeter is not
    part of this text and stays unpinned."""
    import ast
    import types
    path = os.path.join(REF, "isaacgymenvs/learning/common_agent.py")
    tree = ast.parse(open(path).read())
    wanted = {"play_steps", "discount_values"}
    funcs = [n for cls in tree.body if isinstance(cls, ast.ClassDef) for n in cls.body
             if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {f.name for f in funcs} == wanted
    ns = {"torch": torch, "a2c_common": types.SimpleNamespace(swap_and_flatten01=lambda x: x)}
    exec(compile(ast.Module(body=funcs, type_ignores=[]), path, "exec"), ns)
    T, N, scale = 12, 96, 0.01
    g = torch.Generator().manual_seed(1010)
    rewards = torch.randn(T, N, 1, generator=g) * 3.0 + 1.0
    dones = (torch.rand(T, N, generator=g) < 0.2)
    dones[:, :6] = False                                              # some envs never finish
    dones[3, 6:40] = True                                             # a step where a third of the envs finish together
    dones[7] = False                                                  # a step where none does
    values = torch.randn(T, N, 1, generator=g)
    snap = {"cur_r": [], "cur_l": [], "upd_r": [], "upd_l": []}

    class Meter:
        def __init__(self, key):
            self.key = key

        def update(self, x):
            snap[self.key].append(x.clone())

    class Buffer:
        def __init__(self):
            self.tensor_dict = {}

        def update_data(self, name, n, val):
            self.tensor_dict.setdefault(name, [None] * T)[n] = val.clone().float() if torch.is_tensor(val) else val

        def get_transformed_list(self, fn, names):
            return {}

    agent = types.SimpleNamespace(
        horizon_length=T, gamma=0.99, tau=0.95, num_agents=1, batch_size=T * N, use_action_masks=False, has_central_value=False,
        update_list=["values"], tensor_list=[], obs={"obs": torch.zeros(N, 4)}, dones=torch.zeros(N, dtype=torch.uint8),
        current_rewards=torch.zeros(N, 1), current_lengths=torch.zeros(N), experience_buffer=Buffer(),
        game_rewards=Meter("upd_r"), game_lengths=Meter("upd_l"), rewards_shaper=lambda r: r * scale,
        algo_observer=types.SimpleNamespace(process_infos=lambda infos, idx: None), set_eval=lambda: None, step=[0])

    def env_reset_done():
        if agent.step[0] > 0:                                         # accumulators as the previous step left them
            snap["cur_r"].append(agent.current_rewards.clone())
            snap["cur_l"].append(agent.current_lengths.clone())
        return agent.obs, []

    def env_step(actions):
        n = agent.step[0]
        agent.step[0] += 1
        return agent.obs, rewards[n].clone(), dones[n].to(torch.uint8), {"terminate": torch.zeros(N)}

    agent._env_reset_done = env_reset_done
    agent.env_step = env_step
    agent.get_action_values = lambda obs: {"actions": torch.zeros(N, 2), "values": values[agent.step[0]]}
    agent._eval_critic = lambda obs: torch.zeros(N, 1)
    agent.discount_values = lambda *a: ns["discount_values"](agent, *a)
    # the buffer's lists become tensors where play_steps reads them back
    orig_update = agent.experience_buffer.update_data
    ns["play_steps"].__globals__["a2c_common"] = ns["a2c_common"]

    class Dict(dict):
        def __getitem__(self, k):
            v = dict.__getitem__(self, k)
            return torch.stack(v) if isinstance(v, list) else v

    agent.experience_buffer.tensor_dict = Dict()
    ns["play_steps"](agent)
    snap["cur_r"].append(agent.current_rewards.clone())
    snap["cur_l"].append(agent.current_lengths.clone())
    td = agent.experience_buffer.tensor_dict
    assert len(snap["cur_r"]) == T and len(snap["upd_r"]) == T
    # ragged lists of finished episodes -> (sum, count) per step, the form a device-side meter update consumes
    out["f10_rollout_bookkeeping"] = dict(
        rewards=npf(rewards), dones=dones.numpy().astype(np.uint8), reward_scale=np.float64(scale),
        shaped=npf(td["rewards"]), buffer_dones=npf(td["dones"]),
        cur_rewards=npf(torch.stack(snap["cur_r"])), cur_lengths=npf(torch.stack(snap["cur_l"])),
        finished_return_sum=np.array([float(x.double().sum()) for x in snap["upd_r"]]),
        finished_length_sum=np.array([float(x.double().sum()) for x in snap["upd_l"]]),
        finished_count=np.array([int(x.shape[0]) for x in snap["upd_r"]], np.int64))


def main():
    install_stubs()
    vt, v5 = load_reference()
    out = {}
    f1_actions_and_actuation(vt, v5, out)
    f2_observations(vt, v5, out)
    f3_reward(vt, v5, out)
    f4_reset(vt, v5, out)
    f5_reset_sampling(vt, v5, out)
    f6_trajectory(vt, v5, out)       # also writes F7 (wandb_dict keys)
    f8_ppo_loss_terms(out)
    f9_gae(out)
    f10_rollout_bookkeeping(out)
    for name, d in out.items():
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print("%-40s %7.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
