"""``Vine5LinkMovingBase`` task: host-side mirror of the reference task class
(isaacgymenvs/tasks/Vine5LinkMovingBase.py:90-1455) on top of the fused HIP step.

The per-step work of the reference's hooks -- ``pre_physics_step`` (V5:922), the 4x
``compute_and_set_dof_actuation_force_tensor`` + ``gym.simulate`` loop (V5:1028; vec_task.py:338-356),
``post_physics_step`` (V5:1110) with ``reset_idx`` (V5:774), ``compute_observations`` (V5:1339) and
``compute_reward`` (V5:1218) -- is one kernel launch behind ``vine_step`` (include/vine.h).
This class keeps the constructor signature, the buffers and the attributes callers read.
"""
import ctypes as C
import logging
import os
from enum import Enum

import torch

from .. import abi, native
from .base.vec_task import VecTask

# Same constants as V5:48-88
NUM_XYZ = 3
NUM_OBJECT_INFO = 2
N_REVOLUTE_DOFS = 5
N_PRISMATIC_DOFS = 1
N_PRESSURE_ACTIONS = 1
INIT_X, INIT_Y, INIT_Z = 0.0, 0.0, 1.0
CART_Z = 0.975  # 1.0 - 0.025 (URDF slider_to_cart origin)

REWARD_NAMES = ["Position", "Const Negative", "Position Success",
                "Velocity Success", "Velocity", "Rail Velocity Control",
                "FPAM Control", "Rail Velocity Change", "FPAM Change", "Rail Limit",
                "Cart Y", "Tip Y", "Contact Force"]
_REWARD_KEYS = ["POSITION", "CONST_NEGATIVE", "POSITION_SUCCESS", "VELOCITY_SUCCESS", "VELOCITY",
                "U_RAIL_VELOCITY_CONTROL", "U_FPAM_CONTROL", "RAIL_VELOCITY_CHANGE", "U_FPAM_CHANGE",
                "RAIL_LIMIT", "CART_Y", "TIP_Y", "CONTACT_FORCE"]


class ObservationType(Enum):
    POS_ONLY = "POS_ONLY"
    POS_AND_VEL = "POS_AND_VEL"
    POS_AND_FD_VEL = "POS_AND_FD_VEL"
    POS_AND_PREV_POS = "POS_AND_PREV_POS"
    POS_AND_FD_VEL_AND_OBJ_INFO = "POS_AND_FD_VEL_AND_OBJ_INFO"
    TIP_AND_CART_AND_OBJ_INFO = "TIP_AND_CART_AND_OBJ_INFO"


def num_observations(observation_type: ObservationType) -> int:
    """V5:152-170."""
    if observation_type == ObservationType.POS_ONLY:
        return N_REVOLUTE_DOFS + N_PRISMATIC_DOFS + NUM_XYZ + NUM_XYZ + N_PRESSURE_ACTIONS + N_PRISMATIC_DOFS
    if observation_type == ObservationType.TIP_AND_CART_AND_OBJ_INFO:
        return 2 * (N_PRISMATIC_DOFS + NUM_XYZ + NUM_XYZ) + N_PRESSURE_ACTIONS + N_PRISMATIC_DOFS + NUM_OBJECT_INFO
    n = 2 * (N_REVOLUTE_DOFS + N_PRISMATIC_DOFS + NUM_XYZ + NUM_XYZ) + N_PRESSURE_ACTIONS + N_PRISMATIC_DOFS
    if observation_type == ObservationType.POS_AND_FD_VEL_AND_OBJ_INFO:
        n += NUM_OBJECT_INFO
    return n


def vine_config_from_cfg(cfg, lib, seed=None):
    """Freeze the task config dict (reference YAML keys, cfg/task/Vine5LinkMovingBase.yaml) into the flat
    ``VineConfig`` handed to the C ABI.  Raises like the reference for what it cannot do."""
    env, sim, task = cfg["env"], cfg["sim"], cfg["task"]
    c = abi.VineConfig()
    native.check(lib.vine_config_default(C.byref(c)), lib)
    observation_type = ObservationType[env["OBSERVATION_TYPE"]]
    scale_observations = bool(env.get("SCALE_OBSERVATIONS", True))
    if scale_observations and abi.OBS_TYPE_BY_NAME[observation_type.value] not in abi.SCALABLE_OBS_TYPES:
        # the reference raises the same for these types whenever SCALE_OBSERVATIONS is on (V5:267-268)
        raise NotImplementedError(f"Observation scaling not implemented for {observation_type}")
    native.check(lib.vine_config_set_obs_type(C.byref(c), abi.OBS_TYPE_BY_NAME[observation_type.value],
                                              int(scale_observations)), lib)
    if not env.get("USE_MOVING_BASE", True):
        raise NotImplementedError("Not implemented for non-moving base")   # V5:898
    c.num_envs = int(env["numEnvs"])
    c.control_freq_inv = int(env.get("controlFrequencyInv", 1))
    c.max_episode_length = int(env["maxEpisodeLength"])
    c.action_delay = int(env.get("ACTION_DELAY", 0))
    c.clip_observations = float(env.get("clipObservations", float("inf")))
    c.clip_actions = float(env.get("clipActions", float("inf")))
    c.dt = float(sim["dt"])
    c.substeps = int(sim.get("substeps", 2))
    gravity = sim.get("gravity", [0.0, 0.0, -9.81])
    if sim.get("up_axis", "z") != "z" or float(gravity[0]) != 0.0 or float(gravity[1]) != 0.0:
        raise ValueError("Vine5LinkMovingBase requires up_axis 'z' and gravity along -z (V5:441)")
    c.gravity = -float(gravity[2])
    for key, field in [("FPAM_MIN", "fpam_min"), ("FPAM_MAX", "fpam_max"), ("RAIL_VELOCITY_SCALE", "rail_velocity_scale"),
                       ("DAMPING", "damping"), ("STIFFNESS", "stiffness"), ("RAIL_SOFT_LIMIT", "rail_soft_limit"),
                       ("RAIL_P_GAIN", "rail_p_gain"), ("RAIL_D_GAIN", "rail_d_gain"),
                       ("RAIL_ACCELERATION", "rail_acceleration"),
                       ("SMOOTHING_ALPHA_INFLATE", "smoothing_alpha_inflate"),
                       ("SMOOTHING_ALPHA_DEFLATE", "smoothing_alpha_deflate"),
                       ("RANDOM_INIT_CART_MIN_Y", "random_init_cart_min_y"),
                       ("RANDOM_INIT_CART_MAX_Y", "random_init_cart_max_y"), ("SUCCESS_DIST", "success_dist"),
                       ("MIN_TARGET_DEPTH_IN_OBSTACLE", "min_target_depth"),
                       ("MAX_TARGET_DEPTH_IN_OBSTACLE", "max_target_depth"),
                       ("MIN_TARGET_Y", "min_target_y"), ("MAX_TARGET_Y", "max_target_y"),
                       ("MIN_TARGET_Z", "min_target_z"), ("MAX_TARGET_Z", "max_target_z")]:
        setattr(c, field, float(env[key]))
    for i, key in enumerate(_REWARD_KEYS):
        c.reward_weights[i] = float(env[key + "_REWARD_WEIGHT"])
    rp = task.get("randomization_parameters", {})
    c.dyn_scale_min = float(rp.get("DYNAMICS_SCALING_MIN", 1.0))
    c.dyn_scale_max = float(rp.get("DYNAMICS_SCALING_MAX", 1.0))
    c.obs_noise_std = float(rp.get("OBSERVATION_NOISE_STD", 0.0))
    c.action_noise_std = float(rp.get("ACTION_NOISE_STD", 0.0))
    for flag, on in [(abi.FLAG_USE_SMOOTHED_FPAM, env.get("USE_SMOOTHED_FPAM", True)),
                     (abi.FLAG_FORCE_U_FPAM, env.get("FORCE_U_FPAM", False)),
                     (abi.FLAG_FORCE_U_RAIL_VELOCITY, env.get("FORCE_U_RAIL_VELOCITY", False)),
                     (abi.FLAG_CREATE_SHELF, env.get("CREATE_SHELF", False)),
                     (abi.FLAG_CREATE_PIPE, env.get("CREATE_PIPE", False)),
                     (abi.FLAG_RANDOMIZE_DOF_INIT, env.get("RANDOMIZE_DOF_INIT", True)),
                     (abi.FLAG_RANDOMIZE_TARGETS, env.get("RANDOMIZE_TARGETS", True)),
                     (abi.FLAG_USE_TARGET_REACHED_RESET, env.get("USE_TARGET_REACHED_RESET", True)),
                     (abi.FLAG_USE_TIP_LIMIT_HIT_RESET, env.get("USE_TIP_LIMIT_HIT_RESET", False)),
                     (abi.FLAG_USE_NONZERO_CONTACT_FORCE_RESET, env.get("USE_NONZERO_CONTACT_FORCE_RESET", False)),
                     (abi.FLAG_VINE_RANDOMIZE, task.get("vine_randomize", False))]:
        c.set_flag(flag, bool(on))
    # physics-model switches of this build (not reference keys; see DESIGN.md "assumptions")
    model = env.get("physicsModel", {})
    c.set_flag(abi.FLAG_STALE_BODY_STATE_AFTER_RESET, bool(model.get("staleBodyStateAfterReset", True)))
    c.set_flag(abi.FLAG_IMPLICIT_JOINT_DAMPING, bool(model.get("implicitJointDamping", True)))
    c.set_flag(abi.FLAG_FPAM_DAMPING_HELD, bool(model.get("fpamDampingHeld", False)))
    c.link_angular_damping = float(model.get("linkAngularDamping", 0.0))
    c.effort_limit = float(model.get("effortLimit", 0.0))
    c.set_flag(abi.FLAG_INTROSPECT, bool(env.get("introspection", False)))
    c.env_id_offset = int(env.get("envIdOffset", 0))
    if seed is not None:
        c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return c


class Vine5LinkMovingBase(VecTask):
    """Drop-in for ``isaacgym_task_map["Vine5LinkMovingBase"]`` (rlgames_utils.py:78-86)."""

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        self.logger = logging.getLogger(__name__)
        self.max_episode_length = self.cfg["env"]["maxEpisodeLength"]
        self.vine_randomize = self.cfg["task"]["vine_randomize"]

        observation_type = ObservationType[self.cfg["env"]["OBSERVATION_TYPE"]]
        self.cfg["env"]["numObservations"] = num_observations(observation_type)
        self.cfg["env"]["numActions"] = N_PRESSURE_ACTIONS + N_PRISMATIC_DOFS
        if self.cfg["env"].get("CREATE_PIPE", False):
            self.logger.info("CREATE_PIPE: the pipe mesh is simulated as its planar cross-section (two walls)")
        if self.cfg["env"].get("CAPTURE_VIDEO", False):
            self.logger.info("CAPTURE_VIDEO is accepted and ignored (no renderer on the target)")

        self._lib = None
        self._handle = None
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)

        self.num_dof = N_REVOLUTE_DOFS + N_PRISMATIC_DOFS
        self.reward_weights = torch.tensor([[self.cfg["env"][k + "_REWARD_WEIGHT"] for k in _REWARD_KEYS]],
                                           device=self.device, dtype=torch.float)
        self.index_to_view = int(0.1 * self.num_envs)
        self.num_steps = 0
        self.dt = self.cfg["sim"]["dt"]
        self.control_dt = self.dt * self.control_freq_inv
        self.obs_scaling = torch.tensor(list(self._vcfg.obs_scaling[:self.num_obs]), device=self.device)
        self.wandb_dict = {}
        self._reward_matrix = None
        self._stats_out = None
        self._introspection = bool(self._vcfg.flags & abi.FLAG_INTROSPECT)
        self.mat = self.read_mat_file(self.cfg["env"]["MAT_FILE"]) if len(self.cfg["env"].get("MAT_FILE", "")) > 0 else None
        # host-indexed state overwrite every step: cannot live inside a captured hipGraph
        self.graph_capturable = self.mat is None
        if self.mat is not None:
            # replay overwrites the DOF state before every step without moving the bodies: the step kernel must keep
            # reading the tip / cart rigid-body states from memory, which it does with introspection on
            self.set_introspection(True)

    # ------------------------------------------------------------------ MAT_FILE replay (V5:281-297, 947-982)
    def read_mat_file(self, filename):
        """Recorded trajectory: cart_pos (1,T), Q (5,T), moving_target_pos (3,T), target_vel, tip_pos (3,T),
        tip_vel (3,T) -> one [T, 12] device table (q(6), target y/z, tip y/z, tip vy/vz)."""
        import numpy as np
        import scipy.io
        mat = scipy.io.loadmat(filename)
        cart, Q = np.asarray(mat["cart_pos"], np.float64), np.asarray(mat["Q"], np.float64)
        T = cart.shape[1]
        assert cart.shape == (1, T) and Q.shape == (N_REVOLUTE_DOFS, T)
        if np.any(np.asarray(mat["target_vel"], np.float64) != 0.0):
            raise NotImplementedError("MAT_FILE with a moving target: target velocities are identically zero in the "
                                      "step kernel (V5:916-918)")
        rows = np.concatenate([cart, Q, np.asarray(mat["moving_target_pos"], np.float64)[1:3],
                               np.asarray(mat["tip_pos"], np.float64)[1:3],
                               np.asarray(mat["tip_vel"], np.float64)[1:3]], 0).T
        self._mat_table = torch.as_tensor(rows, dtype=torch.float32, device=self.device).contiguous()
        return mat

    def overwrite_with_mat(self):
        """V5:947-982: every env is put on sample ``num_steps % T`` of the recording before the step."""
        T = self._mat_table.shape[0]
        index = self.num_steps % T
        self.logger.info(f"Currently at {index} / {T}")
        row = self._mat_table[index]
        st, f = self._state, abi
        st[f.VF_Q0:f.VF_Q0 + 6] = row[0:6].unsqueeze(-1)
        st[f.VF_QD0:f.VF_QD0 + 6] = 0.0
        st[f.VF_TARGET_Y], st[f.VF_TARGET_Z] = row[6], row[7]
        st[f.VF_TIP_Y], st[f.VF_TIP_Z], st[f.VF_TIP_VY], st[f.VF_TIP_VZ] = row[8], row[9], row[10], row[11]

    # ------------------------------------------------------------------ native handle
    def create_sim(self):
        """Replaces create_sim/_create_envs/prepare_sim (V5:364-556): one ``vine_create``."""
        self._lib = native.load()
        seed = self.cfg.get("seed", None)
        self._vcfg = vine_config_from_cfg(self.cfg, self._lib, seed=seed)
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible to PyTorch-ROCm; vine_robot_isaacgymenvs_amd has no CPU path")
        # torch owns the SoA state block so that the reference's state views are zero-copy tensors
        self._state = torch.zeros((abi.VF_COUNT, self.num_envs), device=self.device, dtype=torch.float32)
        h = C.c_void_p()
        native.check(self._lib.vine_create(C.byref(self._vcfg), self.device_id, self._state.data_ptr(), C.byref(h)),
                     self._lib)
        self._handle = h

    def close(self):
        if self._handle is not None and self._lib is not None:
            torch.cuda.synchronize(self.device)
            self._lib.vine_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _native_step(self, actions, obs_out):
        if self.mat is not None:
            self.overwrite_with_mat()
        native.check(self._lib.vine_step(self._handle, actions.data_ptr(), obs_out.data_ptr(), self.rew_buf.data_ptr(),
                                         self.reset_buf.data_ptr(), self.progress_buf.data_ptr(),
                                         self.timeout_buf.data_ptr(), self._stream()), self._lib)
        self.num_steps += 1

    def rollout_step_blocks(self):
        """Rows of the per-workgroup episode sums ``step_rollout_into`` writes (0: this configuration does not run the
        four-lanes-per-env kernel, the fused rollout step is not available)."""
        return 0 if self.mat is not None else int(self._lib.vine_step_rollout_blocks(self._handle))

    def step_rollout_into(self, args, obs_out):
        """One ROLLOUT step in one launch (``vine_step_rollout``, include/vine_ppo.h): the policy head on the LSTM output
        rows in front of the step, the rollout bookkeeping behind it -- an extension for the PPO loop (the trainer's
        ``_rollout_body_fused``); ``VecTask.step`` / ``step_into`` are untouched.  ``args``: abi.RolloutArgs; the
        observation goes to ``obs_out`` and the buffers are re-bound exactly as ``step_into`` does."""
        import ctypes as C
        native.check(self._lib.vine_step_rollout(self._handle, C.addressof(args), obs_out.data_ptr(), self.rew_buf.data_ptr(),
                                                 self.reset_buf.data_ptr(), self.progress_buf.data_ptr(),
                                                 self.timeout_buf.data_ptr(), self._stream()), self._lib)
        self.num_steps += 1
        self.obs_buf = obs_out
        self.obs_dict["obs"] = obs_out.to(self.rl_device)
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        return obs_out

    def reset_idx(self, env_ids):
        """V5:774-839 for callers outside the step (reset_done, V5:715-718)."""
        ids = torch.as_tensor(env_ids, device=self.device).to(torch.long).contiguous()
        if ids.numel() == 0:
            return
        native.check(self._lib.vine_reset_idx(self._handle, ids.data_ptr(), ids.numel(), self.rew_buf.data_ptr(),
                                              self.reset_buf.data_ptr(), self.progress_buf.data_ptr(), self._stream()),
                     self._lib)

    # ------------------------------------------------------------------ metrics side channel (V5:1250-1322)
    def collect_stats(self):
        """The ~120 scalars the reference puts into ``wandb_dict`` EVERY step with one ``.item()`` sync each
        (V5:1250-1322): here ONE two-stage reduction on the device (``vine_stats``, include/vine.h) and ONE
        device->host copy, on demand.  Same key names, so dashboards carry over.  The fields the step only stores on
        request (VINE_FLAG_INTROSPECT) must have been armed before the step being summarised:
        ``bind_reward_matrix()`` (also needed for the per-term entries) or ``set_introspection(True)``."""
        if not self._introspection:
            raise RuntimeError("collect_stats(): call bind_reward_matrix() or set_introspection(True) before the step "
                               "whose state is to be summarised (the step stores the dashboard-only fields on request)")
        f = abi
        if self._stats_out is None:
            self._stats_out = torch.zeros(f.NUM_STATS, device=self.device, dtype=torch.float32)
        native.check(self._lib.vine_stats(self._handle, self.rew_buf.data_ptr(), self.progress_buf.data_ptr(),
                                          int(self.index_to_view), self._stats_out.data_ptr(), self._stream()), self._lib)
        v = self._stats_out.cpu().tolist()           # the only synchronisation
        d = {}
        for name, k in (("dist_tip_to_target", f.VS_DIST_MEAN), ("target_reached", f.VS_TARGET_REACHED),
                        ("limit_hit", f.VS_LIMIT_HIT), ("tip_limit_hit", f.VS_TIP_LIMIT_HIT), ("abs_tip_y", f.VS_ABS_TIP_Y),
                        ("tip_z", f.VS_TIP_Z), ("max_abs_tip_y", f.VS_MAX_ABS_TIP_Y), ("max_tip_z", f.VS_MAX_TIP_Z),
                        ("tip_velocities", f.VS_TIP_VEL_MEAN), ("tip_velocities_max", f.VS_TIP_VEL_MAX),
                        ("u_rail_velocity", f.VS_U_RAIL_ABS), ("prev_u_rail_velocity", f.VS_PREV_U_RAIL_ABS),
                        ("rail_force", f.VS_RAIL_FORCE_ABS), ("u_fpam", f.VS_U_FPAM_ABS),
                        ("smoothed_u_fpam", f.VS_SMOOTHED_ABS),
                        ("tip_target_velocity_difference", f.VS_TIP_VEL_MEAN),      # target velocities are zero (V5:916-918)
                        ("progress_buf", f.VS_PROGRESS_MEAN), ("contact_forces", f.VS_CONTACT_MEAN),
                        ("nonzero_contact_force", f.VS_CONTACT_NONZERO), ("Aggregated Reward", f.VS_AGG_MEAN)):
            d[name] = v[k]
        d["Aggregated Reward 1 Std Up"] = v[f.VS_AGG_MEAN] + v[f.VS_AGG_STD]
        d["Aggregated Reward 1 Std Down"] = v[f.VS_AGG_MEAN] - v[f.VS_AGG_STD]
        w0 = f.VS_VIEW0
        q, qd, pq = v[w0:w0 + 6], v[w0 + 6:w0 + 12], v[w0 + 12:w0 + 18]
        tip_y, tip_z, tip_vy, tip_vz, ptip_y, ptip_z, cart_y, cart_vy, tgt_y, tgt_z = v[w0 + 18:w0 + 28]
        u_fpam, smoothed, u_rail, rail_force, contact = v[f.VS_VIEW_U:f.VS_VIEW_U + 5]
        fd = [(a - b) / self.control_dt for a, b in zip(q, pq)]
        d["prismatic_q0 at self.index_to_view"] = q[0]
        d["prismatic_qd0 at self.index_to_view"] = qd[0]
        d["prismatic_finite_diff_qd0 at self.index_to_view"] = fd[0]
        for j in range(N_REVOLUTE_DOFS):
            d[f"q{j} at self.index_to_view"] = q[1 + j]
            d[f"qd{j} at self.index_to_view"] = qd[1 + j]
            d[f"finite_diff_qd{j} at self.index_to_view"] = fd[1 + j]
        per_dir = {"x": (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
                   "y": (tip_vy, cart_vy, 0.0, (tip_y - ptip_y) / self.control_dt, tip_y, cart_y, tgt_y),
                   "z": (tip_vz, 0.0, 0.0, (tip_z - ptip_z) / self.control_dt, tip_z, CART_Z, tgt_z)}
        for dr, (tv, cv, gv, ftv, tp, cp, gp) in per_dir.items():
            d[f"tip_vel_{dr} at self.index_to_view"] = tv
            d[f"cart_vel_{dr} at self.index_to_view"] = cv
            d[f"target_vel_{dr} at self.index_to_view"] = gv
            d[f"finite_diff_tip_vel_{dr} at self.index_to_view"] = ftv
            d[f"tip_pos_{dr} at self.index_to_view"] = tp
            d[f"cart_pos_{dr} at self.index_to_view"] = cp
            d[f"target_pos_{dr} at self.index_to_view"] = gp
        d["u_fpam at self.index_to_view"] = u_fpam
        d["smoothed u_fpam at self.index_to_view"] = smoothed
        d["u_rail_velocity at self.index_to_view"] = u_rail
        d["rail_force at self.index_to_view"] = rail_force
        d["contact_force at self.index_to_view"] = contact
        d["nonzero_contact_force at self.index_to_view"] = float(contact > 0)
        if self._reward_matrix is not None:
            for k, name in enumerate(REWARD_NAMES):
                mean, mx, mn = v[f.VS_TERM0 + 3 * k:f.VS_TERM0 + 3 * k + 3]
                w = float(self.cfg["env"][_REWARD_KEYS[k] + "_REWARD_WEIGHT"])
                d[f"Mean {name} Reward"] = mean
                d[f"Max {name} Reward"] = mx
                d[f"Weighted Mean {name} Reward"] = w * mean
                d[f"Weighted Max {name} Reward"] = w * mx if w >= 0 else w * mn
        d["Mean Total Reward"] = v[f.VS_REW_MEAN]
        d["Max Total Reward"] = v[f.VS_REW_MAX]
        self.wandb_dict = d
        return d

    @property
    def step_kernel_name(self):
        """Device kernel ``vine_step`` launches for this configuration (one lane per env, or four lanes per env)."""
        return self._lib.vine_step_kernel_name(self._handle).decode()

    def set_introspection(self, on=True):
        """Arm / disarm the stores of the fields nothing in the step reads back (``prev_dof_pos``, ``prev_tip_positions``,
        ``tip_velocities``, ``u_fpam``, ``u_rail_velocity``, ``prev_u_rail_velocity``, ``rail_force``, the mean contact
        force): the reference exposes them as attributes and dashboard inputs; they are ~70 B of the step's HBM traffic
        per env, so the step only stores them on request.  Takes effect with the next step launched outside a captured
        hipGraph (a captured rollout keeps the setting it was captured with)."""
        native.check(self._lib.vine_set_introspection(self._handle, int(bool(on))), self._lib)
        self._introspection = bool(on)

    def observation_names(self):
        """Column names of the default observation layout (V5:1430-1437); generic names for the other layouts."""
        xyz = ("x", "y", "z")
        names = ([f"joint_pos_{i}" for i in range(self.num_dof)] + [f"joint_vel_{i}" for i in range(self.num_dof)]
                 + [f"tip_pos_{i}" for i in xyz] + [f"tip_vel_{i}" for i in xyz] + [f"target_pos_{i}" for i in xyz]
                 + [f"target_vel_{i}" for i in xyz] + ["smoothed_u_fpam", "prev_u_rail_vel", "target_depth", "target_angle"])
        return names if len(names) == self.num_obs else [f"obs_{i}" for i in range(self.num_obs)]

    def write_histograms(self, rows, directory, bins=20):
        """CREATE_HISTOGRAMS_PERIODICALLY (V5:1392-1452) without wandb: one ``.npz`` per histogram set holding the raw
        rows, and per observation column the counts and bin edges a ``wandb.plot.histogram`` would draw."""
        import numpy as np
        os.makedirs(directory, exist_ok=True)
        data = np.asarray(rows, dtype=np.float32)
        out = {"rows": data, "names": np.array(self.observation_names())}
        for j, name in enumerate(self.observation_names()):
            counts, edges = np.histogram(data[:, j], bins=bins)
            out[name + "_counts"], out[name + "_edges"] = counts, edges
        path = os.path.join(directory, f"observation_histograms_{self.num_steps}.npz")
        np.savez_compressed(path, **out)
        self.logger.info(f"Creating histogram at self.num_steps {self.num_steps}: {path}")
        return path

    # ------------------------------------------------------------------ test / tooling hooks
    def bind_reward_matrix(self):
        """Ask the kernel to also write the [N,13] unweighted reward matrix (V5:1272) each step."""
        self._reward_matrix = torch.zeros((self.num_envs, abi.NUM_REWARDS), device=self.device)
        native.check(self._lib.vine_bind_reward_matrix(self._handle, self._reward_matrix.data_ptr()), self._lib)
        self._introspection = True           # the library arms VINE_FLAG_INTROSPECT together with the matrix
        return self._reward_matrix

    def bind_reset_values(self, values):
        """Deterministic reset draws ([N,10], see include/vine.h); ``None`` restores the counter RNG."""
        if values is None:
            self._reset_values = None
            native.check(self._lib.vine_bind_reset_values(self._handle, None), self._lib)
        else:
            self._reset_values = torch.as_tensor(values, dtype=torch.float32, device=self.device).contiguous()
            assert self._reset_values.shape == (self.num_envs, 10)
            native.check(self._lib.vine_bind_reset_values(self._handle, self._reset_values.data_ptr()), self._lib)

    @property
    def step_count(self):
        return self._lib.vine_get_step_count(self._handle)

    @step_count.setter
    def step_count(self, v):
        native.check(self._lib.vine_set_step_count(self._handle, int(v)), self._lib)

    @property
    def state(self):
        """The [VF_COUNT, N] SoA block (fields: include/vine.h VineField)."""
        return self._state

    # ------------------------------------------------------------------ reference attribute names (views)
    def _col(self, f):
        return self._state[f].unsqueeze(-1)

    def _xyz(self, fy, fz, x=0.0):
        return torch.stack([torch.full_like(self._state[fy], x), self._state[fy], self._state[fz]], dim=-1)

    @property
    def dof_pos(self):            # V5:303
        return self._state[abi.VF_Q0:abi.VF_Q0 + 6].t()

    @property
    def dof_vel(self):            # V5:304
        return self._state[abi.VF_QD0:abi.VF_QD0 + 6].t()

    @property
    def prev_dof_pos(self):       # V5:231
        return self._state[abi.VF_PREV_Q0:abi.VF_PREV_Q0 + 6].t()

    @property
    def tip_positions(self):      # V5:357
        return self._xyz(abi.VF_TIP_Y, abi.VF_TIP_Z)

    @property
    def tip_velocities(self):     # V5:361
        return self._xyz(abi.VF_TIP_VY, abi.VF_TIP_VZ)

    @property
    def prev_tip_positions(self):  # V5:232
        return self._xyz(abi.VF_PREV_TIP_Y, abi.VF_PREV_TIP_Z)

    @property
    def cart_positions(self):     # V5:358
        z = torch.full_like(self._state[abi.VF_CART_Y], CART_Z)
        return torch.stack([torch.zeros_like(z), self._state[abi.VF_CART_Y], z], dim=-1)

    @property
    def cart_velocities(self):    # V5:362
        z = torch.zeros_like(self._state[abi.VF_CART_VY])
        return torch.stack([z, self._state[abi.VF_CART_VY], z], dim=-1)

    @property
    def target_positions(self):   # V5:179
        return self._xyz(abi.VF_TARGET_Y, abi.VF_TARGET_Z)

    @property
    def target_velocities(self):  # V5:180 (always zero, V5:916-918)
        return torch.zeros(self.num_envs, NUM_XYZ, device=self.device)

    @property
    def smoothed_u_fpam(self):    # V5:224
        return self._col(abi.VF_SMOOTHED_U)

    @property
    def u_fpam(self):             # V5:937
        return self._col(abi.VF_U_FPAM)

    @property
    def u_rail_velocity(self):    # V5:937
        return self._col(abi.VF_U_RAIL)

    @property
    def prev_u_rail_velocity(self):  # V5:233
        return self._col(abi.VF_PREV_U_RAIL)

    @property
    def prev_cart_vel(self):      # V5:235
        return self._col(abi.VF_PREV_CART_VEL)

    @property
    def prev_cart_vel_error(self):  # V5:234
        return self._col(abi.VF_PREV_CART_VEL_ERR)

    @property
    def rail_force(self):         # V5:1094
        return self._col(abi.VF_RAIL_FORCE)

    @property
    def object_info(self):        # V5:238
        return self._state[abi.VF_OBJ_DEPTH:abi.VF_OBJ_DEPTH + 2].t()

    @property
    def aggregated_rew_buf(self):  # V5:183
        return self._state[abi.VF_AGG_REW]
