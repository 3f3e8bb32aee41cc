"""ctypes mirror of ``include/vine.h`` (declarations only; no library is loaded here).

Every name below restates one declaration of the header.  ``declare(lib)`` attaches
argument/return types to the C-ABI entry points of a loaded shared library and raises
``AttributeError`` if any symbol the header declares is missing.
"""
import ctypes as C

VINE_ABI_VERSION = 3
NUM_LINKS = 5
NUM_DOFS = 6
NUM_ACTIONS = 2
NUM_REWARDS = 13
MAX_OBS = 28
MAX_DELAY = 8

# VineStatus
OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NO_DEVICE, ERR_ALLOC = 0, -1, -2, -3, -4, -5

# VineObsType (reference ObservationType enum, tasks/Vine5LinkMovingBase.py:67-73)
OBS_POS_AND_FD_VEL_AND_OBJ_INFO = 0
OBS_TIP_AND_CART_AND_OBJ_INFO = 1
OBS_POS_ONLY, OBS_POS_AND_VEL, OBS_POS_AND_FD_VEL, OBS_POS_AND_PREV_POS = 2, 3, 4, 5
OBS_TYPE_BY_NAME = {
    "POS_AND_FD_VEL_AND_OBJ_INFO": OBS_POS_AND_FD_VEL_AND_OBJ_INFO,
    "TIP_AND_CART_AND_OBJ_INFO": OBS_TIP_AND_CART_AND_OBJ_INFO,
    "POS_ONLY": OBS_POS_ONLY,
    "POS_AND_VEL": OBS_POS_AND_VEL,
    "POS_AND_FD_VEL": OBS_POS_AND_FD_VEL,
    "POS_AND_PREV_POS": OBS_POS_AND_PREV_POS,
}
SCALABLE_OBS_TYPES = (OBS_POS_AND_FD_VEL_AND_OBJ_INFO, OBS_TIP_AND_CART_AND_OBJ_INFO)

# VINE_FLAG_*
FLAG_USE_SMOOTHED_FPAM = 1 << 0
FLAG_FORCE_U_FPAM = 1 << 1
FLAG_FORCE_U_RAIL_VELOCITY = 1 << 2
FLAG_CREATE_SHELF = 1 << 3
FLAG_RANDOMIZE_DOF_INIT = 1 << 4
FLAG_RANDOMIZE_TARGETS = 1 << 5
FLAG_USE_TARGET_REACHED_RESET = 1 << 6
FLAG_USE_TIP_LIMIT_HIT_RESET = 1 << 7
FLAG_USE_NONZERO_CONTACT_FORCE_RESET = 1 << 8
FLAG_SCALE_OBSERVATIONS = 1 << 9
FLAG_VINE_RANDOMIZE = 1 << 10
FLAG_STALE_BODY_STATE_AFTER_RESET = 1 << 11
FLAG_IMPLICIT_JOINT_DAMPING = 1 << 12
FLAG_FPAM_DAMPING_HELD = 1 << 13
FLAG_CREATE_PIPE = 1 << 14
FLAG_INTROSPECT = 1 << 15

# VineField
VF_Q0 = 0
VF_QD0 = 6
VF_TIP_Y, VF_TIP_Z, VF_TIP_VY, VF_TIP_VZ = 12, 13, 14, 15
VF_CART_Y, VF_CART_VY = 16, 17
VF_TARGET_Y, VF_TARGET_Z = 18, 19
VF_SMOOTHED_U, VF_U_FPAM, VF_U_RAIL, VF_PREV_U_RAIL = 20, 21, 22, 23
VF_PREV_CART_VEL, VF_PREV_CART_VEL_ERR = 24, 25
VF_OBJ_DEPTH, VF_OBJ_ANGLE = 26, 27
VF_AGG_REW = 28
VF_CONTACT, VF_CONTACT_MEAN = 29, 30
VF_SHELF_Y, VF_SHELF_Z = 31, 32
VF_RAIL_FORCE = 33
VF_PREV_Q0 = 34
VF_PREV_TIP_Y, VF_PREV_TIP_Z = 40, 41
VF_FIFO0 = 42
VF_PIPE_Y, VF_PIPE_Z = 42 + 2 * MAX_DELAY, 43 + 2 * MAX_DELAY
VF_COUNT = 44 + 2 * MAX_DELAY

# VineStat (layout of vine_stats' output vector)
NUM_STATS = 128
(VS_DIST_MEAN, VS_TARGET_REACHED, VS_LIMIT_HIT, VS_TIP_LIMIT_HIT, VS_ABS_TIP_Y, VS_TIP_Z, VS_MAX_ABS_TIP_Y, VS_MAX_TIP_Z,
 VS_TIP_VEL_MEAN, VS_TIP_VEL_MAX, VS_U_RAIL_ABS, VS_PREV_U_RAIL_ABS, VS_RAIL_FORCE_ABS, VS_U_FPAM_ABS, VS_SMOOTHED_ABS,
 VS_PROGRESS_MEAN, VS_CONTACT_MEAN, VS_CONTACT_NONZERO, VS_AGG_MEAN, VS_AGG_STD, VS_REW_MEAN, VS_REW_MAX) = range(22)
VS_VIEW0 = 24
VS_VIEW_U = VS_VIEW0 + 28
VS_TERM0 = 64
VS_COUNT_USED = VS_TERM0 + 3 * NUM_REWARDS


class VineConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("num_envs", C.c_int32),
        ("obs_type", C.c_int32),
        ("control_freq_inv", C.c_int32),
        ("substeps", C.c_int32),
        ("max_episode_length", C.c_int32),
        ("action_delay", C.c_int32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
        ("dt", C.c_float),
        ("gravity", C.c_float),
        ("clip_observations", C.c_float),
        ("clip_actions", C.c_float),
        ("fpam_min", C.c_float),
        ("fpam_max", C.c_float),
        ("rail_velocity_scale", C.c_float),
        ("damping", C.c_float),
        ("stiffness", C.c_float),
        ("rail_soft_limit", C.c_float),
        ("rail_p_gain", C.c_float),
        ("rail_d_gain", C.c_float),
        ("rail_acceleration", C.c_float),
        ("smoothing_alpha_inflate", C.c_float),
        ("smoothing_alpha_deflate", C.c_float),
        ("random_init_cart_min_y", C.c_float),
        ("random_init_cart_max_y", C.c_float),
        ("success_dist", C.c_float),
        ("min_target_depth", C.c_float),
        ("max_target_depth", C.c_float),
        ("min_target_y", C.c_float),
        ("max_target_y", C.c_float),
        ("min_target_z", C.c_float),
        ("max_target_z", C.c_float),
        ("reward_weights", C.c_float * NUM_REWARDS),
        ("dyn_scale_min", C.c_float),
        ("dyn_scale_max", C.c_float),
        ("obs_noise_std", C.c_float),
        ("action_noise_std", C.c_float),
        ("cart_mass", C.c_float),
        ("link_mass", C.c_float * NUM_LINKS),
        ("link_inertia", C.c_float * NUM_LINKS),
        ("link_length", C.c_float),
        ("link_com", C.c_float),
        ("joint1_z", C.c_float),
        ("phi0", C.c_float),
        ("link_angular_damping", C.c_float),
        ("fpam_K", C.c_float * NUM_LINKS),
        ("fpam_C", C.c_float * NUM_LINKS),
        ("fpam_b", C.c_float * NUM_LINKS),
        ("fpam_B", C.c_float * NUM_LINKS),
        ("obs_scaling", C.c_float * MAX_OBS),
        ("env_id_offset", C.c_int32),
        ("effort_limit", C.c_float),
    ]

    def set_flag(self, flag, on):
        self.flags = (self.flags | flag) if on else (self.flags & ~flag)

    def has_flag(self, flag):
        return bool(self.flags & flag)


_P = C.POINTER
_H = C.c_void_p  # VineHandle*

# name -> (restype, argtypes); one row per declaration in include/vine.h
PROTOTYPES = {
    "vine_config_default": (C.c_int, [_P(VineConfig)]),
    "vine_config_set_obs_type": (C.c_int, [_P(VineConfig), C.c_int, C.c_int]),
    "vine_num_obs": (C.c_int, [_P(VineConfig)]),
    "vine_create": (C.c_int, [_P(VineConfig), C.c_int, C.c_void_p, _P(_H)]),
    "vine_destroy": (None, [_H]),
    "vine_step": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vine_reset_idx": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vine_bind_reset_values": (C.c_int, [_H, C.c_void_p]),
    "vine_state_ptr": (C.c_void_p, [_H]),
    "vine_get_step_count": (C.c_int64, [_H]),
    "vine_set_step_count": (C.c_int, [_H, C.c_int64]),
    "vine_bind_reward_matrix": (C.c_int, [_H, C.c_void_p]),
    "vine_set_introspection": (C.c_int, [_H, C.c_int]),
    "vine_stats": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "vine_last_error": (C.c_char_p, []),
    "vine_step_kernel_name": (C.c_char_p, [_H]),
    "vine_backend_name": (C.c_char_p, []),
}


_I64 = C.c_int64
_VP = C.c_void_p
PPO_PARTIAL_BLOCKS = 512   # VINE_PPO_PARTIAL_BLOCKS
PPO_LOSS_SCRATCH_FLOATS = 1024 * 32   # VINE_PPO_LOSS_SCRATCH_FLOATS
ROLLOUT_POST_SCRATCH_FLOATS = 1024 * 3    # VINE_ROLLOUT_POST_SCRATCH_FLOATS
RMS_BLOCKS = 128   # VINE_RMS_BLOCKS
class LossFinalize(C.Structure):
    """``VineLossFinalize`` of include/vine_ppo.h (the deferred last step of ``vine_ln_heads_loss``)."""
    _fields_ = [("partial", C.c_void_p), ("blocks", C.c_int32), ("A", C.c_int32), ("n", C.c_int64), ("logstd", C.c_void_p),
                ("critic_coef", C.c_float), ("entropy_coef", C.c_float), ("bounds_coef", C.c_float), ("stats", C.c_void_p),
                ("grad_logstd", C.c_void_p), ("grad_mu_bias", C.c_void_p), ("grad_value_bias", C.c_void_p),
                ("kl_out", C.c_void_p), ("logstd_grad_accum", C.c_void_p), ("loss_scale", C.c_void_p)]


# include/vine_ppo.h (product library only; the oracle does not implement these)
PPO_PROTOTYPES = {
    "vine_column_sums_batched_fin": (C.c_int, [C.c_int32] + [_VP] * 8 + [_VP, _VP, _VP]),
    "vine_lstm_cell_forward": (C.c_int, [_I64, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _I64, _VP, _I64, _VP, _VP, _VP,
                                         _VP, _I64, C.c_int32, _I64, _VP]),
    "vine_lstm_step_mfma": (C.c_int, [_I64, _I64, _I64, _VP, _I64, _VP, _I64, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _VP,
                                      _I64, _VP, _I64, _VP, _VP, _VP, _VP, _I64, _I64, _VP]),
    "vine_lstm_seq_forward_mfma": (C.c_int, [_I64, _I64, _I64, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                             C.c_int32, _VP, _VP, _VP]),
    "vine_lstm_seq_backward_mfma": (C.c_int, [_I64, _I64, _I64] + [_VP] * 8 + [C.c_int32, _VP, C.c_int32, _VP]),
    "vine_lstm_tile_weights": (C.c_int, [_I64, _I64, _VP, _I64, C.c_int32, _VP, _VP]),
    "vine_linear_elu_mfma": (C.c_int, [_I64, _I64, _I64, _VP, _I64, _VP, _I64, _VP, C.c_float, _VP, _I64, _VP]),
    "vine_linear_bwd_elu_mfma": (C.c_int, [_I64, _I64, _I64, _VP, _I64, _VP, _I64, _VP, _I64, C.c_float, _VP, _I64, _VP,
                                           _VP]),
    "vine_lstm_cell_backward": (C.c_int, [_I64, _I64, _VP, _I64, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _I64, _VP,
                                          _I64, _VP, _VP, _VP, C.c_int32, _VP]),
    "vine_lstm_step_backward_mfma": (C.c_int, [_I64, _I64, _VP, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _I64, _VP, _VP, _VP,
                                               _VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP]),
    "vine_weight_grad_mfma": (C.c_int, [_I64, _I64, _I64, _I64, _VP, _I64, _VP, _I64, _I64, _VP, _VP]),
    "vine_mlp3_elu_mfma": (C.c_int, [_I64, _VP, _I64, _VP, _I64, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _I64, _VP, _I64, _VP,
                                     _I64, _VP, _I64, _VP, _I64, C.c_float, _VP, _VP, _VP, _I64, _VP]),
    "vine_lstm_seq_backward_mlp3_mfma": (C.c_int, [_I64, _I64, _I64] + [_VP] * 9 + [_VP, _I64, _VP, _I64, _VP, _I64, _VP, _I64, _VP,
                                                   _VP, C.c_float] + [_VP] * 6 + [_VP]),
    "vine_trunk_phases": (C.c_int, [_VP, _VP]),
    "vine_trunk_args_size": (C.c_int64, []),
    "vine_mlp3_elu_mfma_prep": (C.c_int, [_I64, _VP, _I64, _VP, _I64, _VP, _VP, C.c_float, C.c_float, _VP, _I64, _VP, _I64, _VP,
                                          _I64, _VP, _I64, _VP, _I64, _VP, _I64, C.c_float, _VP, _VP, _VP, _I64,
                                          C.c_int32] + [_VP] * 10 + [_VP]),
    "vine_ln_heads_loss": (C.c_int, [_I64, _I64, C.c_int32, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                     _VP, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, _VP, _VP, C.c_int32,
                                     _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vine_lp16_format": (C.c_char_p, []),
    "vine_ppo_runtime_init": (C.c_int, []),
    "vine_mlp3_elu_f32": (C.c_int, [_I64, _VP, _I64, _VP, _I64, _VP, _VP, C.c_float, C.c_float, _VP, _I64, _VP, _I64, _VP, _I64,
                                    _VP, _I64, _VP, _I64, _VP, _I64, C.c_float, _VP]),
    "vine_mlp3_elu_f32_fin": (C.c_int, [_I64, _VP, _I64, _VP, _I64, _VP, _VP, C.c_float, C.c_float, _VP, _I64, _VP, _I64, _VP, _I64,
                                        _VP, _I64, _VP, _I64, _VP, _I64, C.c_float, _VP, C.c_float, _VP, _VP, C.c_int32, _VP]),
    "vine_mlp3_elu_f32_split": (C.c_int, [_I64, _VP, _I64, _VP, _I64, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _VP, _VP, C.c_float,
                                          C.c_int, _VP, C.c_float, _VP, _VP, C.c_int32, _VP]),
    "vine_mlp3_tile_weights_split": (C.c_int, [_VP, _I64, _I64, _VP, _I64, _VP, _I64, _VP, _VP]),
    "vine_lstm_step_f32": (C.c_int, [_I64, _I64, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _I64, _VP, _VP, _I64, _VP]),
    "vine_lstm_tile_weights_f32": (C.c_int, [_I64, _I64, _VP, _I64, _VP, _VP]),
    "vine_lstm_step_f32_split": (C.c_int, [_I64, _I64, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _I64, _VP, _VP, _I64, C.c_int, _VP]),
    "vine_lstm_tile_weights_split": (C.c_int, [_I64, _I64, _VP, _I64, _VP, _VP]),
    "vine_ln_heads_loss_rows": (C.c_int, []),
    "vine_mlp3_bwd_elu_mfma": (C.c_int, [_I64, _VP, _I64, _I64, _VP, _I64, _VP, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _I64, _I64,
                                         _I64, C.c_float, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vine_weight_grad_group": (C.c_int, [C.c_int32] + [_VP] * 16 + [_VP]),
    "vine_weight_grad_cat_mfma": (C.c_int, [_I64, _I64, _VP, _I64, _VP, _I64, _I64, _I64, _VP, _I64, _I64, _I64, _I64, _I64,
                                            _VP, _VP, _VP]),
    "vine_weight_grad_cat_seq_mfma": (C.c_int, [_I64, _I64, _VP, _I64, _VP, _I64, _I64, _VP, _I64, _VP, _I64, _I64, _I64, _I64,
                                                _VP, _VP, _VP]),
    "vine_layernorm_forward": (C.c_int, [_I64, _I64, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _VP]),
    "vine_layernorm_backward": (C.c_int, [_I64, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vine_layernorm_heads_forward": (C.c_int, [_I64, _I64, _I64, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vine_layernorm_heads_backward": (C.c_int, [_I64, _I64, _I64] + [_VP] * 10),
    "vine_elu_backward": (C.c_int, [_I64, _I64, _VP, _I64, _VP, _I64, C.c_float, _VP, _I64, _VP, C.c_int32, C.c_int32,
                                    _VP]),
    "vine_bias_elu": (C.c_int, [_I64, _I64, _VP, _VP, C.c_float, _VP, _I64, C.c_int32, _VP]),
    "vine_ppo_loss": (C.c_int, [_I64, C.c_int32] + [_VP] * 10 + [C.c_float, C.c_int32, C.c_float, C.c_float, C.c_float,
                                                                C.c_float] + [_VP] * 4 + [_I64, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vine_copy_batched": (C.c_int, [C.c_int32] + [_VP] * 10 + [_VP]),
    "vine_column_sums_batched": (C.c_int, [C.c_int32] + [_VP] * 8 + [_VP, _VP]),
    "vine_column_sums": (C.c_int, [_I64, _I64, _VP, _I64, _VP, _I64, _VP, C.c_int32, _VP]),
    "vine_policy_head": (C.c_int, [_I64, C.c_int32, _I64] + [_VP] * 8 + [C.c_int32, C.c_uint64] + [_VP] * 6 +
                         [_VP, _VP, C.c_float, _VP]),
    "vine_policy_head_rms": (C.c_int, [_I64, C.c_int32, _I64] + [_VP] * 8 + [C.c_float, C.c_uint64] + [_VP] * 6 +
                             [_VP, _VP, C.c_float, _VP]),
    "vine_rollout_post": (C.c_int, [_I64, _I64] + [_VP] * 4 + [C.c_float] * 3 + [_VP] * 7 + [C.c_float, _VP, _VP, _I64,
                                                                                                  C.c_int32, _VP, _VP]),
    "vine_rollout_post_blocks": (C.c_int32, [_I64]),
    "vine_step_rollout": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vine_step_rollout_blocks": (C.c_int32, [_VP]),
    "vine_step_rollout_args_size": (C.c_int32, []),
    "vine_rollout_head_prep": (C.c_int, [_VP] * 9),
    "vine_rollout_finalize": (C.c_int, [_VP, C.c_float, _VP, _VP, C.c_int32, _VP]),
    "vine_rollout_post_defer": (C.c_int, [_I64, _I64] + [_VP] * 4 + [C.c_float] * 3 + [_VP] * 6 + [_VP, _I64, C.c_int32, _VP, _VP]),
    "vine_gae": (C.c_int, [C.c_int32, _I64, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _VP]),
    "vine_dataset_assemble": (C.c_int, [C.c_int32, _I64, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _VP, C.c_float,
                                        C.c_int32, C.c_int32, _VP, _VP, _VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vine_rms_update": (C.c_int, [_I64, _I64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vine_rms_update_multi": (C.c_int, [C.c_int32, _I64, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vine_normalize_obs": (C.c_int, [_I64, _I64, _VP, _VP, _VP, C.c_float, C.c_float, _VP, _I64, C.c_int32, _VP]),
    "vine_adam_step": (C.c_int, [_I64, _VP, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_float, _VP, _VP]),
    "vine_adam_step_sched": (C.c_int, [_I64, _VP, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, _VP, _VP, C.c_float, C.c_float, C.c_float, C.c_float, _VP]),
    "vine_adam_step_amp": (C.c_int, [_I64, _VP, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, _VP, _VP, C.c_float, C.c_float, C.c_float, C.c_float, _VP, _VP, _VP]),
    "vine_adaptive_lr": (C.c_int, [_VP, _VP, C.c_float, C.c_float, C.c_float, C.c_float, _VP]),
}


class RolloutArgs(C.Structure):
    """VineRolloutArgs of include/vine_ppo.h (vine_step_rollout)."""
    _fields_ = ([(k, C.c_void_p) for k in ("y", "hw", "hc", "logstd", "value_mean", "value_var")] +
                [("ln_eps", C.c_float), ("value_eps", C.c_float), ("seed", C.c_uint64), ("counter", C.c_void_p)] +
                [(k, C.c_void_p) for k in ("mu_out", "sigma_out", "value_out", "action_out", "neglogp_out")] +
                [("reward_shift", C.c_float), ("reward_scale", C.c_float), ("gamma_bootstrap", C.c_float), ("reserved", C.c_int32)] +
                [(k, C.c_void_p) for k in ("shaped_out", "dones_out", "cur_rewards", "cur_lengths", "h_state", "c_state", "h_op")] +
                [("h_op_stride", C.c_int64), ("partial", C.c_void_p)])


class TrunkArgs(C.Structure):
    """VineTrunkArgs of include/vine_ppo.h (vine_trunk_phases)."""
    _fields_ = ([("B", C.c_int64), ("T", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64)] +
                [(k, C.c_void_p) for k in ("w_tiled", "bias", "c0", "h0", "done", "h_out", "c_all", "gates", "c_last", "ln_gamma",
                                           "ln_beta")] +
                [("ln_eps", C.c_float), ("clip_value", C.c_int32)] +
                [(k, C.c_void_p) for k in ("w_heads", "b_heads", "logstd", "actions", "old_neglogp", "advantages", "old_values",
                                           "returns", "old_mu", "old_sigma")] +
                [(k, C.c_float) for k in ("e_clip", "critic_coef", "entropy_coef", "bounds_coef", "soft_bound", "alpha")] +
                [(k, C.c_void_p) for k in ("heads", "d_out", "ln_partial", "loss_partial", "stats", "grad_logstd", "grad_mu_bias",
                                           "grad_value_bias", "kl_out", "logstd_grad_accum", "mu_store", "sigma_store",
                                           "loss_scale", "found_inf", "w_hh_tiled", "dgates", "bias_partial")] +
                [("wt0", C.c_void_p), ("ldw0", C.c_int64), ("wt1", C.c_void_p), ("ldw1", C.c_int64), ("wt2", C.c_void_p),
                 ("ldw2", C.c_int64), ("a3", C.c_void_p), ("a3_stride", C.c_int64), ("a2", C.c_void_p), ("a1", C.c_void_p)] +
                [(k, C.c_void_p) for k in ("gz3", "gz2", "gz1", "part3", "part2", "part1")])


def declare_ppo(lib):
    for name, (restype, argtypes) in PPO_PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def declare(lib):
    """Attach prototypes; raises AttributeError naming the first missing symbol."""
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib
