/*
 * vine.h — C ABI of the Vine5LinkMovingBase environment step for MI355X (gfx950).
 *
 * This is the drop-in boundary of the hot path.  The reference has no native
 * code; what it binds at this level is the 13-call Isaac Gym tensor API plus the
 * Python hooks of one task class.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference checkout):
 *
 *   V5 = isaacgymenvs/tasks/Vine5LinkMovingBase.py
 *   VT = isaacgymenvs/tasks/base/vec_task.py
 *   TY = isaacgymenvs/cfg/task/Vine5LinkMovingBase.yaml
 *   URDF = assets/urdf/Vine5LinkMovingBase.urdf
 *
 * Two libraries export this same ABI:
 *   - libvine_hip.so    (vine_robot_isaacgymenvs_amd/csrc)  device_id >= 0, HIP kernels; the product.
 *   - libvine_oracle_*.so (oracle/)                         device_id == -1, CPU; test infrastructure only.
 *
 * Ownership: the caller owns every I/O buffer (actions, obs, rew, reset, progress,
 * timeouts and, optionally, the SoA state block); the library borrows pointers for
 * the duration of a call.  VineConfig is copied at vine_create.
 * Errors: 0 = ok, negative = VineStatus; message via vine_last_error() (thread-local).
 * The library never exits the process (contrast VT:297-299).
 * Streams: every device entry point enqueues on the caller's stream and does not
 * synchronise; `stream` is a hipStream_t passed as void* (NULL = default stream).
 */
#ifndef VINE_H
#define VINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VINE_ABI_VERSION 3         /* 2: VineConfig.env_id_offset, VINE_FLAG_INTROSPECT, vine_set_introspection, vine_stats;
                                      3: VineConfig.effort_limit */
#define VINE_NUM_LINKS 5          /* N_REVOLUTE_DOFS, V5:54 */
#define VINE_NUM_DOFS 6           /* 1 prismatic + 5 revolute, V5:83 */
#define VINE_NUM_ACTIONS 2        /* V5:171 */
#define VINE_NUM_REWARDS 13       /* REWARD_NAMES, V5:78-81 */
#define VINE_MAX_OBS 28
#define VINE_MAX_DELAY 8          /* ACTION_DELAY upper bound supported by the FIFO ring */

typedef enum VineStatus {
    VINE_OK = 0,
    VINE_ERR_INVALID_ARG = -1,
    VINE_ERR_UNSUPPORTED = -2,    /* e.g. CREATE_PIPE, unsupported OBSERVATION_TYPE (V5:268, 1380) */
    VINE_ERR_DEVICE = -3,         /* HIP runtime error */
    VINE_ERR_NO_DEVICE = -4,      /* product library asked to run without a GPU */
    VINE_ERR_ALLOC = -5
} VineStatus;

/* ObservationType, V5:67-73.  The reference can scale only the first two
 * (V5:245-268); the other four work with SCALE_OBSERVATIONS=False and raise
 * NotImplementedError otherwise (V5:267-268): vine_config_set_obs_type returns
 * VINE_ERR_UNSUPPORTED for the same combinations. */
typedef enum VineObsType {
    VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO = 0,   /* 28 columns, V5:1369-1373 */
    VINE_OBS_TIP_AND_CART_AND_OBJ_INFO = 1,     /* 18 columns, V5:1374-1378 */
    VINE_OBS_POS_ONLY = 2,                      /* 14 columns, V5:1354-1356 */
    VINE_OBS_POS_AND_VEL = 3,                   /* 26 columns, simulator velocities, V5:1357-1360 */
    VINE_OBS_POS_AND_FD_VEL = 4,                /* 26 columns, finite-difference velocities, V5:1361-1364 */
    VINE_OBS_POS_AND_PREV_POS = 5               /* 26 columns, previous positions, V5:1365-1368 */
} VineObsType;

/* Boolean switches (TY keys unless stated). */
enum {
    VINE_FLAG_USE_SMOOTHED_FPAM              = 1u << 0,  /* TY:22, V5:1059 */
    VINE_FLAG_FORCE_U_FPAM                   = 1u << 1,  /* TY:25, V5:1023 */
    VINE_FLAG_FORCE_U_RAIL_VELOCITY          = 1u << 2,  /* TY:26, V5:1025 */
    VINE_FLAG_CREATE_SHELF                   = 1u << 3,  /* TY:34 */
    VINE_FLAG_RANDOMIZE_DOF_INIT             = 1u << 4,  /* TY:63, V5:775 */
    VINE_FLAG_RANDOMIZE_TARGETS              = 1u << 5,  /* TY:66, V5:901 */
    VINE_FLAG_USE_TARGET_REACHED_RESET       = 1u << 6,  /* TY:92 */
    VINE_FLAG_USE_TIP_LIMIT_HIT_RESET        = 1u << 7,  /* TY:93 */
    VINE_FLAG_USE_NONZERO_CONTACT_FORCE_RESET= 1u << 8,  /* TY:94 */
    VINE_FLAG_SCALE_OBSERVATIONS             = 1u << 9,  /* TY:97 */
    VINE_FLAG_VINE_RANDOMIZE                 = 1u << 10, /* TY:126 task.vine_randomize */
    /* Physics-model switches (PhysX is closed: see DESIGN.md "assumptions"). */
    VINE_FLAG_STALE_BODY_STATE_AFTER_RESET   = 1u << 11, /* V5:796-797: tip/cart body states are not refreshed by reset_idx */
    VINE_FLAG_IMPLICIT_JOINT_DAMPING         = 1u << 12, /* DOF damping integrated implicitly (articulation drive) instead of explicitly */
    VINE_FLAG_FPAM_DAMPING_HELD              = 1u << 13, /* hold the C*qd torque term over the sim step exactly as V5:1062 does
                                                            (unstable with the reference coefficients in this integrator; default off:
                                                            C joins the implicitly integrated DOF damping) */
    VINE_FLAG_CREATE_PIPE                    = 1u << 14, /* TY:35 CREATE_PIPE: the 13.8 cm-ID tube of assets/urdf/pipe as its planar
                                                            cross-section (two walls), pose and object_info per V5:841-885 */
    VINE_FLAG_INTROSPECT                     = 1u << 15  /* also store the fields the step itself never reads back -- the reference
                                                            exposes them as attributes / dashboard inputs (V5:231-235, 937, 1094,
                                                            1250-1322): VF_PREV_Q0.., VF_PREV_TIP_*, VF_TIP_VY/VZ, VF_U_FPAM, VF_U_RAIL,
                                                            VF_PREV_U_RAIL, VF_RAIL_FORCE, VF_CONTACT_MEAN.  Off by default (they are
                                                            ~70 B of the step's HBM traffic per env); vine_bind_reward_matrix and
                                                            vine_set_introspection arm it, vine_stats needs it. */
};

/* Flat POD configuration = TY `env.*`, `sim.*`, `task.*` + URDF constants. */
typedef struct VineConfig {
    int32_t  abi_version;            /* must be VINE_ABI_VERSION */
    int32_t  num_envs;               /* TY:8 numEnvs */
    int32_t  obs_type;               /* VineObsType, TY:60 */
    int32_t  control_freq_inv;       /* TY:15 (4) */
    int32_t  substeps;               /* TY:104 (10) */
    int32_t  max_episode_length;     /* TY:14 (500) */
    int32_t  action_delay;           /* TY:100 ACTION_DELAY (1) */
    uint32_t flags;                  /* VINE_FLAG_* */
    uint64_t seed;                   /* counter-based RNG key (reference: torch global CPU generator, TR:79) */

    float dt;                        /* TY:103 0.00833 */
    float gravity;                   /* TY:107 9.81 (applied along -z) */
    float clip_observations;         /* TY:11 5.0 */
    float clip_actions;              /* TY:12 1.0 */

    float fpam_min, fpam_max;        /* TY:45-46 */
    float rail_velocity_scale;       /* TY:47 */
    float damping, stiffness;        /* TY:49-50, V5:504, 511 */
    float rail_soft_limit;           /* TY:53 */
    float rail_p_gain, rail_d_gain;  /* TY:56-57 */
    float rail_acceleration;         /* TY:58 */
    float smoothing_alpha_inflate;   /* TY:29 */
    float smoothing_alpha_deflate;   /* TY:30 */
    float random_init_cart_min_y;    /* TY:64 */
    float random_init_cart_max_y;    /* TY:65 */
    float success_dist;              /* TY:68 */
    float min_target_depth, max_target_depth;   /* TY:69-70 */
    float min_target_y, max_target_y;           /* TY:71-72 */
    float min_target_z, max_target_z;           /* TY:73-74 */
    float reward_weights[VINE_NUM_REWARDS];     /* TY:77-89 in REWARD_NAMES order (V5:78-81) */

    float dyn_scale_min, dyn_scale_max;         /* TY:130-131 */
    float obs_noise_std, action_noise_std;      /* TY:133-134 */

    /* Model constants (URDF + V5), filled by vine_config_default. */
    float cart_mass;                 /* URDF:68-70 */
    float link_mass[VINE_NUM_LINKS];     /* URDF:84, 230 */
    float link_inertia[VINE_NUM_LINKS];  /* Ixx about COM, URDF:85, 231 */
    float link_length;               /* joint spacing 0.0885, URDF:296-317 */
    float link_com;                  /* 0.04425, URDF:83 */
    float joint1_z;                  /* world z of first revolute joint: 1.0 - 0.025 - 0.01, V5:85, URDF:274, 288 */
    float phi0;                      /* roll of first joint 3.1415, URDF:288 */
    float link_angular_damping;      /* Isaac Gym AssetOptions.angular_damping; 0 = off (assumption switch) */
    float fpam_K[VINE_NUM_LINKS];    /* V5:1045 */
    float fpam_C[VINE_NUM_LINKS];    /* V5:1046 */
    float fpam_b[VINE_NUM_LINKS];    /* V5:1047 */
    float fpam_B[VINE_NUM_LINKS];    /* V5:1048 */
    float obs_scaling[VINE_MAX_OBS]; /* V5:246-266 (ones when SCALE_OBSERVATIONS is off) */
    int32_t env_id_offset;           /* global id of env 0 of this handle: the counter-based RNG is keyed by
                                        (seed, env_id_offset + env, step, purpose), so a shard of a larger batch (one rank of a
                                        strong-scaling run; the reference shards by giving every rank its own seed, TR:78)
                                        draws exactly what the same envs draw inside the whole batch.  0 by default. */
    float effort_limit;              /* clamp of the five joint efforts handed to the simulator (Isaac Gym clamps
                                        set_dof_actuation_force_tensor to the DOF `effort` property; the URDF has no
                                        <limit effort>: Vine5LinkMovingBase.urdf:278,292).  Applied to the held efforts of
                                        V5:1101-1106, revolute DOFs only; 0 = no clamp (default; assumption switch) */
} VineConfig;

/* Persistent per-env state, struct-of-arrays: field f of env e lives at
 * state[f * num_envs + e] (env-major contiguous per field => coalesced loads). */
typedef enum VineField {
    VF_Q0 = 0,            /* dof_pos[:,0] cart y .. dof_pos[:,5]   (V5:303) */
    VF_QD0 = 6,           /* dof_vel[:,0..5]                       (V5:304) */
    VF_TIP_Y = 12,        /* tip rigid-body position y,z           (V5:357) */
    VF_TIP_Z = 13,
    VF_TIP_VY = 14,       /* tip rigid-body linear velocity        (V5:361) */
    VF_TIP_VZ = 15,
    VF_CART_Y = 16,       /* cart rigid-body position y            (V5:358) */
    VF_CART_VY = 17,      /* cart rigid-body velocity y            (V5:362) */
    VF_TARGET_Y = 18,     /* target_positions[:,1:3]               (V5:179) */
    VF_TARGET_Z = 19,
    VF_SMOOTHED_U = 20,   /* smoothed_u_fpam                       (V5:224) */
    VF_U_FPAM = 21,       /* u_fpam applied this step              (V5:937) */
    VF_U_RAIL = 22,       /* u_rail_velocity applied this step     (V5:937) */
    VF_PREV_U_RAIL = 23,  /* prev_u_rail_velocity                  (V5:233, 945) */
    VF_PREV_CART_VEL = 24,      /* V5:235, 1098 */
    VF_PREV_CART_VEL_ERR = 25,  /* V5:234, 1097 */
    VF_OBJ_DEPTH = 26,    /* object_info[:,0]                      (V5:238, 839) */
    VF_OBJ_ANGLE = 27,    /* object_info[:,1]                      (V5:885; 0 without pipe) */
    VF_AGG_REW = 28,      /* aggregated_rew_buf                    (V5:183, 1278) */
    VF_CONTACT = 29,      /* ||net contact force on shelf_link|| after the last simulate (VT:349-350) */
    VF_CONTACT_MEAN = 30, /* mean over the control_freq_inv sim steps (V5:1243) */
    VF_SHELF_Y = 31,      /* shelf root position y,z               (V5:829-831) */
    VF_SHELF_Z = 32,
    VF_RAIL_FORCE = 33,   /* rail_force of the last actuation      (V5:1094) */
    VF_PREV_Q0 = 34,      /* prev_dof_pos[:,0..5]                  (V5:231, 943) */
    VF_PREV_TIP_Y = 40,   /* prev_tip_positions[:,1:3]             (V5:232, 944) */
    VF_PREV_TIP_Z = 41,
    VF_FIFO0 = 42,        /* actions_history ring: slot s holds (u_rail, u_fpam) at 42+2s, 43+2s (V5:289-291) */
    VF_PIPE_Y = 42 + 2 * VINE_MAX_DELAY,      /* pipe root position y,z (V5:873-874); its angle theta' is VF_OBJ_ANGLE */
    VF_PIPE_Z = 43 + 2 * VINE_MAX_DELAY,
    VF_COUNT = 44 + 2 * VINE_MAX_DELAY
} VineField;

typedef struct VineHandle VineHandle;

/* Fill `cfg` with the defaults of TY + URDF + V5 (vine_randomize=True as in TY:126,
 * CREATE_PIPE treated as False: mesh collision is out of scope). */
int vine_config_default(VineConfig* cfg);

/* Select the observation type and fill cfg->obs_scaling with the reference's per-column
 * constants (V5:241-268); ones when scale_observations == 0. */
int vine_config_set_obs_type(VineConfig* cfg, int obs_type, int scale_observations);

/* Number of observation columns for cfg->obs_type (V5:152-170). */
int vine_num_obs(const VineConfig* cfg);

/* Replaces create_sim/_create_envs/prepare_sim/allocate_buffers/initialize_state_tensors
 * (V5:364-556, 299-362; VT:216-221, 260-283).
 * device_id: HIP device ordinal (product) or -1 (oracle).
 * state_storage: optional caller-owned block of VF_COUNT*num_envs floats on that
 * device (zero-filled by the library); NULL = library allocates and owns it. */
int vine_create(const VineConfig* cfg, int device_id, float* state_storage, VineHandle** out);
void vine_destroy(VineHandle* h);

/* One VecTask.step (VT:319-380) for all envs, fused:
 *   clamp actions (VT:333) -> pre_physics_step (V5:922-945)
 *   -> control_freq_inv x [refresh, actuation (V5:1028-1106), shelf contact norm (VT:348-351), simulate (VT:356)]
 *   -> post_physics_step: progress += 1, reset_idx of flagged envs (V5:1111-1116, 774-839),
 *      compute_observations (V5:1339-1390), compute_reward (V5:1218-1331), compute_reset (V5:1540-1558)
 *   -> timeout_buf (VT:366), clamp obs (VT:374).
 * actions  [N,2]  in
 * obs      [N,num_obs] out, already clamped to +-clip_observations (= obs_dict["obs"])
 * rew      [N]    out (rew_buf)
 * reset    [N]    in/out int64 (reset_buf: flags consumed at the start of the post phase, new flags written)
 * progress [N]    in/out int64 (progress_buf)
 * timeouts [N]    out uint8/bool (extras["time_outs"])
 */
int vine_step(VineHandle* h, const float* actions, float* obs, float* rew,
              int64_t* reset, int64_t* progress, uint8_t* timeouts, void* stream);

/* reset_idx(env_ids) called from outside the step (VT:412-427 reset_done, V5:715-718):
 * same sampling as inside the step; also clears reset/progress/rew of those envs. */
int vine_reset_idx(VineHandle* h, const int64_t* env_ids, int64_t n,
                   float* rew, int64_t* reset, int64_t* progress, void* stream);

/* Deterministic reset values for parity tests: values[e*10 + k] =
 * (q1..q5, cart_y, pipe_depth (CREATE_PIPE; else ignored), target_y, target_z, shelf_depth) used INSTEAD of the
 * counter-based draw for every later reset (the reference draws them from the torch
 * CPU generator, V5:780-788, 904-909, 822-823).  NULL switches back to the RNG.
 * The buffer lives on the handle's device and must stay valid while bound. */
int vine_bind_reset_values(VineHandle* h, const float* values);

/* Borrowed pointer to the SoA state block (VF_COUNT * num_envs floats, on the handle's device). */
float* vine_state_ptr(VineHandle* h);

/* Number of vine_step calls so far (RNG counter / FIFO slot); settable for tests. */
int64_t vine_get_step_count(VineHandle* h);
int vine_set_step_count(VineHandle* h, int64_t step_count);

/* Per-term reward matrix of the last step, [N,13] row-major (reward_matrix of V5:1272; device/host
 * pointer matching the handle).  Optional: pass NULL to vine_bind_reward_matrix to stop writing it. */
int vine_bind_reward_matrix(VineHandle* h, float* reward_matrix);

/* Arm / disarm VINE_FLAG_INTROSPECT for the steps launched from now on (a step already captured in a hipGraph keeps the
 * setting it was captured with).  vine_bind_reward_matrix(non-NULL) arms it too.  Without it the tip / cart rigid-body
 * fields (VF_TIP_*, VF_CART_*) are stored only by envs reset in the step (the step re-derives them from the DOF state):
 * hosts that READ them between steps must arm it.  Switching it ON mid-run makes the NEXT vine_step refresh those fields
 * from the DOF state first (one extra small launch on that step's stream; do the switch outside a graph capture), so the
 * switch does not perturb the trajectory. */
int vine_set_introspection(VineHandle* h, int on);

/* The dashboard scalars of compute_reward (V5:1250-1322: the ~120 `.mean()/.max()/.item()` the reference evaluates every
 * step) from the state the LAST vine_step left behind, as one two-stage reduction on the device (no atomics, no
 * memset: graph-capturable; bit-reproducible).  out[VINE_NUM_STATS] on the handle's device (host memory for the
 * oracle), layout = VineStat below; the per-term reward entries need a bound reward matrix (else they are 0), the
 * introspection-only inputs need VINE_FLAG_INTROSPECT (else VINE_ERR_INVALID_ARG).  rew/progress: the step's
 * rew_buf / progress_buf. */
#define VINE_NUM_STATS 128
typedef enum VineStat {
    VS_DIST_MEAN = 0,          /* dist_tip_to_target                               V5:1250 */
    VS_TARGET_REACHED,         /* mean(dist < SUCCESS_DIST)                        V5:1251 */
    VS_LIMIT_HIT,              /* mean(|cart_y| > RAIL_SOFT_LIMIT)                 V5:1252 */
    VS_TIP_LIMIT_HIT,          /* mean(tip_y < target_y)                           V5:1253 */
    VS_ABS_TIP_Y, VS_TIP_Z, VS_MAX_ABS_TIP_Y, VS_MAX_TIP_Z,                     /* V5:1254-1257 */
    VS_TIP_VEL_MEAN, VS_TIP_VEL_MAX,                                            /* V5:1258-1259 */
    VS_U_RAIL_ABS, VS_PREV_U_RAIL_ABS, VS_RAIL_FORCE_ABS, VS_U_FPAM_ABS, VS_SMOOTHED_ABS,  /* V5:1260-1264 */
    VS_PROGRESS_MEAN,          /* progress_buf                                     V5:1266 */
    VS_CONTACT_MEAN, VS_CONTACT_NONZERO,                                        /* V5:1267-1268 */
    VS_AGG_MEAN, VS_AGG_STD,   /* aggregated_rew_buf mean / unbiased std           V5:1279-1281 */
    VS_REW_MEAN, VS_REW_MAX,   /* Mean / Max Total Reward                          V5:1285-1286 */
    VS_VIEW0 = 24,             /* 28 values of env `index_to_view` (V5:1287-1322): q(6), qd(6), prev_q(6), tip y/z/vy/vz,
                                  prev tip y/z, cart y/vy, target y/z */
    VS_VIEW_U = VS_VIEW0 + 28, /* u_fpam, smoothed, u_rail, rail_force, contact_mean of that env */
    VS_TERM0 = 64,             /* per reward term k = 0..12: mean, max, min of the UNWEIGHTED term at VS_TERM0 + 3k, +1, +2
                                  (V5:1272-1284; the weighted entries follow on the host: w * mean, w >= 0 ? w * max : w * min) */
    VS_COUNT_USED = VS_TERM0 + 3 * VINE_NUM_REWARDS
} VineStat;
int vine_stats(VineHandle* h, const float* rew, const int64_t* progress, int64_t index_to_view, float* out, void* stream);

const char* vine_last_error(void);
/* Which device kernel vine_step launches for this handle's configuration ("vine_step_kernel": one env per lane;
 * "vine_step_quad_kernel": four lanes per env, chosen up to 16384 envs without obstacles) -- for profiles and bench lines;
 * results do not depend on it (tests/test_hip_parity.py runs every case through both). */
const char* vine_step_kernel_name(VineHandle* h);
const char* vine_backend_name(void);   /* "hip-gfx950" or "oracle-f64"/"oracle-f32" */

#ifdef __cplusplus
}
#endif
#endif /* VINE_H */
