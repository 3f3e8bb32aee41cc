/*
 * vine_oracle.c — CPU ORACLE for the Vine5LinkMovingBase env step.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  The product
 * (vine_robot_isaacgymenvs_amd/) never does; it fails loudly without its HIP
 * extension.
 *
 * Parity status
 *   - glue (actions, FPAM torque model, rail controller, observations, reward,
 *     reset logic, step ordering): a restatement of the reference's Python,
 *     pinned by golden vectors generated from the reference's own code
 *     (tests/golden/, made by tests/golden/make_golden.py).
 *   - rigid-body physics (row P1): the reference delegates to NVIDIA Isaac Gym
 *     Preview 4 / PhysX (closed binary, not vendored, not installed).  The model
 *     here is derived from the URDF and the sim block of the task YAML;
 *     **parity with PhysX is unpinned**.  It is pinned against itself: three
 *     independent formulations (Jacobian CRBA in relative coordinates, planar
 *     Featherstone ABA, closed-form absolute-angle Lagrangian) must agree, and
 *     energy / equilibrium invariants must hold (tests/test_oracle_physics.py).
 *
 * Citations: V5 = isaacgymenvs/tasks/Vine5LinkMovingBase.py,
 *            VT = isaacgymenvs/tasks/base/vec_task.py,
 *            TY = isaacgymenvs/cfg/task/Vine5LinkMovingBase.yaml,
 *            URDF = assets/urdf/Vine5LinkMovingBase.urdf  (all under the reference root).
 *
 * Build: oracle/Makefile compiles this file twice, -DVINE_REAL=double and
 * -DVINE_REAL=float, into oracle/_build/libvine_oracle_f64.so / _f32.so.
 */
#include "../include/vine.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef VINE_REAL
#define VINE_REAL double
#endif
typedef VINE_REAL real;

#define NL VINE_NUM_LINKS
#define ND VINE_NUM_DOFS

/* Flop instrumentation (-DVINE_COUNT_FLOPS, oracle/Makefile `flops`): every arithmetic block of the step path adds its
 * operation count (+, -, *, / and each sin / cos / sqrt / log / exp counted as ONE) to one of three buckets as it runs,
 * so loops, branches (randomisation, resets) and the solver's triangular loops are counted as executed:
 *   bucket 0 = one substep (forward dynamics + integration), 1 = one actuation call, 2 = everything else of a step.
 * vine_oracle_flop_counters() returns the totals and the number of substeps / actuation calls / env steps seen. */
#ifdef VINE_COUNT_FLOPS
static double g_flops[3], g_calls[3];
static int g_bucket = 2;
#define FLOPS(n) (g_flops[g_bucket] += (n))
#define FLOP_BUCKET(b) (g_bucket = (b))
#define FLOP_CALL(b) (g_calls[b] += 1)
void vine_oracle_flop_counters(double* flops3, double* calls3, int reset) {
    for (int i = 0; i < 3; ++i) { flops3[i] = g_flops[i]; calls3[i] = g_calls[i]; if (reset) { g_flops[i] = 0; g_calls[i] = 0; } }
}
#else
#define FLOPS(n) ((void)0)
#define FLOP_BUCKET(b) ((void)0)
#define FLOP_CALL(b) ((void)0)
#endif

static __thread char g_err[256];
static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
const char* vine_last_error(void) { return g_err; }
const char* vine_step_kernel_name(VineHandle* h) { (void)h; return "oracle"; }
const char* vine_backend_name(void) { return sizeof(real) == 8 ? "oracle-f64" : "oracle-f32"; }
int vine_oracle_real_bytes(void) { return (int)sizeof(real); }
/* threads used by vine_step in the -fopenmp build (cpu_baseline leg of bench.py); returns the count in force */
int vine_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Counter-based RNG: Philox4x32-10 (Salmon et al. 2011).  The reference draws
 * from torch's global CPU generator (V5:780-788, 904-909, 931, 1054, 1389);
 * streams cannot match, so the build defines its own keyed by
 * (seed, env, step, purpose) and the product uses the same function. */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* exported for tests (known-answer vectors of the Random123 distribution) */
void vine_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}
static inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); } /* [0,1) */

/* purposes (counter word 2) */
enum { RNG_RESET = 1, RNG_ACTION_NOISE = 2, RNG_DYN_SCALE = 3, RNG_OBS_NOISE = 4 };

static void rng4(uint64_t seed, uint32_t env, uint64_t step, uint32_t purpose, uint32_t idx, uint32_t out[4]) {
    /* env = global env id (VineConfig.env_id_offset + local index) */
    philox4x32_10(env, (uint32_t)step, purpose | ((uint32_t)(step >> 32) << 8), idx,
                  (uint32_t)seed, (uint32_t)(seed >> 32), out);
}
static void normal2(uint32_t a, uint32_t b, float* n0, float* n1) { /* Box-Muller */
    float u1 = 1.0f - u01(a);      /* (0,1] */
    float u2 = u01(b);
    float r = sqrtf(-2.0f * logf(u1));
    float t = 6.283185307179586f * u2;
    *n0 = r * cosf(t);
    *n1 = r * sinf(t);
}

/* ------------------------------------------------------------------------- */
/* Defaults: TY + URDF + V5 constants. */
int vine_config_default(VineConfig* c) {
    if (!c) return fail(VINE_ERR_INVALID_ARG, "cfg is NULL");
    memset(c, 0, sizeof *c);
    c->abi_version = VINE_ABI_VERSION;
    c->num_envs = 4096;                 /* TY:8 */
    c->obs_type = VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO; /* TY:60 */
    c->control_freq_inv = 4;            /* TY:15 */
    c->substeps = 10;                   /* TY:104 */
    c->max_episode_length = 500;        /* TY:14 */
    c->action_delay = 1;                /* TY:100 */
    c->flags = VINE_FLAG_USE_SMOOTHED_FPAM | VINE_FLAG_RANDOMIZE_DOF_INIT | VINE_FLAG_RANDOMIZE_TARGETS |
               VINE_FLAG_USE_TARGET_REACHED_RESET | VINE_FLAG_SCALE_OBSERVATIONS | VINE_FLAG_VINE_RANDOMIZE |
               VINE_FLAG_STALE_BODY_STATE_AFTER_RESET | VINE_FLAG_IMPLICIT_JOINT_DAMPING;
    c->seed = 42;                       /* cfg/config.yaml:28 */
    c->dt = 0.00833f; c->gravity = 9.81f; c->clip_observations = 5.0f; c->clip_actions = 1.0f;
    c->fpam_min = -0.1f; c->fpam_max = 3.0f; c->rail_velocity_scale = 1.0f;
    c->damping = 2e-2f; c->stiffness = 0.0f;
    c->rail_soft_limit = 0.3f; c->rail_p_gain = 10.0f; c->rail_d_gain = 0.0f; c->rail_acceleration = 8.0f;
    c->smoothing_alpha_inflate = 0.81f; c->smoothing_alpha_deflate = 0.86f;
    c->random_init_cart_min_y = (float)(-0.1 * 0.3); c->random_init_cart_max_y = 0.3f; /* TY:64-65 */
    c->success_dist = 0.08f;
    c->min_target_depth = -0.05f; c->max_target_depth = 0.2f;
    c->min_target_y = -0.48f; c->max_target_y = -0.4f; c->min_target_z = 0.58f; c->max_target_z = 0.67f;
    static const float w[VINE_NUM_REWARDS] = {0, 0, 1.0f, 0, 0.1f, 0, 0, 0, 0, 1.0f, 0, 0, 0.10f}; /* TY:77-89 */
    memcpy(c->reward_weights, w, sizeof w);
    c->dyn_scale_min = 0.999f; c->dyn_scale_max = 1.001f; c->obs_noise_std = 0; c->action_noise_std = 0;
    c->cart_mass = 0.4f;
    for (int i = 0; i < NL; ++i) { c->link_mass[i] = 0.005f; c->link_inertia[i] = 0.00000689246f; }
    c->link_mass[4] = 0.1f; c->link_inertia[4] = 0.000101559f;
    c->link_length = 0.0885f; c->link_com = 0.04425f;
    c->joint1_z = (float)(1.0 - 0.025 - 0.01); c->phi0 = 3.1415f;
    c->link_angular_damping = 0.0f;
    static const float K[NL] = {0.8385f, 1.5400f, 1.5109f, 1.2887f, 0.4347f};
    static const float C[NL] = {0.0178f, 0.0304f, 0.0528f, 0.0367f, 0.0223f};
    static const float b[NL] = {0.0007f, 0.0062f, 0.0402f, 0.0160f, 0.0133f};
    static const float B[NL] = {0.0247f, 0.0616f, 0.0779f, 0.0498f, 0.0268f};
    memcpy(c->fpam_K, K, sizeof K); memcpy(c->fpam_C, C, sizeof C);
    memcpy(c->fpam_b, b, sizeof b); memcpy(c->fpam_B, B, sizeof B);
    static const float s28[28] = {0.12f, 0.269f, 0.148f, 0.249f, 0.148f, 0.344f, 0.67f, 2.22f, 1.47f, 1.14f, 0.903f,
                                  0.716f, 0.0656f, 0.238f, 0.0656f, 0.732f, 2.0f, 0.732f, 0.02f, 0.0235f, 0.02f,
                                  0.732f, 2.0f, 0.732f, 0.845f, 0.86f, 0.0385f, 0.5f}; /* V5:246-255 */
    memcpy(c->obs_scaling, s28, sizeof s28);
    return VINE_OK;
}

/* obs_scaling for the two scalable observation types (V5:242-268). */
int vine_config_set_obs_type(VineConfig* c, int obs_type, int scale_observations) {
    if (!c) return fail(VINE_ERR_INVALID_ARG, "cfg is NULL");
    static const float s28[28] = {0.12f, 0.269f, 0.148f, 0.249f, 0.148f, 0.344f, 0.67f, 2.22f, 1.47f, 1.14f, 0.903f,
                                  0.716f, 0.0656f, 0.238f, 0.0656f, 0.732f, 2.0f, 0.732f, 0.02f, 0.0235f, 0.02f,
                                  0.732f, 2.0f, 0.732f, 0.845f, 0.86f, 0.0385f, 0.5f}; /* V5:246-255 */
    static const float s18[18] = {0.12f, 0.67f, 0.0656f, 0.238f, 0.0656f, 0.732f, 2.0f, 0.732f, 0.02f, 0.0235f, 0.02f,
                                  0.732f, 2.0f, 0.732f, 0.845f, 0.86f, 0.0385f, 0.5f}; /* V5:257-266 */
    if (obs_type < VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO || obs_type > VINE_OBS_POS_AND_PREV_POS)
        return fail(VINE_ERR_INVALID_ARG, "unknown observation type (V5:1380)");
    const int scalable = obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO || obs_type == VINE_OBS_TIP_AND_CART_AND_OBJ_INFO;
    if (scale_observations && !scalable)
        return fail(VINE_ERR_UNSUPPORTED, "observation scaling not implemented for this observation type (V5:267-268)");
    c->obs_type = obs_type;
    for (int i = 0; i < VINE_MAX_OBS; ++i) c->obs_scaling[i] = 1.0f;                     /* V5:241 */
    if (scale_observations) {
        c->flags |= VINE_FLAG_SCALE_OBSERVATIONS;
        if (obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) memcpy(c->obs_scaling, s28, sizeof s28);
        else memcpy(c->obs_scaling, s18, sizeof s18);
    } else c->flags &= ~(uint32_t)VINE_FLAG_SCALE_OBSERVATIONS;
    return VINE_OK;
}

int vine_num_obs(const VineConfig* c) {
    if (!c) return fail(VINE_ERR_INVALID_ARG, "cfg is NULL");
    if (c->obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) return 28; /* V5:164-170 */
    if (c->obs_type == VINE_OBS_TIP_AND_CART_AND_OBJ_INFO) return 18;   /* V5:157-162 */
    if (c->obs_type == VINE_OBS_POS_ONLY) return 14;                    /* V5:153-155 */
    if (c->obs_type >= VINE_OBS_POS_AND_VEL && c->obs_type <= VINE_OBS_POS_AND_PREV_POS) return 26; /* V5:164-167 */
    return fail(VINE_ERR_INVALID_ARG, "unknown observation type (V5:1380)");
}

/* ------------------------------------------------------------------------- */
/* Rigid-body model (row P1). Planar (y,z), coordinates
 * q = (y_c, q1..q5).  phi_k = phi0 + sum_{i<=k} q_i; link direction
 * d_k = (-sin phi_k, cos phi_k); joints spaced L, COM at l. */
typedef struct Model {
    real mc, m[NL], I[NL], L, l, z1, phi0, g, d, kq, cad;
    real b[NL];        /* b_i = m_i l + L sum_{k>i} m_k                     */
    real a[NL][NL];    /* a_ij = L b_max(i,j) (i!=j); a_ii = m_i l^2 + L^2 sum_{k>i} m_k + I_i */
    real mtot;
    int implicit_damping;
    /* probe switches of tests/test_oracle_physics.py (unverifiable Isaac Gym / PhysX defaults); all off
     * by default and never part of the product: joint armature (added to the joint-space inertia), clamp of the
     * links' world angular velocity (AssetOptions.max_angular_velocity, 64 rad/s), clamp of the joint velocities
     * (PhysX maxJointVelocity, 100 rad/s), joint efforts applied during the first substep of a simulate only */
    real armature, vmax_link, vmax_joint;
    int effort_first_substep_only;
} Model;

static void model_init(Model* M, const VineConfig* c) {
    M->mc = c->cart_mass; M->L = c->link_length; M->l = c->link_com; M->z1 = c->joint1_z;
    M->phi0 = c->phi0; M->g = c->gravity; M->d = c->damping; M->kq = c->stiffness;
    M->cad = c->link_angular_damping;
    M->implicit_damping = (c->flags & VINE_FLAG_IMPLICIT_JOINT_DAMPING) != 0;
    M->armature = 0; M->vmax_link = 0; M->vmax_joint = 0; M->effort_first_substep_only = 0;
    M->mtot = M->mc;
    for (int i = 0; i < NL; ++i) { M->m[i] = c->link_mass[i]; M->I[i] = c->link_inertia[i]; M->mtot += M->m[i]; }
    for (int i = 0; i < NL; ++i) {
        real distal = 0;
        for (int k = i + 1; k < NL; ++k) distal += M->m[k];
        M->b[i] = M->m[i] * M->l + M->L * distal;
        M->a[i][i] = M->m[i] * M->l * M->l + M->L * M->L * distal + M->I[i];
    }
    for (int i = 0; i < NL; ++i)
        for (int j = 0; j < NL; ++j)
            if (i != j) M->a[i][j] = M->L * M->b[i > j ? i : j];
}

/* Dense SPD solve (Cholesky), n <= 6. A is overwritten. */
static int chol_solve(int n, real A[ND][ND], real* x) {
    for (int j = 0; j < n; ++j) {
        real s = A[j][j];
        for (int k = 0; k < j; ++k) s -= A[j][k] * A[j][k];
        if (!(s > 0)) return -1;
        real Ljj = (real)sqrt((double)s);
        FLOPS(2 * j + 1);
        A[j][j] = Ljj;
        for (int i = j + 1; i < n; ++i) {
            real t = A[i][j];
            for (int k = 0; k < j; ++k) t -= A[i][k] * A[j][k];
            A[i][j] = t / Ljj;
            FLOPS(2 * j + 1);
        }
    }
    for (int i = 0; i < n; ++i) {
        real t = x[i];
        for (int k = 0; k < i; ++k) t -= A[i][k] * x[k];
        x[i] = t / A[i][i];
        FLOPS(2 * i + 1);
    }
    for (int i = n - 1; i >= 0; --i) {
        real t = x[i];
        for (int k = i + 1; k < n; ++k) t -= A[k][i] * x[k];
        x[i] = t / A[i][i];
        FLOPS(2 * (n - 1 - i) + 1);
    }
    return 0;
}

/* Generalised force in relative coordinates from actuation + passive joint terms.
 * eff[6] = (rail_force, tau_1..5) as set by set_dof_actuation_force_tensor (V5:1101-1106).
 * Passive: DOF damping d on all six DOFs (V5:504), spring `stiffness` on revolutes (V5:511),
 * optional per-link angular damping -cad*I_k*omega_k. */
static void passive_forces(const Model* M, const real* cj, const real* q, const real* qd, const real* eff, real* Q) {
    for (int i = 0; i < ND; ++i) Q[i] = eff[i] - cj[i] * qd[i];
    for (int i = 1; i < ND; ++i) Q[i] -= M->kq * q[i];
    if (M->cad != 0) { /* torque -cad*I_k*w_k on link k maps to every proximal revolute joint */
        real w = 0, wk[NL];
        for (int k = 0; k < NL; ++k) { w += qd[k + 1]; wk[k] = w; }
        for (int j = 0; j < NL; ++j)
            for (int k = j; k < NL; ++k) Q[j + 1] -= M->cad * M->I[k] * wk[k];
    }
}
/* Matrix that implicit integration of the passive velocity-dependent terms adds to the mass
 * matrix, in relative coordinates: h * d(-Q)/d(qd). */
static void implicit_matrix(const Model* M, const real* cj, real h, real D[ND][ND]) {
    memset(D, 0, sizeof(real) * ND * ND);
    for (int i = 1; i < ND; ++i) D[i][i] = M->armature;
    if (!M->implicit_damping) return;
    for (int i = 0; i < ND; ++i) D[i][i] += h * cj[i];
    if (M->cad != 0)
        for (int i = 0; i < NL; ++i)
            for (int j = 0; j < NL; ++j) {
                real s = 0;
                for (int k = (i > j ? i : j); k < NL; ++k) s += M->I[k];
                D[i + 1][j + 1] += h * M->cad * s;
            }
}

/* --- Formulation A: Jacobian-based CRBA in relative coordinates ---------- */
static int fd_crba(const Model* M, const real* cj, const real* q, const real* qd, const real* eff, real h, real* qdd) {
    real phi[NL], w[NL], dy[NL], dz[NL], py[NL + 1], pz[NL + 1], cy[NL], cz[NL];
    real ang = M->phi0, om = 0;
    py[0] = q[0]; pz[0] = M->z1;
    for (int k = 0; k < NL; ++k) {
        ang += q[k + 1]; om += qd[k + 1];
        phi[k] = ang; w[k] = om;
        dy[k] = -(real)sin((double)ang); dz[k] = (real)cos((double)ang);
        cy[k] = py[k] + M->l * dy[k]; cz[k] = pz[k] + M->l * dz[k];
        py[k + 1] = py[k] + M->L * dy[k]; pz[k + 1] = pz[k] + M->L * dz[k];
    }
    (void)phi;
    real A[ND][ND]; memset(A, 0, sizeof A);
    real bias[ND]; memset(bias, 0, sizeof bias);
    A[0][0] = M->mc;
    /* COM acceleration with qdd = 0: centripetal terms accumulated along the chain */
    real ay = 0, az = 0;
    for (int k = 0; k < NL; ++k) {
        real Jy[ND], Jz[ND], Jw[ND];
        Jy[0] = 1; Jz[0] = 0; Jw[0] = 0;
        for (int j = 0; j < NL; ++j) {
            if (j <= k) { Jy[j + 1] = -(cz[k] - pz[j]); Jz[j + 1] = (cy[k] - py[j]); Jw[j + 1] = 1; }
            else { Jy[j + 1] = 0; Jz[j + 1] = 0; Jw[j + 1] = 0; }
        }
        for (int i = 0; i < ND; ++i)
            for (int j = 0; j < ND; ++j)
                A[i][j] += M->m[k] * (Jy[i] * Jy[j] + Jz[i] * Jz[j]) + M->I[k] * Jw[i] * Jw[j];
        real acy = ay - w[k] * w[k] * M->l * dy[k];
        real acz = az - w[k] * w[k] * M->l * dz[k];
        for (int i = 0; i < ND; ++i) bias[i] += M->m[k] * (Jy[i] * acy + Jz[i] * (acz + M->g));
        ay -= w[k] * w[k] * M->L * dy[k];
        az -= w[k] * w[k] * M->L * dz[k];
    }
    real Q[ND], D[ND][ND];
    passive_forces(M, cj, q, qd, eff, Q);
    implicit_matrix(M, cj, h, D);
    for (int i = 0; i < ND; ++i) {
        qdd[i] = Q[i] - bias[i];
        for (int j = 0; j < ND; ++j) A[i][j] += D[i][j];
    }
    return chol_solve(ND, A, qdd);
}

/* --- Formulation B: planar Featherstone ABA, all quantities in world axes at the world origin.
 * Motion vectors (w, vy, vz): vy,vz = velocity of the body-fixed point at the origin.
 * Force vectors (n, fy, fz): n = moment about the origin.  Only valid with explicit passive
 * terms or with a diagonal implicit matrix (cad == 0): joint-space diagonal terms enter the
 * ABA as joint "armature" d_i. */
typedef struct { real w, y, z; } V3;
static inline V3 v3(real w, real y, real z) { V3 r = {w, y, z}; return r; }
static inline real dot3(V3 a, V3 b) { return a.w * b.w + a.y * b.y + a.z * b.z; }
static int fd_aba(const Model* M, const real* cj, const real* q, const real* qd, const real* eff, real h, real* qdd) {
    if (M->implicit_damping && M->cad != 0) return -2;
    const int nb = ND; /* body 0 = cart, bodies 1..5 = links */
    real py[NL + 1], pz[NL + 1];
    real ang = M->phi0;
    py[0] = q[0]; pz[0] = M->z1;
    V3 S[ND], v[ND], c[ND], pA[ND], U[ND];
    real IA[ND][3][3], Dj[ND], u[ND];
    real Q[ND];
    passive_forces(M, cj, q, qd, eff, Q);
    /* pass 1: velocities, bias accelerations, rigid-body inertias and bias forces */
    S[0] = v3(0, 1, 0); v[0] = v3(0, qd[0], 0); c[0] = v3(0, 0, 0);
    for (int i = 0; i < nb; ++i) {
        real m, Ic, cy, cz;
        if (i == 0) { m = M->mc; Ic = 0; cy = q[0]; cz = M->z1; }
        else {
            int k = i - 1;
            ang += q[i];
            real dy = -(real)sin((double)ang), dz = (real)cos((double)ang);
            S[i] = v3(1, pz[k], -py[k]);          /* rotation about joint k: v_O = w * perp(-p) */
            V3 vj = v3(S[i].w * qd[i], S[i].y * qd[i], S[i].z * qd[i]);
            v[i] = v3(v[i - 1].w + vj.w, v[i - 1].y + vj.y, v[i - 1].z + vj.z);
            /* c = v_i x vJ (planar motion cross product): (0, w1*perp(v2) - w2*perp(v1)) */
            c[i] = v3(0, v[i].w * (-vj.z) - vj.w * (-v[i].z), v[i].w * (vj.y) - vj.w * (v[i].y));
            m = M->m[k]; Ic = M->I[k];
            cy = py[k] + M->l * dy; cz = pz[k] + M->l * dz;
            py[k + 1] = py[k] + M->L * dy; pz[k + 1] = pz[k] + M->L * dz;
        }
        real (*Ii)[3] = IA[i];
        Ii[0][0] = Ic + m * (cy * cy + cz * cz); Ii[0][1] = -m * cz; Ii[0][2] = m * cy;
        Ii[1][0] = -m * cz; Ii[1][1] = m; Ii[1][2] = 0;
        Ii[2][0] = m * cy;  Ii[2][1] = 0; Ii[2][2] = m;
        /* momentum hmom = I v ; bias force p = v x* hmom - gravity wrench */
        V3 hm = v3(Ii[0][0] * v[i].w + Ii[0][1] * v[i].y + Ii[0][2] * v[i].z,
                   Ii[1][0] * v[i].w + Ii[1][1] * v[i].y + Ii[1][2] * v[i].z,
                   Ii[2][0] * v[i].w + Ii[2][1] * v[i].y + Ii[2][2] * v[i].z);
        /* v x* f = (vy*fz - vz*fy, w*perp(f)) with perp(y,z) = (-z, y) */
        pA[i] = v3(v[i].y * hm.z - v[i].z * hm.y, v[i].w * (-hm.z), v[i].w * (hm.y));
        /* gravity: force (0, 0, -m g) at the COM -> wrench about origin: n = cy*fz - cz*fy */
        pA[i].w -= cy * (-m * M->g);
        pA[i].z -= (-m * M->g);
    }
    /* pass 2: articulated inertias, tip to base */
    for (int i = nb - 1; i >= 0; --i) {
        real (*Ii)[3] = IA[i];
        U[i] = v3(Ii[0][0] * S[i].w + Ii[0][1] * S[i].y + Ii[0][2] * S[i].z,
                  Ii[1][0] * S[i].w + Ii[1][1] * S[i].y + Ii[1][2] * S[i].z,
                  Ii[2][0] * S[i].w + Ii[2][1] * S[i].y + Ii[2][2] * S[i].z);
        Dj[i] = dot3(S[i], U[i]) + (M->implicit_damping ? h * cj[i] : 0) + (i > 0 ? M->armature : 0);
        u[i] = Q[i] - dot3(S[i], pA[i]);
        if (i > 0) {
            real Ua[3] = {U[i].w, U[i].y, U[i].z};
            real Ia[3][3];
            for (int r = 0; r < 3; ++r)
                for (int s = 0; s < 3; ++s) Ia[r][s] = Ii[r][s] - Ua[r] * Ua[s] / Dj[i];
            real cc[3] = {c[i].w, c[i].y, c[i].z};
            real pa[3] = {pA[i].w, pA[i].y, pA[i].z};
            for (int r = 0; r < 3; ++r) {
                real t = pa[r] + Ua[r] * u[i] / Dj[i];
                for (int s = 0; s < 3; ++s) t += Ia[r][s] * cc[s];
                pa[r] = t;
            }
            for (int r = 0; r < 3; ++r)
                for (int s = 0; s < 3; ++s) IA[i - 1][r][s] += Ia[r][s];
            pA[i - 1].w += pa[0]; pA[i - 1].y += pa[1]; pA[i - 1].z += pa[2];
        }
    }
    /* pass 3: accelerations, base to tip (fixed base: a_parent(0) = 0; gravity is in pA) */
    V3 a = v3(0, 0, 0);
    for (int i = 0; i < nb; ++i) {
        V3 ap = v3(a.w + c[i].w, a.y + c[i].y, a.z + c[i].z);
        qdd[i] = (u[i] - dot3(U[i], ap)) / Dj[i];
        a = v3(ap.w + S[i].w * qdd[i], ap.y + S[i].y * qdd[i], ap.z + S[i].z * qdd[i]);
    }
    return 0;
}

/* --- Formulation C: closed-form Lagrangian in absolute angles (the layout the HIP kernel uses).
 * Coordinates x = (y, th_1..th_5), th_k = sum_{i<=k} q_i (phi_k = phi0 + th_k).
 *   row 0:  mtot*ydd - sum_i b_i cos(phi_i) thdd_i = F - sum_i b_i sin(phi_i) w_i^2
 *   row i: -b_i cos(phi_i) ydd + sum_j a_ij cos(th_i-th_j) thdd_j
 *            = Qa_i - sum_j a_ij sin(th_i-th_j) w_j^2 + g b_i sin(phi_i)
 * with Qa_i = T_i - T_{i+1} (relative joint torques T, T_6 = 0) - cad*I_i*w_i.
 * Returns accelerations in the ABSOLUTE coordinates (ydd, thdd_1..5). */
static int fd_abs(const Model* M, const real* cj, real ycart, const real* th, real vy, const real* w,
                  const real* eff, real h, real* acc) {
    (void)ycart;
    real s[NL], c[NL], sp[NL], cp[NL];
    real s0 = (real)sin((double)M->phi0), c0 = (real)cos((double)M->phi0);
    for (int i = 0; i < NL; ++i) {
        s[i] = (real)sin((double)th[i]); c[i] = (real)cos((double)th[i]);
        sp[i] = s0 * c[i] + c0 * s[i];   /* sin(phi_i) */
        cp[i] = c0 * c[i] - s0 * s[i];   /* cos(phi_i) */
        FLOPS(8);
    }
    real T[NL + 1];
    T[0] = eff[1] - cj[1] * w[0] - M->kq * th[0];
    for (int i = 1; i < NL; ++i) T[i] = eff[i + 1] - cj[i + 1] * (w[i] - w[i - 1]) - M->kq * (th[i] - th[i - 1]);
    T[NL] = 0;
    FLOPS(4 + 6 * (NL - 1));
    real A[ND][ND], r[ND];
    A[0][0] = M->mtot;
    r[0] = eff[0] - cj[0] * vy;
    FLOPS(2);
    for (int i = 0; i < NL; ++i) {
        A[0][i + 1] = A[i + 1][0] = -M->b[i] * cp[i];
        r[0] -= M->b[i] * sp[i] * w[i] * w[i];
        real ri = T[i] - T[i + 1] - M->cad * M->I[i] * w[i] + M->g * M->b[i] * sp[i];
        for (int j = 0; j < NL; ++j) {
            real cd = c[i] * c[j] + s[i] * s[j];
            real sd = s[i] * c[j] - c[i] * s[j];
            A[i + 1][j + 1] = M->a[i][j] * cd;
            ri -= M->a[i][j] * sd * w[j] * w[j];
            FLOPS(11);
        }
        r[i + 1] = ri;
        FLOPS(1 + 4 + 8);
    }
    if (M->implicit_damping) { /* h * T^T diag(cj) T with qd = T x_dot: tridiagonal */
        A[0][0] += h * cj[0];
        for (int i = 0; i < NL; ++i) {
            real cn = (i < NL - 1) ? cj[i + 2] : 0;
            A[i + 1][i + 1] += h * (cj[i + 1] + cn) + h * M->cad * M->I[i];
            if (i < NL - 1) { A[i + 1][i + 2] -= h * cn; A[i + 2][i + 1] -= h * cn; }
            FLOPS(10);
        }
        FLOPS(2);
    }
    if (M->armature != 0)      /* T^T diag(armature) T, same tridiagonal pattern */
        for (int i = 0; i < NL; ++i) {
            A[i + 1][i + 1] += M->armature * (i < NL - 1 ? 2 : 1);
            if (i < NL - 1) { A[i + 1][i + 2] -= M->armature; A[i + 2][i + 1] -= M->armature; }
        }
    for (int i = 0; i < ND; ++i) acc[i] = r[i];
    return chol_solve(ND, A, acc);
}

/* Forward dynamics in relative coordinates through the chosen formulation. */
enum { FORM_CRBA = 0, FORM_ABA = 1, FORM_ABS = 2 };
static int forward_dynamics(const Model* M, int form, const real* cj, const real* q, const real* qd, const real* eff,
                            real h, real* qdd) {
    if (form == FORM_CRBA) return fd_crba(M, cj, q, qd, eff, h, qdd);
    if (form == FORM_ABA) return fd_aba(M, cj, q, qd, eff, h, qdd);
    real th[NL], w[NL], acc[ND];
    real a = 0, b = 0;
    for (int k = 0; k < NL; ++k) { a += q[k + 1]; b += qd[k + 1]; th[k] = a; w[k] = b; }
    int rc = fd_abs(M, cj, q[0], th, qd[0], w, eff, h, acc);
    qdd[0] = acc[0]; qdd[1] = acc[1];
    for (int k = 1; k < NL; ++k) qdd[k + 1] = acc[k + 1] - acc[k];
    FLOPS(2 * NL + NL - 1);
    return rc;
}

/* exported for tests: forward dynamics of one state through one formulation */
int vine_oracle_forward_dynamics(const VineConfig* cfg, int form, const double* q, const double* qd,
                                 const double* eff, const double* cjoint, double h, double* qdd) {
    Model M; model_init(&M, cfg);
    real rq[ND], rqd[ND], re[ND], ra[ND], cj[ND];
    for (int i = 0; i < ND; ++i) { rq[i] = (real)q[i]; rqd[i] = (real)qd[i]; re[i] = (real)eff[i]; cj[i] = cjoint ? (real)cjoint[i] : M.d; }
    int rc = forward_dynamics(&M, form, cj, rq, rqd, re, (real)h, ra);
    for (int i = 0; i < ND; ++i) qdd[i] = ra[i];
    return rc;
}

/* Forward kinematics of the massless `tip` body (URDF:264, 322-326) and total energy. */
static void tip_kinematics(const Model* M, const real* q, const real* qd, real* tip /*y,z,vy,vz*/) {
    real ang = M->phi0, om = 0, y = q[0], z = M->z1, vy = qd[0], vz = 0;
    for (int k = 0; k < NL; ++k) {
        ang += q[k + 1]; om += qd[k + 1];
        real s = (real)sin((double)ang), c = (real)cos((double)ang);
        y += M->L * (-s); z += M->L * c;
        vy += M->L * om * (-c); vz += M->L * om * (-s);
        FLOPS(2 + 2 + 4 + 6);
    }
    tip[0] = y; tip[1] = z; tip[2] = vy; tip[3] = vz;
}
void vine_oracle_tip(const VineConfig* cfg, const double* q, const double* qd, double* tip) {
    Model M; model_init(&M, cfg);
    real rq[ND], rqd[ND], t[4];
    for (int i = 0; i < ND; ++i) { rq[i] = (real)q[i]; rqd[i] = (real)qd[i]; }
    tip_kinematics(&M, rq, rqd, t);
    for (int i = 0; i < 4; ++i) tip[i] = t[i];
}
double vine_oracle_energy(const VineConfig* cfg, const double* q, const double* qd) {
    Model M; model_init(&M, cfg);
    double ang = M.phi0, om = 0, py = q[0], pz = M.z1, pvy = qd[0], pvz = 0;
    double E = 0.5 * M.mc * qd[0] * qd[0];
    for (int k = 0; k < NL; ++k) {
        ang += q[k + 1]; om += qd[k + 1];
        double s = sin(ang), c = cos(ang);
        double cvy = pvy + M.l * om * (-c), cvz = pvz + M.l * om * (-s);
        double cz = pz + M.l * c;
        E += 0.5 * M.m[k] * (cvy * cvy + cvz * cvz) + 0.5 * M.I[k] * om * om + M.m[k] * M.g * cz;
        py += M.L * (-s); pz += M.L * c; pvy += M.L * om * (-c); pvz += M.L * om * (-s);
    }
    (void)py;
    return E;
}

/* One substep: semi-implicit Euler (qd += h*qdd; q += h*qd). */
static int substep(const Model* M, int form, const real* cj, real* q, real* qd, const real* eff, real h) {
    real qdd[ND];
    int rc = forward_dynamics(M, form, cj, q, qd, eff, h, qdd);
    if (rc) return rc;
    for (int i = 0; i < ND; ++i) qd[i] += h * qdd[i];
    FLOPS(4 * ND);
    if (M->vmax_link > 0) {          /* clamp of the links' world angular velocities */
        real w = 0, prev = 0;
        for (int k = 0; k < NL; ++k) {
            w += qd[k + 1];                       /* world rate of link k before the clamp */
            real wc = w > M->vmax_link ? M->vmax_link : (w < -M->vmax_link ? -M->vmax_link : w);
            qd[k + 1] = wc - prev; prev = wc;
        }
    }
    if (M->vmax_joint > 0)
        for (int i = 1; i < ND; ++i) qd[i] = qd[i] > M->vmax_joint ? M->vmax_joint : (qd[i] < -M->vmax_joint ? -M->vmax_joint : qd[i]);
    for (int i = 0; i < ND; ++i) q[i] += h * qd[i];
    return 0;
}
/* exported for tests: n substeps with constant efforts */
int vine_oracle_simulate(const VineConfig* cfg, int form, double* q, double* qd, const double* eff,
                         const double* cjoint, double h, int n) {
    Model M; model_init(&M, cfg);
    real rq[ND], rqd[ND], re[ND], cj[ND];
    for (int i = 0; i < ND; ++i) { rq[i] = (real)q[i]; rqd[i] = (real)qd[i]; re[i] = (real)eff[i]; cj[i] = cjoint ? (real)cjoint[i] : M.d; }
    for (int s = 0; s < n; ++s) { int rc = substep(&M, form, cj, rq, rqd, re, (real)h); if (rc) return rc; }
    for (int i = 0; i < ND; ++i) { q[i] = rq[i]; qd[i] = rqd[i]; }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Planar shelf contact (config 5).  The contact solver is the build's own choice
 * (PhysX is unavailable: unpinned): frictionless penalty contacts, explicit per substep.
 *   (a) six points of every link rectangle (4 corners + 2 mid-edge) against the two boards of link `shelf`
 *       (custom_shelf.urdf:82-93), axis-aligned boxes in the world;
 *   (b) the two front corners of the 2 mm front-edge strip on `shelf_link` (custom_shelf.urdf:139-152) against
 *       every link rectangle; only (b) feeds the reported contact force (V5:329-336; VT:349-351).
 * Link rectangle in link-local (y, z): y in [-0.0381, +0.0719] (main cylinder r = 0.0381 + FPAM cylinder at
 * y = 0.055, r = 0.0169; URDF:95-115), z in [0, L] (link_0: length 0.1 centred at 0.04425 -> [-0.00575, 0.09425]).
 * Local z maps to d = (-sin phi, cos phi), local y to l = (cos phi, sin phi) in world (y, z). */
#define CONTACT_K ((real)2000.0)  /* N/m   penalty stiffness */
#define CONTACT_C ((real)2.0)     /* N s/m penalty damping   */
#define LINK_Y0 ((real)-0.0381)
#define LINK_Y1 ((real)0.0719)
static const real BOARD[2][4] = {
    /* cy,      cz,   hy,     hz   in the shelf frame */
    {-0.001f, 0.0f, 0.1995f, 0.005f},
    {0.0f, 0.2f, 0.2f, 0.005f},
};

/* Adds the contact generalised forces (relative coordinates) to Qc; returns |F| on the strip. */
static real shelf_contact(const Model* M, const real* q, const real* qd, real shelf_y, real shelf_z, real* Qc) {
    real strip_fy = 0, strip_fz = 0;
    real ang = M->phi0, om = 0;
    real py[NL + 1], pz[NL + 1], pvy = qd[0], pvz = 0;
    py[0] = q[0]; pz[0] = M->z1;
    for (int k = 0; k < NL; ++k) {
        ang += q[k + 1]; om += qd[k + 1];
        real s = (real)sin((double)ang), c = (real)cos((double)ang);
        real dy = -s, dz = c, ly = c, lz = s;
        real z0 = (k == 0) ? (real)-0.00575 : 0, z1 = (k == 0) ? (real)0.09425 : M->L;
        real fy_tot = 0, fz_tot = 0;   /* applied to this link, with moments about every proximal joint */
        real mom[NL];
        for (int j = 0; j <= k; ++j) mom[j] = 0;
        /* (a) link points vs boards */
        for (int e = 0; e < 2; ++e)
            for (int t = 0; t < 3; ++t) {
                real yl = e ? LINK_Y1 : LINK_Y0;
                real zl = (t == 0) ? z0 : (t == 1 ? (real)0.5 * (z0 + z1) : z1);
                real ry = zl * dy + yl * ly, rz = zl * dz + yl * lz;
                real wy = py[k] + ry, wz = pz[k] + rz;
                real vy = pvy - om * rz, vz = pvz + om * ry;
                for (int bx = 0; bx < 2; ++bx) {
                    real ddy = wy - (shelf_y + BOARD[bx][0]), ddz = wz - (shelf_z + BOARD[bx][1]);
                    real ey = BOARD[bx][2] - (real)fabs((double)ddy), ez = BOARD[bx][3] - (real)fabs((double)ddz);
                    if (ey <= 0 || ez <= 0) continue;
                    real fy = 0, fz = 0;
                    if (ey < ez) {
                        real sg = (ddy > 0) ? (real)1 : (real)-1;
                        real f = CONTACT_K * ey - CONTACT_C * sg * vy;
                        fy = sg * (f > 0 ? f : 0);
                    } else {
                        real sg = (ddz > 0) ? (real)1 : (real)-1;
                        real f = CONTACT_K * ez - CONTACT_C * sg * vz;
                        fz = sg * (f > 0 ? f : 0);
                    }
                    fy_tot += fy; fz_tot += fz;
                    for (int j = 0; j <= k; ++j) mom[j] += (wy - py[j]) * fz - (wz - pz[j]) * fy;
                }
            }
        /* (b) strip front corners vs this link's rectangle */
        for (int e = 0; e < 2; ++e) {
            real wy = shelf_y + (real)0.2, wz = shelf_z + (e ? (real)0.005 : (real)-0.005);
            real ry = wy - py[k], rz = wz - pz[k];
            real zl = ry * dy + rz * dz, yl = ry * ly + rz * lz;
            if (!(zl > z0 && zl < z1 && yl > LINK_Y0 && yl < LINK_Y1)) continue;
            /* exit through the nearest face: outward normal n, depth dep */
            real dep = zl - z0, ny = -dy, nz = -dz;
            if (z1 - zl < dep) { dep = z1 - zl; ny = dy; nz = dz; }
            if (yl - LINK_Y0 < dep) { dep = yl - LINK_Y0; ny = -ly; nz = -lz; }
            if (LINK_Y1 - yl < dep) { dep = LINK_Y1 - yl; ny = ly; nz = lz; }
            real vy = pvy - om * rz, vz = pvz + om * ry;          /* link material velocity at the point */
            real f = CONTACT_K * dep + CONTACT_C * (vy * ny + vz * nz);
            if (f < 0) f = 0;
            strip_fy += f * ny; strip_fz += f * nz;               /* on the strip */
            real fy = -f * ny, fz = -f * nz;                      /* on the link  */
            fy_tot += fy; fz_tot += fz;
            for (int j = 0; j <= k; ++j) mom[j] += (wy - py[j]) * fz - (wz - pz[j]) * fy;
        }
        Qc[0] += fy_tot;
        for (int j = 0; j <= k; ++j) Qc[j + 1] += mom[j];
        (void)fz_tot;
        py[k + 1] = py[k] + M->L * dy; pz[k + 1] = pz[k] + M->L * dz;
        pvy += M->L * om * (-c); pvz += M->L * om * (-s);
    }
    return (real)sqrt((double)(strip_fy * strip_fy + strip_fz * strip_fz));
}
/* exported for tests */
double vine_oracle_shelf_contact(const VineConfig* cfg, const double* q, const double* qd, double shelf_y,
                                 double shelf_z, double* Qc) {
    Model M; model_init(&M, cfg);
    real rq[ND], rqd[ND], Q[ND];
    for (int i = 0; i < ND; ++i) { rq[i] = (real)q[i]; rqd[i] = (real)qd[i]; Q[i] = 0; }
    real f = shelf_contact(&M, rq, rqd, (real)shelf_y, (real)shelf_z, Q);
    for (int i = 0; i < ND; ++i) Qc[i] = Q[i];
    return f;
}

/* Planar pipe contact (CREATE_PIPE; V5:841-885, assets/urdf/pipe: mesh cylinder-13_8cm-ID.STL scaled by
 * 0.001 * PIPE_ADDITIONAL_SCALING = 0.00105 -> tube of length 0.34125, inner radius 0.07245, wall 0.00525, whose
 * local origin is a corner of its bounding box and whose axis runs along local z).  The root is rotated about x by
 * theta = theta' + 90 deg; in the vine's plane (the tube axis sits at world x = 0.0042) the obstacle is two wall
 * rectangles in the pipe frame: y_l in [0, 0.00525] and [0.15015, 0.1554], z_l in [0, 0.34125].
 * Same penalty model as the shelf: (a) 6 points per link vs the two walls, (b) the 4 rim corners of the entrance
 * (z_l = 0) and of the far end vs every link rectangle.  No contact force is reported for the pipe (V5:1246). */
#define PIPE_LEN ((real)0.34125)
#define PIPE_WALL ((real)0.00525)
#define PIPE_OUTER ((real)0.1554)
static void pipe_contact(const Model* M, const real* q, const real* qd, real pipe_y, real pipe_z, real theta_prime,
                         real* Qc) {
    const real th = theta_prime + (real)1.5707963267948966;
    const real ct = (real)cos((double)th), st = (real)sin((double)th);
    /* pipe-frame axes in world (y,z): e_y = (ct, st), e_z = (-st, ct) */
    const real wall_lo[2] = {0, PIPE_OUTER - PIPE_WALL};
    real ang = M->phi0, om = 0;
    real py[NL + 1], pz[NL + 1], pvy = qd[0], pvz = 0;
    py[0] = q[0]; pz[0] = M->z1;
    for (int k = 0; k < NL; ++k) {
        ang += q[k + 1]; om += qd[k + 1];
        real s = (real)sin((double)ang), c = (real)cos((double)ang);
        real dy = -s, dz = c, ly = c, lz = s;
        real z0 = (k == 0) ? (real)-0.00575 : 0, z1 = (k == 0) ? (real)0.09425 : M->L;
        real fy_tot = 0, mom[NL];
        for (int j = 0; j <= k; ++j) mom[j] = 0;
        for (int e = 0; e < 2; ++e)
            for (int t = 0; t < 3; ++t) {
                real yl = e ? LINK_Y1 : LINK_Y0;
                real zl = (t == 0) ? z0 : (t == 1 ? (real)0.5 * (z0 + z1) : z1);
                real ry = zl * dy + yl * ly, rz = zl * dz + yl * lz;
                real wy = py[k] + ry, wz = pz[k] + rz;
                real vy = pvy - om * rz, vz = pvz + om * ry;
                /* into the pipe frame */
                real gy = wy - pipe_y, gz = wz - pipe_z;
                real pyl = gy * ct + gz * st, pzl = -gy * st + gz * ct;
                real vyl = vy * ct + vz * st, vzl = -vy * st + vz * ct;
                for (int w = 0; w < 2; ++w) {
                    real cy = wall_lo[w] + (real)0.5 * PIPE_WALL, cz = (real)0.5 * PIPE_LEN;
                    real ddy = pyl - cy, ddz = pzl - cz;
                    real ey = (real)0.5 * PIPE_WALL - (real)fabs((double)ddy), ez = (real)0.5 * PIPE_LEN - (real)fabs((double)ddz);
                    if (ey <= 0 || ez <= 0) continue;
                    real fyl = 0, fzl = 0;
                    if (ey < ez) {
                        real sg = (ddy > 0) ? (real)1 : (real)-1;
                        real f = CONTACT_K * ey - CONTACT_C * sg * vyl;
                        fyl = sg * (f > 0 ? f : 0);
                    } else {
                        real sg = (ddz > 0) ? (real)1 : (real)-1;
                        real f = CONTACT_K * ez - CONTACT_C * sg * vzl;
                        fzl = sg * (f > 0 ? f : 0);
                    }
                    real fy = fyl * ct - fzl * st, fz = fyl * st + fzl * ct;   /* back to world */
                    fy_tot += fy;
                    for (int j = 0; j <= k; ++j) mom[j] += (wy - py[j]) * fz - (wz - pz[j]) * fy;
                }
            }
        /* rim corners of both walls at both ends vs this link's rectangle */
        for (int w = 0; w < 2; ++w)
            for (int cidx = 0; cidx < 4; ++cidx) {
                real pyl = wall_lo[w] + ((cidx & 1) ? PIPE_WALL : 0), pzl = (cidx & 2) ? PIPE_LEN : 0;
                real wy = pipe_y + pyl * ct - pzl * st, wz = pipe_z + pyl * st + pzl * ct;
                real ry = wy - py[k], rz = wz - pz[k];
                real zl = ry * dy + rz * dz, yl = ry * ly + rz * lz;
                if (!(zl > z0 && zl < z1 && yl > LINK_Y0 && yl < LINK_Y1)) continue;
                real dep = zl - z0, ny = -dy, nz = -dz;
                if (z1 - zl < dep) { dep = z1 - zl; ny = dy; nz = dz; }
                if (yl - LINK_Y0 < dep) { dep = yl - LINK_Y0; ny = -ly; nz = -lz; }
                if (LINK_Y1 - yl < dep) { dep = LINK_Y1 - yl; ny = ly; nz = lz; }
                real vy = pvy - om * rz, vz = pvz + om * ry;
                real f = CONTACT_K * dep + CONTACT_C * (vy * ny + vz * nz);
                if (f < 0) f = 0;
                real fy = -f * ny, fz = -f * nz;
                fy_tot += fy;
                for (int j = 0; j <= k; ++j) mom[j] += (wy - py[j]) * fz - (wz - pz[j]) * fy;
            }
        Qc[0] += fy_tot;
        for (int j = 0; j <= k; ++j) Qc[j + 1] += mom[j];
        py[k + 1] = py[k] + M->L * dy; pz[k + 1] = pz[k] + M->L * dz;
        pvy += M->L * om * (-c); pvz += M->L * om * (-s);
    }
}
void vine_oracle_pipe_contact(const VineConfig* cfg, const double* q, const double* qd, double pipe_y, double pipe_z,
                              double theta_prime, double* Qc) {
    Model M; model_init(&M, cfg);
    real rq[ND], rqd[ND], Q[ND];
    for (int i = 0; i < ND; ++i) { rq[i] = (real)q[i]; rqd[i] = (real)qd[i]; Q[i] = 0; }
    pipe_contact(&M, rq, rqd, (real)pipe_y, (real)pipe_z, (real)theta_prime, Q);
    for (int i = 0; i < ND; ++i) Qc[i] = Q[i];
}

/* ------------------------------------------------------------------------- */
/* Glue, as pure per-env functions (float-typed like the reference tensors when real=float). */

/* rescale_to_u / rescale_to_u_rail_velocity, V5:1458-1463, 984-997 */
/* One ``gym.simulate`` (VT:356): ``substeps`` substeps with the efforts held and the obstacle contacts re-evaluated in
 * each.  Returns what the net-contact-force tensor reports for `shelf_link` after it (this build's choice: the mean
 * over the substeps of |F| on the front-edge strip; 0 without a shelf).  Shared by the env step and by
 * ``vine_oracle_simulate_obstacles`` (the ``simulate`` of the golden generator's fake tensor API). */
static real sim_step(const Model* M, int form, int substeps, const real* cj, real* q, real* qd, const real* eff, real hsub,
                     int shelf, int pipe, real shelf_y, real shelf_z, real pipe_y, real pipe_z, real pipe_tp) {
    real csum = 0;
    for (int s = 0; s < substeps; ++s) {
        real effc[ND];
        for (int i = 0; i < ND; ++i) effc[i] = (i > 0 && s > 0 && M->effort_first_substep_only) ? 0 : eff[i];
        if (shelf) csum += shelf_contact(M, q, qd, shelf_y, shelf_z, effc);
        if (pipe) pipe_contact(M, q, qd, pipe_y, pipe_z, pipe_tp, effc);
        FLOP_BUCKET(0); FLOP_CALL(0);
        substep(M, form, cj, q, qd, effc, hsub);
        FLOP_BUCKET(2);
    }
    return shelf ? csum / (real)substeps : 0;
}
/* exported for tests/golden/make_golden.py: one simulate with obstacles; obstacle = {shelf_y, shelf_z, pipe_y, pipe_z,
 * theta'}; returns the contact value described above. */
double vine_oracle_simulate_obstacles(const VineConfig* cfg, int form, double* q, double* qd, const double* eff,
                                      const double* cjoint, double h, int n, int shelf, int pipe, const double* obstacle) {
    Model M; model_init(&M, cfg);
    real rq[ND], rqd[ND], re[ND], cj[ND];
    for (int i = 0; i < ND; ++i) { rq[i] = (real)q[i]; rqd[i] = (real)qd[i]; re[i] = (real)eff[i]; cj[i] = cjoint ? (real)cjoint[i] : M.d; }
    real contact = sim_step(&M, form, n, cj, rq, rqd, re, (real)h, shelf, pipe, (real)obstacle[0], (real)obstacle[1],
                            (real)obstacle[2], (real)obstacle[3], (real)obstacle[4]);
    for (int i = 0; i < ND; ++i) { q[i] = rq[i]; qd[i] = rqd[i]; }
    return contact;
}

static void raw_actions_to_actions(const VineConfig* c, real a0, real a1, real* u_rail, real* u_fpam) {
    *u_rail = a0 * (real)c->rail_velocity_scale;
    *u_fpam = (a1 + (real)1.0) / (real)2.0 * (real)((double)c->fpam_max - (double)c->fpam_min) + (real)c->fpam_min;
}
/* u_fpam_to_smoothed_u_fpam, V5:999-1005 */
static real smooth_fpam(const VineConfig* c, real u, real sm) {
    real alpha = (u > sm) ? (real)c->smoothing_alpha_inflate : (real)c->smoothing_alpha_deflate;
    return alpha * sm + ((real)1 - alpha) * u;
}
/* compute_and_set_dof_actuation_force_tensor, V5:1028-1106.
 * scale[20] multiplies the diagonals of [K C diag(b) diag(B)] (V5:1053-1055), order
 * scale[0..4]=K, [5..9]=C, [10..14]=b, [15..19]=B; all ones when vine_randomize is off. */
static void actuation(const VineConfig* c, const real* q, const real* qd, real cart_vy, real u_rail, real u_used,
                      const real* scale, real* prev_cart_vel, real* prev_cart_vel_err, real* eff, real* cj) {
    /* eff = the efforts the reference hands to set_dof_actuation_force_tensor (V5:1101-1106).
     * cj (optional) = per-DOF damping the simulate stage integrates.  Unless FPAM_DAMPING_HELD is
     * set, the velocity term C_j*qd_j is NOT held over the sim step: it is removed from eff and
     * its coefficient joins the DOF damping (assumption P4: a held velocity
     * feedback with C_j > DAMPING is unstable in the light inter-link modes). */
    const int held = (c->flags & VINE_FLAG_FPAM_DAMPING_HELD) != 0 || cj == NULL;
    if (cj) cj[0] = (real)c->damping;
    for (int j = 0; j < NL; ++j) {
        real t = (real)c->fpam_K[j] * scale[j] * q[j + 1];
        real cv = (real)c->fpam_C[j] * scale[5 + j];
        if (held) t += cv * qd[j + 1];
        t += (real)c->fpam_b[j] * scale[10 + j];
        t += (real)c->fpam_B[j] * scale[15 + j] * u_used;
        eff[j + 1] = -t;
        if (cj) cj[j + 1] = (real)c->damping + (held ? 0 : cv);
        FLOPS(2 + 1 + (held ? 2 : 1) + 2 + 3 + 1);
    }
    FLOPS(1 + 1 + 2 + 2 + 4 + 1);
    real err = u_rail - cart_vy;                                   /* V5:1070 */
    real acc = (real)c->rail_acceleration;
    real fmax = acc / (real)2.0;                                   /* V5:1075 */
    real minmax = (err > 0) ? fmax : -fmax;                        /* V5:1076 */
    real accel = (cart_vy - *prev_cart_vel) / (real)c->dt;         /* V5:1079 (sim dt) */
    real accel_target = (err > 0) ? acc : -acc;                    /* V5:1081 */
    minmax += (real)0.30 * (accel_target - accel);                 /* V5:1083-1087 */
    real pid = (real)c->rail_p_gain * err + (real)c->rail_d_gain * (err - *prev_cart_vel_err); /* V5:1090 */
    eff[0] = ((real)fabs((double)err) > (real)0.1) ? minmax : pid; /* V5:1094 */
    *prev_cart_vel_err = err;                                      /* V5:1097 */
    *prev_cart_vel = cart_vy;                                      /* V5:1098 */
}
/* compute_observations, V5:1339-1385 (before noise and before the VT:374 clamp). Returns the column count. */
/* tip = (y, z, vy, vz) of the tip body; qd = simulator joint velocities (used by POS_AND_VEL only). */
static int observations(const VineConfig* c, const real* q, const real* qd, const real* prev_q, const real* tip,
                        real prev_tip_y, real prev_tip_z, real ty, real tz, real smoothed, real prev_u_rail,
                        real obj_depth, real obj_angle, real* o) {
    const real cdt = (real)c->dt * (real)c->control_freq_inv;        /* V5:228 */
    int k = 0;
    real fd_tip_y = (tip[0] - prev_tip_y) / cdt, fd_tip_z = (tip[1] - prev_tip_z) / cdt; /* V5:1348 */
    if (c->obs_type == VINE_OBS_POS_ONLY) {                          /* V5:1354-1356 */
        for (int i = 0; i < ND; ++i) o[k++] = q[i];
        o[k++] = 0; o[k++] = tip[0]; o[k++] = tip[1];
        o[k++] = 0; o[k++] = ty; o[k++] = tz;
        o[k++] = smoothed; o[k++] = prev_u_rail;
        return k;                                                    /* obs_scaling is ones (V5:241, 267-268) */
    }
    if (c->obs_type >= VINE_OBS_POS_AND_VEL && c->obs_type <= VINE_OBS_POS_AND_PREV_POS) { /* V5:1357-1368 */
        const int t = c->obs_type;
        for (int i = 0; i < ND; ++i) o[k++] = q[i];
        for (int i = 0; i < ND; ++i)
            o[k++] = (t == VINE_OBS_POS_AND_VEL) ? qd[i] : (t == VINE_OBS_POS_AND_FD_VEL) ? (q[i] - prev_q[i]) / cdt : prev_q[i];
        o[k++] = 0; o[k++] = tip[0]; o[k++] = tip[1];
        o[k++] = 0;
        o[k++] = (t == VINE_OBS_POS_AND_VEL) ? tip[2] : (t == VINE_OBS_POS_AND_FD_VEL) ? fd_tip_y : prev_tip_y;
        o[k++] = (t == VINE_OBS_POS_AND_VEL) ? tip[3] : (t == VINE_OBS_POS_AND_FD_VEL) ? fd_tip_z : prev_tip_z;
        o[k++] = 0; o[k++] = ty; o[k++] = tz;
        o[k++] = 0; o[k++] = 0; o[k++] = 0;                          /* target_velocities == 0, V5:916-918 */
        o[k++] = smoothed; o[k++] = prev_u_rail;
        return k;
    }
    if (c->obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) {
        for (int i = 0; i < ND; ++i) o[k++] = q[i];
        for (int i = 0; i < ND; ++i) o[k++] = (q[i] - prev_q[i]) / cdt;                  /* V5:1347 */
    } else {
        o[k++] = q[0];
        o[k++] = (q[0] - prev_q[0]) / cdt;
    }
    o[k++] = 0; o[k++] = tip[0]; o[k++] = tip[1];
    o[k++] = 0; o[k++] = fd_tip_y; o[k++] = fd_tip_z;
    o[k++] = 0; o[k++] = ty; o[k++] = tz;
    o[k++] = 0; o[k++] = 0; o[k++] = 0;                              /* target_velocities == 0, V5:916-918 */
    o[k++] = smoothed; o[k++] = prev_u_rail; o[k++] = obj_depth; o[k++] = obj_angle;
    for (int i = 0; i < k; ++i) o[i] = o[i] / (real)c->obs_scaling[i];                   /* V5:1385 */
    FLOPS(1 + 4 + (c->obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO ? 2 * ND : 2) + k);
    return k;
}
int vine_oracle_observations_ex(const VineConfig* c, const double* q, const double* qd, const double* prev_q,
                                const double* tip_yz, const double* tip_vel_yz, const double* prev_tip_yz,
                                const double* target_yz, double smoothed, double prev_u_rail, const double* obj_info,
                                double* obs) {
    real rq[ND], rv[ND], rp[ND], o[VINE_MAX_OBS];
    real t[4] = {(real)tip_yz[0], (real)tip_yz[1], tip_vel_yz ? (real)tip_vel_yz[0] : 0, tip_vel_yz ? (real)tip_vel_yz[1] : 0};
    for (int i = 0; i < ND; ++i) { rq[i] = (real)q[i]; rp[i] = (real)prev_q[i]; rv[i] = qd ? (real)qd[i] : 0; }
    int k = observations(c, rq, rv, rp, t, (real)prev_tip_yz[0], (real)prev_tip_yz[1], (real)target_yz[0],
                         (real)target_yz[1], (real)smoothed, (real)prev_u_rail, (real)obj_info[0], (real)obj_info[1], o);
    for (int i = 0; i < k; ++i) obs[i] = o[i];
    return k;
}
int vine_oracle_observations(const VineConfig* c, const double* q, const double* prev_q, const double* tip_yz,
                             const double* prev_tip_yz, const double* target_yz, double smoothed, double prev_u_rail,
                             const double* obj_info, double* obs) {
    real rq[ND], rp[ND], t[4] = {(real)tip_yz[0], (real)tip_yz[1], 0, 0}, o[VINE_MAX_OBS];
    for (int i = 0; i < ND; ++i) { rq[i] = (real)q[i]; rp[i] = (real)prev_q[i]; }
    int k = observations(c, rq, rq, rp, t, (real)prev_tip_yz[0], (real)prev_tip_yz[1], (real)target_yz[0], (real)target_yz[1],
                         (real)smoothed, (real)prev_u_rail, (real)obj_info[0], (real)obj_info[1], o);
    for (int i = 0; i < k; ++i) obs[i] = o[i];
    return k;
}

/* compute_reward_jit, V5:1470-1537.  rm[13] = unweighted terms. */
static real reward_terms(const VineConfig* c, real dist, int reached, real tip_vy, real tip_vz, real u_rail,
                         real u_fpam, real prev_u_rail, real smoothed, int limit_hit, int tip_limit_hit, real cart_y,
                         real contact, real* rm) {
    real vnorm = (real)sqrt((double)(tip_vy * tip_vy + tip_vz * tip_vz)); /* target velocity is zero, V5:916-918 */
    rm[0] = -dist;
    rm[1] = -1;
    rm[2] = reached ? (real)1000.0 : 0;
    rm[3] = -(reached ? vnorm : 0);
    rm[4] = vnorm;
    rm[5] = -(real)fabs((double)u_rail);
    rm[6] = -(real)fabs((double)u_fpam);
    rm[7] = -(real)fabs((double)(u_rail - prev_u_rail));
    rm[8] = -(real)fabs((double)(u_fpam - smoothed));
    rm[9] = limit_hit ? (real)-100.0 : 0;
    rm[10] = -(real)fabs((double)cart_y);
    rm[11] = tip_limit_hit ? (real)-100.0 : 0;
    rm[12] = -((contact > 0) ? contact : 0);
    real total = 0;
    for (int i = 0; i < VINE_NUM_REWARDS; ++i) total += rm[i] * (real)c->reward_weights[i];
    FLOPS(4 + 8 + 2 * VINE_NUM_REWARDS);
    return total;
}
/* compute_reset_jit, V5:1540-1558 */
static int64_t reset_logic(const VineConfig* c, int64_t reset_in, int64_t progress, int reached, int limit_hit,
                           int tip_limit_hit, int nonzero_contact) {
    int64_t r = (progress >= (int64_t)c->max_episode_length - 1) ? 1 : reset_in;
    if (reached && (c->flags & VINE_FLAG_USE_TARGET_REACHED_RESET)) r = 1;
    if (tip_limit_hit && (c->flags & VINE_FLAG_USE_TIP_LIMIT_HIT_RESET)) r = 1;
    if (limit_hit) r = 1;
    if (nonzero_contact && (c->flags & VINE_FLAG_USE_NONZERO_CONTACT_FORCE_RESET)) r = 1;
    return r;
}

/* exported pure-function entry points for the golden-vector tests (double in/out) */
void vine_oracle_actions(const VineConfig* c, double a0, double a1, double* u_rail, double* u_fpam) {
    real r, f; raw_actions_to_actions(c, (real)a0, (real)a1, &r, &f); *u_rail = r; *u_fpam = f;
}
double vine_oracle_smooth(const VineConfig* c, double u, double sm) { return smooth_fpam(c, (real)u, (real)sm); }
void vine_oracle_actuation(const VineConfig* c, const double* q, const double* qd, double cart_vy, double u_rail,
                           double u_used, const double* scale, double* prev_cart_vel, double* prev_cart_vel_err,
                           double* eff) {
    real rq[ND], rqd[ND], sc[20], pv = (real)*prev_cart_vel, pe = (real)*prev_cart_vel_err, e[ND];
    for (int i = 0; i < ND; ++i) { rq[i] = (real)q[i]; rqd[i] = (real)qd[i]; }
    for (int i = 0; i < 20; ++i) sc[i] = scale ? (real)scale[i] : (real)1;
    actuation(c, rq, rqd, (real)cart_vy, (real)u_rail, (real)u_used, sc, &pv, &pe, e, NULL);
    for (int i = 0; i < ND; ++i) eff[i] = e[i];
    *prev_cart_vel = pv; *prev_cart_vel_err = pe;
}
double vine_oracle_reward(const VineConfig* c, double dist, int reached, double tip_vy, double tip_vz, double u_rail,
                          double u_fpam, double prev_u_rail, double smoothed, int limit_hit, int tip_limit_hit,
                          double cart_y, double contact, double* rm13) {
    real rm[VINE_NUM_REWARDS];
    real t = reward_terms(c, (real)dist, reached, (real)tip_vy, (real)tip_vz, (real)u_rail, (real)u_fpam,
                          (real)prev_u_rail, (real)smoothed, limit_hit, tip_limit_hit, (real)cart_y, (real)contact, rm);
    for (int i = 0; i < VINE_NUM_REWARDS; ++i) rm13[i] = rm[i];
    return t;
}
int64_t vine_oracle_reset_logic(const VineConfig* c, int64_t reset_in, int64_t progress, int reached, int limit_hit,
                                int tip_limit_hit, int nonzero_contact) {
    return reset_logic(c, reset_in, progress, reached, limit_hit, tip_limit_hit, nonzero_contact);
}

/* ------------------------------------------------------------------------- */
struct VineHandle {
    VineConfig cfg;
    Model model;
    int n, num_obs, form;
    real* st;            /* [VF_COUNT][n] in `real` */
    float* st_f32;       /* float mirror handed out by vine_state_ptr (== st when real is float) */
    int owns_st, owns_f32;
    int64_t step_count;
    const float* reset_values;
    float* reward_matrix;
};
#define ST(h, f, e) ((h)->st[(size_t)(f) * (h)->n + (e)])

static void sync_mirror(VineHandle* h) {
    if (sizeof(real) == 4) return;
    size_t cnt = (size_t)VF_COUNT * h->n;
    for (size_t i = 0; i < cnt; ++i) h->st_f32[i] = (float)h->st[i];
}
/* f64 build: push edits made through the float mirror back into the double state */
void vine_oracle_pull_mirror(VineHandle* h) {
    if (sizeof(real) == 4) return;
    size_t cnt = (size_t)VF_COUNT * h->n;
    for (size_t i = 0; i < cnt; ++i) h->st[i] = (real)h->st_f32[i];
}
void* vine_oracle_state(VineHandle* h) { return h->st; }
/* probe switches (see Model); oracle-only, used by the physics-assumption sweep in tests/test_oracle_physics.py */
void vine_oracle_set_probe(VineHandle* h, double armature, double vmax_link, double vmax_joint, int effort_first_substep_only) {
    h->model.armature = (real)armature; h->model.vmax_link = (real)vmax_link; h->model.vmax_joint = (real)vmax_joint;
    h->model.effort_first_substep_only = effort_first_substep_only;
}
void vine_oracle_set_formulation(VineHandle* h, int form) { h->form = form; }

static int validate(const VineConfig* c) {
    if (!c) return fail(VINE_ERR_INVALID_ARG, "cfg is NULL");
    if (c->abi_version != VINE_ABI_VERSION) return fail(VINE_ERR_INVALID_ARG, "abi_version mismatch");
    if (c->num_envs <= 0) return fail(VINE_ERR_INVALID_ARG, "num_envs must be positive");
    if (c->control_freq_inv <= 0 || c->substeps <= 0) return fail(VINE_ERR_INVALID_ARG, "control_freq_inv/substeps must be positive");
    if (c->action_delay < 0 || c->action_delay > VINE_MAX_DELAY) return fail(VINE_ERR_INVALID_ARG, "ACTION_DELAY out of range");
    if (vine_num_obs(c) < 0) return VINE_ERR_UNSUPPORTED;
    return VINE_OK;
}

int vine_create(const VineConfig* cfg, int device_id, float* state_storage, VineHandle** out) {
    int rc = validate(cfg);
    if (rc) return rc;
    if (!out) return fail(VINE_ERR_INVALID_ARG, "out is NULL");
    if (device_id != -1) return fail(VINE_ERR_INVALID_ARG, "the oracle only runs with device_id == -1");
    VineHandle* h = (VineHandle*)calloc(1, sizeof *h);
    if (!h) return fail(VINE_ERR_ALLOC, "calloc");
    h->cfg = *cfg; h->n = cfg->num_envs; h->num_obs = vine_num_obs(cfg); h->form = FORM_ABS;
    model_init(&h->model, cfg);
    size_t cnt = (size_t)VF_COUNT * h->n;
    if (sizeof(real) == 4 && state_storage) { h->st = (real*)state_storage; memset(h->st, 0, cnt * sizeof(real)); }
    else { h->st = (real*)calloc(cnt, sizeof(real)); h->owns_st = 1; }
    if (sizeof(real) == 4) { h->st_f32 = (float*)h->st; }
    else if (state_storage) { h->st_f32 = state_storage; memset(h->st_f32, 0, cnt * sizeof(float)); }
    else { h->st_f32 = (float*)calloc(cnt, sizeof(float)); h->owns_f32 = 1; }
    if (!h->st || !h->st_f32) return fail(VINE_ERR_ALLOC, "state alloc");
    /* Initial asset pose: all DOFs zero (V5:440-445), body states from FK; the shelf starts at
     * (0, 0.2, 0) (V5:468-470); targets sampled at construction (V5:179) are overwritten by the
     * all-env reset inside the first step (VT:275-276), so they start at zero here. */
    real q0[ND] = {0}, tip[4];
    tip_kinematics(&h->model, q0, q0, tip);
    for (int e = 0; e < h->n; ++e) {
        ST(h, VF_TIP_Y, e) = tip[0]; ST(h, VF_TIP_Z, e) = tip[1];
        ST(h, VF_PREV_TIP_Y, e) = tip[0]; ST(h, VF_PREV_TIP_Z, e) = tip[1];
        ST(h, VF_SHELF_Y, e) = (real)0.2; ST(h, VF_SHELF_Z, e) = 0;
        if (cfg->flags & VINE_FLAG_CREATE_PIPE) {   /* V5:482-484 initial pose, identity orientation (theta = 0) */
            ST(h, VF_PIPE_Y, e) = (real)-0.4; ST(h, VF_PIPE_Z, e) = (real)0.5;
            ST(h, VF_OBJ_ANGLE, e) = (real)-1.5707963267948966;
        }
    }
    sync_mirror(h);
    *out = h;
    return VINE_OK;
}
void vine_destroy(VineHandle* h) {
    if (!h) return;
    if (h->owns_st) free(h->st);
    if (h->owns_f32) free(h->st_f32);
    free(h);
}
float* vine_state_ptr(VineHandle* h) { return h ? h->st_f32 : NULL; }
int64_t vine_get_step_count(VineHandle* h) { return h->step_count; }
int vine_set_step_count(VineHandle* h, int64_t s) { h->step_count = s; return VINE_OK; }
int vine_bind_reset_values(VineHandle* h, const float* v) { h->reset_values = v; return VINE_OK; }
int vine_bind_reward_matrix(VineHandle* h, float* rm) { h->reward_matrix = rm; return VINE_OK; }
int vine_set_introspection(VineHandle* h, int on) { (void)h; (void)on; return VINE_OK; }   /* the oracle always stores every field */

/* vine_stats: the dashboard scalars of V5:1250-1322 from the state the last step left behind (plain loops, double). */
int vine_stats(VineHandle* h, const float* rew, const int64_t* progress, int64_t view, float* out, void* stream) {
    (void)stream;
    if (!h || !rew || !progress || !out || view < 0 || view >= h->n) return fail(VINE_ERR_INVALID_ARG, "bad argument to vine_stats");
    const VineConfig* c = &h->cfg;
    const int n = h->n;
    double s[VINE_NUM_STATS]; memset(s, 0, sizeof s);
    double mx_ty = -1e300, mx_tz = -1e300, mx_tv = -1e300, mx_rew = -1e300, agg_sum = 0;
    for (int e = 0; e < n; ++e) {
        double ty = ST(h, VF_TIP_Y, e), tz = ST(h, VF_TIP_Z, e), gy = ST(h, VF_TARGET_Y, e), gz = ST(h, VF_TARGET_Z, e);
        double dist = sqrt((ty - gy) * (ty - gy) + (tz - gz) * (tz - gz));
        double cy = ST(h, VF_CART_Y, e), tv = sqrt((double)ST(h, VF_TIP_VY, e) * ST(h, VF_TIP_VY, e) + (double)ST(h, VF_TIP_VZ, e) * ST(h, VF_TIP_VZ, e));
        double cm = ST(h, VF_CONTACT_MEAN, e);
        s[VS_DIST_MEAN] += dist; s[VS_TARGET_REACHED] += dist < (double)c->success_dist;
        s[VS_LIMIT_HIT] += (cy > (double)c->rail_soft_limit) || (cy < -(double)c->rail_soft_limit);
        s[VS_TIP_LIMIT_HIT] += ty < gy; s[VS_ABS_TIP_Y] += fabs(ty); s[VS_TIP_Z] += tz;
        if (fabs(ty) > mx_ty) mx_ty = fabs(ty);
        if (tz > mx_tz) mx_tz = tz;
        if (tv > mx_tv) mx_tv = tv;
        s[VS_TIP_VEL_MEAN] += tv;
        s[VS_U_RAIL_ABS] += fabs((double)ST(h, VF_U_RAIL, e)); s[VS_PREV_U_RAIL_ABS] += fabs((double)ST(h, VF_PREV_U_RAIL, e));
        s[VS_RAIL_FORCE_ABS] += fabs((double)ST(h, VF_RAIL_FORCE, e)); s[VS_U_FPAM_ABS] += fabs((double)ST(h, VF_U_FPAM, e));
        s[VS_SMOOTHED_ABS] += fabs((double)ST(h, VF_SMOOTHED_U, e)); s[VS_PROGRESS_MEAN] += (double)progress[e];
        s[VS_CONTACT_MEAN] += cm; s[VS_CONTACT_NONZERO] += cm > 0;
        agg_sum += ST(h, VF_AGG_REW, e);
        s[VS_REW_MEAN] += rew[e];
        if (rew[e] > mx_rew) mx_rew = rew[e];
    }
    static const int means[] = {VS_DIST_MEAN, VS_TARGET_REACHED, VS_LIMIT_HIT, VS_TIP_LIMIT_HIT, VS_ABS_TIP_Y, VS_TIP_Z,
                                VS_TIP_VEL_MEAN, VS_U_RAIL_ABS, VS_PREV_U_RAIL_ABS, VS_RAIL_FORCE_ABS, VS_U_FPAM_ABS,
                                VS_SMOOTHED_ABS, VS_PROGRESS_MEAN, VS_CONTACT_MEAN, VS_CONTACT_NONZERO, VS_REW_MEAN};
    for (size_t i = 0; i < sizeof means / sizeof means[0]; ++i) s[means[i]] /= n;
    s[VS_MAX_ABS_TIP_Y] = mx_ty; s[VS_MAX_TIP_Z] = mx_tz; s[VS_TIP_VEL_MAX] = mx_tv; s[VS_REW_MAX] = mx_rew;
    double am = agg_sum / n, var = 0;
    for (int e = 0; e < n; ++e) { double d = ST(h, VF_AGG_REW, e) - am; var += d * d; }
    s[VS_AGG_MEAN] = am; s[VS_AGG_STD] = n > 1 ? sqrt(var / (n - 1)) : 0;          /* torch.std: unbiased */
    const int e = (int)view;
    for (int i = 0; i < 6; ++i) { s[VS_VIEW0 + i] = ST(h, VF_Q0 + i, e); s[VS_VIEW0 + 6 + i] = ST(h, VF_QD0 + i, e); s[VS_VIEW0 + 12 + i] = ST(h, VF_PREV_Q0 + i, e); }
    static const int vf[10] = {VF_TIP_Y, VF_TIP_Z, VF_TIP_VY, VF_TIP_VZ, VF_PREV_TIP_Y, VF_PREV_TIP_Z, VF_CART_Y, VF_CART_VY, VF_TARGET_Y, VF_TARGET_Z};
    for (int i = 0; i < 10; ++i) s[VS_VIEW0 + 18 + i] = ST(h, vf[i], e);
    static const int vu[5] = {VF_U_FPAM, VF_SMOOTHED_U, VF_U_RAIL, VF_RAIL_FORCE, VF_CONTACT_MEAN};
    for (int i = 0; i < 5; ++i) s[VS_VIEW_U + i] = ST(h, vu[i], e);
    if (h->reward_matrix)
        for (int k = 0; k < VINE_NUM_REWARDS; ++k) {
            double sum = 0, mx = -1e300, mn = 1e300;
            for (int i = 0; i < n; ++i) { double v = h->reward_matrix[(size_t)i * VINE_NUM_REWARDS + k]; sum += v; if (v > mx) mx = v; if (v < mn) mn = v; }
            s[VS_TERM0 + 3 * k] = sum / n; s[VS_TERM0 + 3 * k + 1] = mx; s[VS_TERM0 + 3 * k + 2] = mn;
        }
    for (int i = 0; i < VINE_NUM_STATS; ++i) out[i] = (float)s[i];
    return VINE_OK;
}

/* reset_idx for one env, V5:774-839 + sample_target_positions V5:887-914.
 * Body states (tip, cart) are left untouched when STALE_BODY_STATE_AFTER_RESET (V5:796-797). */
static void reset_env(VineHandle* h, int e, uint64_t step) {
    const VineConfig* c = &h->cfg;
    real qn[ND], ty, tz, depth, pdepth;
    const real ten = (real)(10.0 * 3.14159265358979323846 / 180.0); /* math.radians(10), V5:778-779 */
    if (h->reset_values) {
        const float* v = h->reset_values + (size_t)e * 10;
        for (int k = 0; k < NL; ++k) qn[k + 1] = v[k];
        qn[0] = v[5]; pdepth = v[6]; ty = v[7]; tz = v[8]; depth = v[9];
    } else {
        uint32_t r0[4], r1[4], r2[4];
        rng4(c->seed, (uint32_t)(c->env_id_offset + e), step, RNG_RESET, 0, r0);
        rng4(c->seed, (uint32_t)(c->env_id_offset + e), step, RNG_RESET, 1, r1);
        rng4(c->seed, (uint32_t)(c->env_id_offset + e), step, RNG_RESET, 2, r2);
        float u[10] = {u01(r0[0]), u01(r0[1]), u01(r0[2]), u01(r0[3]), u01(r1[0]),
                       u01(r1[1]), u01(r1[2]), u01(r1[3]), u01(r2[0]), u01(r2[1])};
        for (int k = 0; k < NL; ++k) qn[k + 1] = -ten + ((real)2 * ten) * (real)u[k];
        qn[0] = (real)c->random_init_cart_min_y +
                ((real)c->random_init_cart_max_y - (real)c->random_init_cart_min_y) * (real)u[5];
        ty = (real)c->min_target_y + ((real)c->max_target_y - (real)c->min_target_y) * (real)u[7];
        tz = (real)c->min_target_z + ((real)c->max_target_z - (real)c->min_target_z) * (real)u[8];
        depth = (real)c->min_target_depth + ((real)c->max_target_depth - (real)c->min_target_depth) * (real)u[9];
        pdepth = (real)c->min_target_depth + ((real)c->max_target_depth - (real)c->min_target_depth) * (real)u[6];
    }
    if (!(c->flags & VINE_FLAG_RANDOMIZE_DOF_INIT)) for (int i = 0; i < ND; ++i) qn[i] = 0;  /* V5:790 */
    if (!(c->flags & VINE_FLAG_RANDOMIZE_TARGETS)) { ty = (real)c->max_target_y; tz = (real)c->min_target_z; } /* V5:911-912 */
    for (int i = 0; i < ND; ++i) {
        ST(h, VF_Q0 + i, e) = qn[i];
        ST(h, VF_QD0 + i, e) = 0;                    /* V5:793 */
        ST(h, VF_PREV_Q0 + i, e) = qn[i];            /* V5:794 */
    }
    ST(h, VF_PREV_TIP_Y, e) = ST(h, VF_TIP_Y, e);    /* V5:797 (stale tip) */
    ST(h, VF_PREV_TIP_Z, e) = ST(h, VF_TIP_Z, e);
    ST(h, VF_PREV_U_RAIL, e) = 0;                    /* V5:798 */
    ST(h, VF_PREV_CART_VEL_ERR, e) = 0;              /* V5:799 (prev_cart_vel is NOT reset) */
    ST(h, VF_AGG_REW, e) = 0;                        /* V5:810 */
    ST(h, VF_TARGET_Y, e) = ty; ST(h, VF_TARGET_Z, e) = tz; /* V5:813 */
    if (c->flags & VINE_FLAG_CREATE_SHELF) {         /* V5:816-839 */
        ST(h, VF_SHELF_Y, e) = ty + (-(real)0.2 + depth);
        ST(h, VF_SHELF_Z, e) = tz - (real)0.01;
        ST(h, VF_OBJ_DEPTH, e) = depth;
    }
    if (c->flags & VINE_FLAG_CREATE_PIPE) {          /* V5:841-885 */
        const real R = (real)(0.07 * 1.05);           /* PIPE_RADIUS, V5:88 */
        real ez = (real)1.0 - tz;                    /* effective_z = INIT_Z - target z */
        real deg = (((real)13199.0 * ez - (real)12276.0) * ez + (real)4045.0) * ez - (real)447.0;   /* polyval, V5:855-857 */
        real tp = deg * (real)(3.14159265358979323846 / 180.0);
        real ctp = (real)cos((double)tp), stp = (real)sin((double)tp);
        ST(h, VF_PIPE_Y, e) = ty + pdepth * ctp + R * stp;       /* V5:867-869 */
        ST(h, VF_PIPE_Z, e) = tz + pdepth * stp - R * ctp;       /* V5:868-870 */
        ST(h, VF_OBJ_DEPTH, e) = pdepth;                         /* V5:884 (overrides the shelf's entry) */
        ST(h, VF_OBJ_ANGLE, e) = tp;                             /* V5:885 */
    }
    if (!(c->flags & VINE_FLAG_STALE_BODY_STATE_AFTER_RESET)) {
        real qd0[ND] = {0}, tip[4];
        tip_kinematics(&h->model, qn, qd0, tip);
        ST(h, VF_TIP_Y, e) = tip[0]; ST(h, VF_TIP_Z, e) = tip[1]; ST(h, VF_TIP_VY, e) = 0; ST(h, VF_TIP_VZ, e) = 0;
        ST(h, VF_PREV_TIP_Y, e) = tip[0]; ST(h, VF_PREV_TIP_Z, e) = tip[1];
        ST(h, VF_CART_Y, e) = qn[0]; ST(h, VF_CART_VY, e) = 0;
    }
}

int vine_reset_idx(VineHandle* h, const int64_t* env_ids, int64_t n, float* rew, int64_t* reset, int64_t* progress,
                   void* stream) {
    (void)stream;
    if (!h || (!env_ids && n > 0)) return fail(VINE_ERR_INVALID_ARG, "null argument");
    for (int64_t i = 0; i < n; ++i) {
        int64_t e = env_ids[i];
        if (e < 0 || e >= h->n) return fail(VINE_ERR_INVALID_ARG, "env id out of range");
        /* reset counter word: distinct from in-step resets by the high bit of the step */
        reset_env(h, (int)e, (uint64_t)h->step_count | (1ull << 62));
        if (reset) reset[e] = 0;           /* V5:807 */
        if (progress) progress[e] = 0;     /* V5:808 */
        if (rew) rew[e] = 0;               /* V5:809 */
    }
    sync_mirror(h);
    return VINE_OK;
}

static void step_env(VineHandle* h, int e, const float* actions, float* obs, float* rew, int64_t* reset,
                     int64_t* progress, uint8_t* timeouts) {
    const VineConfig* c = &h->cfg;
    const Model* M = &h->model;
    const uint64_t step = (uint64_t)h->step_count;
    FLOP_CALL(2);
    FLOPS(2 + 6 + 4);      /* clamp, action rescale (V5:1458-1463), smoothing filter (V5:999-1005) */
    const int randomize = (c->flags & VINE_FLAG_VINE_RANDOMIZE) != 0;
    const int shelf = (c->flags & VINE_FLAG_CREATE_SHELF) != 0;
    const int pipe = (c->flags & VINE_FLAG_CREATE_PIPE) != 0;

    /* ---- VecTask.step: clamp (VT:333) ---- */
    real a0 = (real)actions[2 * e], a1 = (real)actions[2 * e + 1];
    real ca = (real)c->clip_actions;
    a0 = a0 < -ca ? -ca : (a0 > ca ? ca : a0);
    a1 = a1 < -ca ? -ca : (a1 > ca ? ca : a1);
    /* ---- pre_physics_step (V5:922-945) ---- */
    if (randomize) {                                       /* V5:930-932 */
        uint32_t r[4]; float n0, n1;
        rng4(c->seed, (uint32_t)(c->env_id_offset + e), step, RNG_ACTION_NOISE, 0, r);
        normal2(r[0], r[1], &n0, &n1);
        a0 += (real)c->action_noise_std * (real)n0;
        a1 += (real)c->action_noise_std * (real)n1;
    }
    real new_rail, new_fpam, u_rail, u_fpam;
    raw_actions_to_actions(c, a0, a1, &new_rail, &new_fpam); /* V5:935 */
    if (c->action_delay > 0) {                             /* FIFO: push newest, pop oldest (V5:936-937) */
        int slot = (int)(step % (uint64_t)c->action_delay);
        u_rail = ST(h, VF_FIFO0 + 2 * slot, e); u_fpam = ST(h, VF_FIFO0 + 2 * slot + 1, e);
        ST(h, VF_FIFO0 + 2 * slot, e) = new_rail; ST(h, VF_FIFO0 + 2 * slot + 1, e) = new_fpam;
    } else { u_rail = new_rail; u_fpam = new_fpam; }
    if (c->flags & VINE_FLAG_FORCE_U_FPAM) u_fpam = 0;               /* V5:1023 */
    if (c->flags & VINE_FLAG_FORCE_U_RAIL_VELOCITY) u_rail = 0;      /* V5:1025 */
    real smoothed = smooth_fpam(c, u_fpam, ST(h, VF_SMOOTHED_U, e)); /* V5:940 */
    real q[ND], qd[ND], prev_q[ND];
    for (int i = 0; i < ND; ++i) { q[i] = ST(h, VF_Q0 + i, e); qd[i] = ST(h, VF_QD0 + i, e); prev_q[i] = q[i]; } /* V5:943 */
    real tip[4] = {ST(h, VF_TIP_Y, e), ST(h, VF_TIP_Z, e), ST(h, VF_TIP_VY, e), ST(h, VF_TIP_VZ, e)};
    real prev_tip_y = tip[0], prev_tip_z = tip[1];                   /* V5:944 */
    real prev_u_rail = u_rail;                                       /* V5:945 */
    real cart_y = ST(h, VF_CART_Y, e), cart_vy = ST(h, VF_CART_VY, e);
    real pcv = ST(h, VF_PREV_CART_VEL, e), pce = ST(h, VF_PREV_CART_VEL_ERR, e);
    real contact = ST(h, VF_CONTACT, e), contact_sum = 0, rail_force = 0;
    real shelf_y = ST(h, VF_SHELF_Y, e), shelf_z = ST(h, VF_SHELF_Z, e);
    real pipe_y = ST(h, VF_PIPE_Y, e), pipe_z = ST(h, VF_PIPE_Z, e), pipe_tp = ST(h, VF_OBJ_ANGLE, e);
    real u_used = (c->flags & VINE_FLAG_USE_SMOOTHED_FPAM) ? smoothed : u_fpam; /* V5:1059 */
    const real hsub = (real)c->dt / (real)c->substeps;

    /* ---- control_freq_inv x [refresh, actuation, contact norm, simulate] (VT:338-356) ---- */
    for (int it = 0; it < c->control_freq_inv; ++it) {
        real scale[20];
        if (randomize) {                                             /* V5:1053-1055 */
            /* 20 factors per control iteration from 3 Philox calls: 16-bit uniforms (2 per 32-bit word) */
            for (int g = 0; g < 3; ++g) {
                uint32_t r[4];
                rng4(c->seed, (uint32_t)(c->env_id_offset + e), step, RNG_DYN_SCALE, (uint32_t)(it * 3 + g), r);
                for (int k = 0; k < 8 && g * 8 + k < 20; ++k) {
                    float u = (float)((r[k >> 1] >> (16 * (k & 1))) & 0xffffu) * (1.0f / 65536.0f);
                    scale[g * 8 + k] = (real)c->dyn_scale_min + ((real)c->dyn_scale_max - (real)c->dyn_scale_min) * (real)u;
                }
            }
        } else for (int k = 0; k < 20; ++k) scale[k] = 1;
        real eff[ND], cj[ND];
        FLOP_BUCKET(1); FLOP_CALL(1);
        actuation(c, q, qd, cart_vy, u_rail, u_used, scale, &pcv, &pce, eff, cj);
        FLOP_BUCKET(2);
        rail_force = eff[0];
        if (c->effort_limit > 0) {   /* DOF effort clamp of the simulator (assumption switch, include/vine.h) */
            const real lim = (real)c->effort_limit;
            for (int i = 1; i < ND; ++i) eff[i] = eff[i] > lim ? lim : (eff[i] < -lim ? -lim : eff[i]);
        }
        if (shelf) contact_sum += contact;                           /* VT:348-351 */
        contact = sim_step(M, h->form, c->substeps, cj, q, qd, eff, hsub, shelf, pipe, shelf_y, shelf_z, pipe_y, pipe_z,
                           pipe_tp);                                 /* gym.simulate, VT:356 */
        tip_kinematics(M, q, qd, tip);                               /* refreshed rigid-body states */
        cart_y = q[0]; cart_vy = qd[0];
    }

    /* ---- post_physics_step (V5:1110-1120) ---- */
    int64_t prog = progress[e] + 1;                                  /* V5:1111 */
    real agg = ST(h, VF_AGG_REW, e);
    real ty = ST(h, VF_TARGET_Y, e), tz = ST(h, VF_TARGET_Z, e);
    /* commit simulated state before a possible reset overwrites parts of it */
    for (int i = 0; i < ND; ++i) { ST(h, VF_Q0 + i, e) = q[i]; ST(h, VF_QD0 + i, e) = qd[i]; ST(h, VF_PREV_Q0 + i, e) = prev_q[i]; }
    ST(h, VF_TIP_Y, e) = tip[0]; ST(h, VF_TIP_Z, e) = tip[1]; ST(h, VF_TIP_VY, e) = tip[2]; ST(h, VF_TIP_VZ, e) = tip[3];
    ST(h, VF_CART_Y, e) = cart_y; ST(h, VF_CART_VY, e) = cart_vy;
    ST(h, VF_PREV_TIP_Y, e) = prev_tip_y; ST(h, VF_PREV_TIP_Z, e) = prev_tip_z;
    ST(h, VF_PREV_U_RAIL, e) = prev_u_rail;
    ST(h, VF_PREV_CART_VEL_ERR, e) = pce; ST(h, VF_PREV_CART_VEL, e) = pcv;
    int64_t rst = reset[e];
    if (rst != 0) {                                                  /* V5:1114-1116 */
        reset_env(h, e, step);
        rst = 0; prog = 0;                                           /* V5:807-808 */
        for (int i = 0; i < ND; ++i) { q[i] = ST(h, VF_Q0 + i, e); qd[i] = 0; prev_q[i] = q[i]; }
        tip[0] = ST(h, VF_TIP_Y, e); tip[1] = ST(h, VF_TIP_Z, e); tip[2] = ST(h, VF_TIP_VY, e); tip[3] = ST(h, VF_TIP_VZ, e);
        prev_tip_y = ST(h, VF_PREV_TIP_Y, e); prev_tip_z = ST(h, VF_PREV_TIP_Z, e);
        cart_y = ST(h, VF_CART_Y, e); cart_vy = ST(h, VF_CART_VY, e);
        prev_u_rail = 0; pce = 0; agg = 0;
        ty = ST(h, VF_TARGET_Y, e); tz = ST(h, VF_TARGET_Z, e);
    }
    real obj_depth = ST(h, VF_OBJ_DEPTH, e), obj_angle = ST(h, VF_OBJ_ANGLE, e);
    /* compute_observations (V5:1339-1390) */
    real o[VINE_MAX_OBS];
    int k = observations(c, q, qd, prev_q, tip, prev_tip_y, prev_tip_z, ty, tz, smoothed, prev_u_rail, obj_depth,
                         obj_angle, o);
    if (randomize) {                                                 /* V5:1388-1390 */
        for (int i = 0; i < k; i += 4) {
            uint32_t r[4]; float nn[4];
            rng4(c->seed, (uint32_t)(c->env_id_offset + e), step, RNG_OBS_NOISE, (uint32_t)(i / 4), r);
            normal2(r[0], r[1], &nn[0], &nn[1]); normal2(r[2], r[3], &nn[2], &nn[3]);
            for (int j = 0; j < 4 && i + j < k; ++j) o[i + j] += (real)c->obs_noise_std * (real)nn[j];
        }
    }
    /* compute_reward (V5:1218-1331) */
    real dy = tip[0] - ty, dz = tip[1] - tz;
    real dist = (real)sqrt((double)(dy * dy + dz * dz));             /* x components are both 0 */
    FLOPS(6 + 3 + 1);      /* distance, comparisons, aggregated reward */
    int reached = dist < (real)c->success_dist;                      /* V5:1228 */
    int limit_hit = (cart_y > (real)c->rail_soft_limit) || (cart_y < -(real)c->rail_soft_limit); /* V5:1232 */
    int tip_limit_hit = tip[0] < ty;                                 /* V5:1237 */
    real cmean = shelf ? contact_sum / (real)c->control_freq_inv : 0; /* V5:1242-1248 */
    int nonzero = cmean > 0;
    real rm[VINE_NUM_REWARDS];
    real total = reward_terms(c, dist, reached, tip[2], tip[3], u_rail, u_fpam, prev_u_rail, smoothed, limit_hit,
                              tip_limit_hit, cart_y, cmean, rm);
    agg += total;                                                    /* V5:1278 */
    rst = reset_logic(c, rst, prog, reached, limit_hit, tip_limit_hit, nonzero);        /* V5:1324 */
    /* ---- VecTask.step epilogue (VT:366-380) ---- */
    uint8_t to = (prog >= (int64_t)c->max_episode_length - 1) && (rst != 0);             /* VT:366 */
    real co = (real)c->clip_observations;
    for (int i = 0; i < k; ++i) {
        real v = o[i]; v = v < -co ? -co : (v > co ? co : v);        /* VT:374 */
        obs[(size_t)e * h->num_obs + i] = (float)v;
    }
    rew[e] = (float)total; reset[e] = rst; progress[e] = prog; timeouts[e] = to;
    if (h->reward_matrix) for (int i = 0; i < VINE_NUM_REWARDS; ++i) h->reward_matrix[(size_t)e * VINE_NUM_REWARDS + i] = (float)rm[i];
    /* persistent state */
    ST(h, VF_SMOOTHED_U, e) = smoothed; ST(h, VF_U_FPAM, e) = u_fpam; ST(h, VF_U_RAIL, e) = u_rail;
    ST(h, VF_PREV_U_RAIL, e) = prev_u_rail;
    ST(h, VF_PREV_CART_VEL, e) = pcv; ST(h, VF_PREV_CART_VEL_ERR, e) = pce;
    ST(h, VF_AGG_REW, e) = agg; ST(h, VF_CONTACT, e) = contact; ST(h, VF_CONTACT_MEAN, e) = cmean;
    ST(h, VF_RAIL_FORCE, e) = rail_force;
}

int vine_step(VineHandle* h, const float* actions, float* obs, float* rew, int64_t* reset, int64_t* progress,
              uint8_t* timeouts, void* stream) {
    (void)stream;
    if (!h || !actions || !obs || !rew || !reset || !progress || !timeouts)
        return fail(VINE_ERR_INVALID_ARG, "null argument to vine_step");
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int e = 0; e < h->n; ++e) step_env(h, e, actions, obs, rew, reset, progress, timeouts);
    h->step_count += 1;
    sync_mirror(h);
    return VINE_OK;
}
