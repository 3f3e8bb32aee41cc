// vine_hip.hip — Vine5LinkMovingBase env step for MI355X (gfx950), hand-written HIP.
//
// One kernel launch = one VecTask.step for all envs (reference call stack: vec_task.py:319-380 ->
// Vine5LinkMovingBase.py:922-945, 1028-1106, 1110-1120, 774-839, 1339-1390, 1218-1331, 1540-1558;
// `gym.simulate` (PhysX, closed) is replaced by a closed-form planar 6-DoF articulation solve).
//
// Design:
//  * env-major SoA state (`state[field * N + env]`): every load/store of a wave is one contiguous
//    256-B segment per field.
//  * the whole step (pre -> 4 x [actuation, 10 substeps] -> reset -> obs -> reward -> reset flags)
//    runs out of registers; HBM sees ~320 B per env-step.
//  * dynamics in absolute link angles: M_ij = a_ij cos(th_i - th_j) with constant a_ij, so the
//    mass matrix costs 5 sincos + 20 FMAs; 6x6 Cholesky fully unrolled in VGPRs.
//  * model constants travel in the kernarg segment (scalar loads into SGPRs), not in LDS.
//  * no host sync, no allocation, no device-wide barrier inside a step: graph-capturable.
//    The step counter (RNG counter / FIFO slot) lives in device memory and is advanced by the
//    last workgroup to finish (ticket), so a captured launch replays correctly.
//
// Exports the C ABI of include/vine.h.  There is no CPU fallback in this library.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/vine.h"
#include "../../include/vine_ppo.h"      // VineRolloutArgs (vine_step_rollout)

#define NL VINE_NUM_LINKS
#define ND VINE_NUM_DOFS
#ifndef VINE_STEP_THREADS
#define VINE_STEP_THREADS 256     // threads per workgroup of the step kernel: 4 waves, one per SIMD of a CU (sweep on MI355X at
                                  // 16384 envs: 64 -> 38.5 us, 128 -> 36.1, 192 -> 34.8, 256 -> 34.4, 320+ -> 50)
#endif

namespace {

thread_local char g_err[256];
int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return VINE_ERR_DEVICE;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

// Everything the kernels need, precomputed on the host from VineConfig; passed by value (kernarg).
struct DevParams {
    int n, num_obs, obs_type, cfi, substeps, max_len, delay;
    int glog;      // log2 of the step kernel's (power-of-two) grid: see step_of()
    unsigned flags, seed_lo, seed_hi, env_off;
    float hsub, dt, cdt, inv_dt, inv_cdt, clip_obs, clip_act;
    float fpam_min, fpam_span, rail_scale, damping, kq, cad, eff_lim;
    float soft_limit, p_gain, d_gain, rail_acc, alpha_inf, alpha_def, success_dist;
    float cart_min, cart_span, ty_min, ty_span, tz_min, tz_span, depth_min, depth_span, ty_max, tz_fixed;
    float dyn_min, dyn_span, obs_noise, act_noise;
    float g, mtot, s0, c0, L, z1;
    float b[NL], gb[NL], I[NL];
    float a[NL][NL];
    float K[NL], C[NL], bb[NL], B[NL];
    float rw[VINE_NUM_REWARDS];
    float inv_obs_scale[VINE_MAX_OBS];   // reciprocals computed in double on the host: one v_mul instead of a
                                        // ~10-instruction IEEE division per column (<= 1 ulp from the reference's `/`)
};

enum { RNG_RESET = 1, RNG_ACTION_NOISE = 2, RNG_DYN_SCALE = 3, RNG_OBS_NOISE = 4 };

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void rng4(const DevParams& P, unsigned env, unsigned long long step, unsigned purpose,
                                     unsigned idx, unsigned out[4]) {
    // env: local index; the key is the GLOBAL env id (VineConfig.env_id_offset), so a shard draws what the whole batch does
    philox4x32_10(env + P.env_off, (unsigned)step, purpose | ((unsigned)(step >> 32) << 8), idx, P.seed_lo, P.seed_hi, out);
}
// The step count the random streams are keyed by.  counters[0] = a base the host sets, counters[1] = workgroups of step
// launches that have FINISHED since then.  A step launch has a power-of-two grid of 2^glog workgroups, each of which adds
// one to counters[1] as its last act (an atomic without a return value): a workgroup of launch k reads a value in
// [k G, (k + 1) G) whatever the others are doing -- its own increment comes after its reads -- so counters[1] >> glog is
// k for every workgroup, with no "last one advances the counter" election.  (Rounds 1-3a: a ticket per workgroup with
// the last arrival writing step + 1 -- a returning atomic on one word from 256 workgroups, a barrier and a fence behind the
// workgroup's stores, and two dependent writes by the last workgroup: 2-3 us at the tail of a 24 us kernel.)
__device__ __forceinline__ unsigned long long step_of(const DevParams& P, const unsigned long long* counters) {
    return counters[0] + (counters[1] >> P.glog);
}
__device__ __forceinline__ void step_arrive(unsigned long long* counters) {
    __builtin_amdgcn_s_barrier();      // every wave of the workgroup has read the counters long ago; no fence: nothing is published
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(&counters[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ void normal2(unsigned a, unsigned b, float& n0, float& n1) {
    // Box-Muller on the hardware's log2 / sin / cos (v_log_f32, v_sin_f32, v_cos_f32 take the angle in revolutions, i.e.
    // u2 itself): ~10 instructions instead of the ~250 of logf + sinf + cosf with their range reductions; the step draws
    // five pairs per lane.  Absolute error of a deviate ~1e-6, scaled by the noise amplitudes (1e-2 .. 1e-3): far below
    // the fp32 rounding of the observation it is added to; the oracle keeps libm (tests: tolerance).
    float u1 = 1.0f - u01(a), u2 = u01(b);
    float r = sqrtf(-2.0f * __logf(u1));
    float t = 6.283185307179586f * u2;
    n0 = r * __cosf(t);
    n1 = r * __sinf(t);
}

// Articulation state in absolute coordinates: cart (y, vy), link angles th_k = sum_{i<=k} q_i, rates w_k.
// sn/cs = sin/cos of the WORLD link angle phi_k = phi0 + th_k, carried along by exact-to-rounding incremental
// rotations (see substep) and re-synchronised from th once per env step.
struct Dyn {
    float y, vy, th[NL], w[NL], sn[NL], cs[NL];
};

__device__ __forceinline__ void dyn_sync_trig(const DevParams& P, Dyn& s) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        float st, ct;
        sincosf(s.th[i], &st, &ct);
        s.sn[i] = P.s0 * ct + P.c0 * st;
        s.cs[i] = P.c0 * ct - P.s0 * st;
    }
}

// One semi-implicit Euler substep.  eff[6] = held efforts (rail force, joint torques), cj[6] = per-DOF
// damping, hc[6] = h * cj (implicit part), all constant over one `simulate`.
//   row 0:  mtot*ydd - sum_i b_i cos(phi_i) thdd_i = F - cj0*vy - sum_i b_i sin(phi_i) w_i^2
//   row i: -b_i cos(phi_i) ydd + sum_j a_ij cos(th_i-th_j) thdd_j
//            = T_i - T_{i+1} - cad*I_i*w_i - sum_j a_ij sin(th_i-th_j) w_j^2 + g b_i sin(phi_i)
// Constants of one `simulate` call (10 substeps): diagonal of the mass matrix with the implicit damping folded in,
// the negated sub-diagonal damping terms, the pivot of the cart column and the pre-scaled cart-column coefficients.
struct SimConst {
    float adiag[NL];   // a_ii + hc[i+1] + hc[i+2] (+ h*cad*I_i)
    float ncn[NL];     // -hc[i+1]: added to A[i+1][i] (coupling of neighbouring joints through the damping)
    float a00, p0;     // mtot + hc[0], rsqrt of it
    float bp[NL];      // -b_i * p0:  L_i0 = bp_i * cos(phi_i)
};

template <bool IMPLICIT, bool EXTRAS>
__device__ __forceinline__ void make_sim_const(const DevParams& P, const float (&hc)[ND], SimConst& k) {
    k.a00 = P.mtot + (IMPLICIT ? hc[0] : 0.0f);
    k.p0 = __builtin_amdgcn_rsqf(k.a00);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const float cn = (IMPLICIT && i < NL - 1) ? hc[i + 2] : 0.0f;
        k.adiag[i] = P.a[i][i] + (IMPLICIT ? hc[i + 1] + cn : 0.0f);
        if (IMPLICIT && EXTRAS) k.adiag[i] += P.hsub * P.cad * P.I[i];
        k.ncn[i] = IMPLICIT ? -hc[i + 1] : 0.0f;
        k.bp[i] = -P.b[i] * k.p0;
    }
}

template <bool IMPLICIT, bool CONTACT, bool EXTRAS>   // EXTRAS: joint stiffness / link angular damping switched on
__device__ __forceinline__ void substep(const DevParams& P, Dyn& s, const float (&eff)[ND], const float (&cj)[ND],
                                        const SimConst& K, const float (&qa)[ND]
                                        ) {
    const float (&sp)[NL] = s.sn;   // sin(phi_i), cos(phi_i)
    const float (&cp)[NL] = s.cs;
    float w2[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) w2[i] = s.w[i] * s.w[i];
    float T[NL + 1];
    T[0] = eff[1] - cj[1] * s.w[0];
#pragma unroll
    for (int i = 1; i < NL; ++i) T[i] = eff[i + 1] - cj[i + 1] * (s.w[i] - s.w[i - 1]);
    if (EXTRAS) {
        T[0] -= P.kq * s.th[0];
#pragma unroll
        for (int i = 1; i < NL; ++i) T[i] -= P.kq * (s.th[i] - s.th[i - 1]);
    }
    T[NL] = 0.0f;

    float A[ND][ND];  // lower triangle used; column 0 is written already factorised (L_i0)
    float r[ND];
    r[0] = eff[0] - cj[0] * s.vy;
    if (CONTACT) r[0] += qa[0];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        A[i + 1][0] = K.bp[i] * cp[i];                           // L_i0 = A_i0 * p0 = (-b_i p0) cos(phi_i)
        r[0] -= P.b[i] * sp[i] * w2[i];
        r[i + 1] = T[i] - T[i + 1] + P.gb[i] * sp[i];
        if (EXTRAS) r[i + 1] -= P.cad * P.I[i] * s.w[i];
        if (CONTACT) r[i + 1] += qa[i + 1];
        A[i + 1][i + 1] = K.adiag[i];
    }
    // a_ij = L*b_i for every j < i: fold it into row i's sin/cos once, then each pair costs 2+2+2 operations;
    // the implicit-damping coupling -hc of neighbouring joints rides in as the accumulator's start value
#pragma unroll
    for (int i = 1; i < NL; ++i) {
        const float Ci = P.a[i][0] * cp[i], Si = P.a[i][0] * sp[i];
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const float base = (IMPLICIT && j == i - 1) ? K.ncn[i] : 0.0f;
            A[i + 1][j + 1] = fmaf(Si, sp[j], fmaf(Ci, cp[j], base));   // a_ij cos(phi_i - phi_j) [- hc]
            const float asd = Si * cp[j] - Ci * sp[j];                  // a_ij sin(phi_i - phi_j)
            r[i + 1] -= asd * w2[j];
            r[j + 1] += asd * w2[i];
        }
    }
    // Cholesky A = L L^T in place (lower), then forward/back substitution; fully unrolled.  Column 0 is known in
    // closed form (constant pivot), so the factorisation starts at column 1.
    float inv[ND];
    inv[0] = K.p0;
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        float d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
        float ri = __builtin_amdgcn_rsqf(d);  // raw v_rsq_f32 (1 ulp); pivots are O(1e-4..1), never denormal
        inv[j] = ri;
#pragma unroll
        for (int i = j + 1; i < ND; ++i) {
            float t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= A[i][k] * A[j][k];
            A[i][j] = t * ri;
        }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        float t = r[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t -= A[i][k] * r[k];
        r[i] = t * inv[i];
    }
#pragma unroll
    for (int i = ND - 1; i >= 0; --i) {
        float t = r[i];
#pragma unroll
        for (int k = i + 1; k < ND; ++k) t -= A[k][i] * r[k];
        r[i] = t * inv[i];
    }
    s.vy += P.hsub * r[0];
    s.y += P.hsub * s.vy;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        s.w[i] += P.hsub * r[i + 1];
        const float d = P.hsub * s.w[i];
        s.th[i] += d;
        // rotate (sn, cs) by d: |d| = h*|w| < 0.06 rad even at 64 rad/s, so the degree-3/4 Taylor polynomials
        // are exact to float rounding (dropped terms d^5/120 < 7e-9, d^6/720 < 7e-11)
        const float d2 = d * d;
        const float cd = fmaf(d2, fmaf(d2, 1.0f / 24.0f, -0.5f), 1.0f);
        const float sd = d * fmaf(d2, -1.0f / 6.0f, 1.0f);
        const float s_old = s.sn[i], c_old = s.cs[i];
        s.sn[i] = fmaf(s_old, cd, c_old * sd);
        s.cs[i] = fmaf(c_old, cd, -(s_old * sd));
    }
}

// Planar frictionless penalty contact against the shelf (CREATE_SHELF; DESIGN.md section 3, oracle/vine_oracle.c
// shelf_contact): (a) 6 points of every link rectangle vs the two boards, (b) the two front corners of the
// `shelf_link` strip vs every link rectangle.  Writes generalised forces in ABSOLUTE coordinates
// (d x / d th_i = L n_i for i < k, z n_k + y d_k for i = k) and returns |F| on the strip.
//
// Broad phase (round 3): the narrow phase above is ~230 instructions per link, 40 times per env step, and almost always
// finds nothing.  Every link first takes a CONSERVATIVE bounding test against each board (a few instructions: the link
// rectangle lies within LINK_REACH of the segment between its two joints); the narrow phase of a (link, board) pair runs
// only where that test cannot exclude contact.  A wave skips a block none of its lanes needs (the compiler branches
// on EXEC == 0 around it), so links far from the obstacle -- the proximal ones nearly always -- cost the test alone.
// Culled pairs contribute exactly zero, so the result is bit-identical to the unculled evaluation.
#define CONTACT_K 2000.0f
#define CONTACT_C 2.0f
#define LINK_Y0 (-0.0381f)
#define LINK_Y1 0.0719f
#define LINK_REACH 0.078f   // lateral half-extent 0.0719 + link_0's 5.75 mm axial overhang beyond its joints, rounded up
// One link against the shelf: the link's joint position (py, pz) and velocity (pvy, pvz), sin / cos of its world angle,
// its rate, the axial extent [z0, z1] of its rectangle.  Adds the force on the link (fy, fz), its moment about the
// link's joint (mom) and the reaction on the `shelf_link` strip (sfy, sfz); returns whether any narrow phase ran.
__device__ __forceinline__ bool shelf_link_contact(const DevParams& P, float z0, float z1, float py, float pz, float pvy,
                                                   float pvz, float sp, float cp, float om, float shelf_y, float shelf_z,
                                                   float& fy_tot, float& fz_tot, float& mom, float& strip_fy,
                                                   float& strip_fz) {
    const float board[2][4] = {{-0.001f, 0.0f, 0.1995f, 0.005f}, {0.0f, 0.2f, 0.2f, 0.005f}};
    const float dy = -sp, dz = cp, ly = cp, lz = sp;     // link axis d, lateral l; n = d(d)/d(phi) = (-cp, -sp) = -l
    // every shelf shape lies at y <= shelf_y + 0.2; board A / the strip around z = shelf_z, board B around shelf_z + 0.2
    const float ycut = shelf_y + 0.2f;
    const float a_lo = shelf_z - 0.005f, a_hi = shelf_z + 0.005f, b_lo = shelf_z + 0.195f, b_hi = shelf_z + 0.205f;
    const float qy = py + P.L * dy, qz = pz + P.L * dz;  // the next joint
    const float ylo = fminf(py, qy) - LINK_REACH;
    const float zlo = fminf(pz, qz) - LINK_REACH, zhi = fmaxf(pz, qz) + LINK_REACH;
    const bool near_a = ylo < ycut && zlo < a_hi && zhi > a_lo;      // board A and the strip's corners
    const bool near_b = ylo < ycut && zlo < b_hi && zhi > b_lo;
    // the boards are 0.2 m apart in z and 1 cm thick: a point can be inside the one on its side of z = shelf_z + 0.1 only,
    // so ONE test body per point (the board selected by the point's height) instead of one block per board -- the waves
    // that set the kernel's duration hold envs near board A and envs near board B and ran both blocks
    if (near_a || near_b) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float yl = e ? LINK_Y1 : LINK_Y0;
                const float zl = (t == 0) ? z0 : (t == 1 ? 0.5f * (z0 + z1) : z1);
                const float ry = zl * dy + yl * ly, rz = zl * dz + yl * lz;
                const float wy = py + ry, wz = pz + rz;
                const bool bsel = wz - shelf_z > 0.1f;
                if (bsel ? near_b : near_a) {
                    const float ddy = wy - (shelf_y + (bsel ? board[1][0] : board[0][0]));
                    const float ddz = wz - (shelf_z + (bsel ? board[1][1] : board[0][1]));
                    const float ey = (bsel ? board[1][2] : board[0][2]) - fabsf(ddy), ez = board[0][3] - fabsf(ddz);
                    if (ey > 0.0f && ez > 0.0f) {
                        const float vy = pvy - om * rz, vz = pvz + om * ry;
                        float fy = 0.0f, fz = 0.0f;
                        if (ey < ez) {
                            const float sg = (ddy > 0.0f) ? 1.0f : -1.0f;
                            fy = sg * fmaxf(CONTACT_K * ey - CONTACT_C * sg * vy, 0.0f);
                        } else {
                            const float sg = (ddz > 0.0f) ? 1.0f : -1.0f;
                            fz = sg * fmaxf(CONTACT_K * ez - CONTACT_C * sg * vz, 0.0f);
                        }
                        fy_tot += fy; fz_tot += fz;
                        mom += -rz * fy + ry * fz;          // F . (z n_k + y d_k) = r x F about joint k
                    }
                }
            }
        }
    }
    if (near_a) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float wy = shelf_y + 0.2f, wz = shelf_z + (e ? 0.005f : -0.005f);
            const float ry = wy - py, rz = wz - pz;
            const float zl = ry * dy + rz * dz, yl = ry * ly + rz * lz;
            if (zl > z0 && zl < z1 && yl > LINK_Y0 && yl < LINK_Y1) {
                float dep = zl - z0, ny = -dy, nz = -dz;
                if (z1 - zl < dep) { dep = z1 - zl; ny = dy; nz = dz; }
                if (yl - LINK_Y0 < dep) { dep = yl - LINK_Y0; ny = -ly; nz = -lz; }
                if (LINK_Y1 - yl < dep) { dep = LINK_Y1 - yl; ny = ly; nz = lz; }
                const float vy = pvy - om * rz, vz = pvz + om * ry;
                const float f = fmaxf(CONTACT_K * dep + CONTACT_C * (vy * ny + vz * nz), 0.0f);
                strip_fy += f * ny; strip_fz += f * nz;
                const float fy = -f * ny, fz = -f * nz;
                fy_tot += fy; fz_tot += fz;
                mom += -rz * fy + ry * fz;
            }
        }
    }
    return near_a || near_b;
}

__device__ __forceinline__ float shelf_contact(const DevParams& P, const Dyn& s, float shelf_y, float shelf_z,
                                               float (&qa)[ND]) {
    float strip_fy = 0.0f, strip_fz = 0.0f;
    float py = s.y, pz = P.z1, pvy = s.vy, pvz = 0.0f;
    float Fy[NL], Fz[NL], ny_[NL], nz_[NL];
    bool any = false;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const float sp = s.sn[k], cp = s.cs[k];              // sin/cos of the world link angle
        const float om = s.w[k];
        const float z0 = (k == 0) ? -0.00575f : 0.0f, z1 = (k == 0) ? 0.09425f : P.L;
        float fy_tot = 0.0f, fz_tot = 0.0f, mom = 0.0f;
        ny_[k] = -cp; nz_[k] = -sp;
        any |= shelf_link_contact(P, z0, z1, py, pz, pvy, pvz, sp, cp, om, shelf_y, shelf_z, fy_tot, fz_tot, mom, strip_fy,
                                  strip_fz);
        Fy[k] = fy_tot; Fz[k] = fz_tot;
        qa[k + 1] = mom;
        py += P.L * (-sp); pz += P.L * cp;
        pvy += P.L * om * (-cp); pvz += P.L * om * (-sp);
    }
    if (!any) {          // (qa[1..5] are zero already)
        qa[0] = 0.0f;
        return 0.0f;
    }
    // forces on distal links act on joint i through the lever L n_i
    float sy = 0.0f, sz = 0.0f;
#pragma unroll
    for (int i = NL - 1; i >= 0; --i) {
        qa[i + 1] += P.L * (ny_[i] * sy + nz_[i] * sz);
        sy += Fy[i]; sz += Fz[i];
    }
    qa[0] = sy;
    return sqrtf(strip_fy * strip_fy + strip_fz * strip_fz);
}

// Planar pipe contact (CREATE_PIPE): the tube of assets/urdf/pipe as two wall rectangles in the pipe frame
// (oracle/vine_oracle.c pipe_contact).  ADDS its generalised forces to qa.
// Broad phase, three levels per link: (1) bounding circles of the link and of the whole tube (7 instructions);
// (2) the link rectangle's exact axis-aligned box in the pipe frame against each wall's box; (3) the narrow phase of
// the (link, wall) pairs that are left.  A vine reaching INTO the tube sits between the walls: level 2 is what keeps it
// out of the narrow phase until it actually comes within a rounding margin of a wall.
#define PIPE_LEN 0.34125f
#define PIPE_WALL 0.00525f
#define PIPE_OUTER 0.1554f
#define PIPE_CULL_EPS 1.0e-5f      // the box tests use other (equivalent) expressions than the narrow phase: rounding margin
struct PipePose { float y, z, ct, st, ccy, ccz; };      // origin, cos / sin of the tube's axis angle, centre of its box
__device__ __forceinline__ PipePose pipe_pose(float pipe_y, float pipe_z, float ct, float st) {
    const float hcy = 0.5f * PIPE_OUTER, hcz = 0.5f * PIPE_LEN;
    return PipePose{pipe_y, pipe_z, ct, st, pipe_y + hcy * ct - hcz * st, pipe_z + hcy * st + hcz * ct};
}
__device__ __forceinline__ unsigned pipe_broad_phase(float z0, float z1, float py, float pz, float sp, float cp,
                                                     const PipePose& T);
// nearbits: pipe_broad_phase of this link when the caller has it already (the four-lane kernel), PIPE_BROAD_HERE otherwise
#define PIPE_BROAD_HERE 0xffffffffu
__device__ __forceinline__ bool pipe_link_contact(const DevParams& P, float z0, float z1, float py, float pz, float pvy,
                                                  float pvz, float sp, float cp, float om, const PipePose& T, float& fy_tot,
                                                  float& fz_tot, float& mom, unsigned nearbits = PIPE_BROAD_HERE) {
    const float wall_lo[2] = {0.0f, PIPE_OUTER - PIPE_WALL};
    const float pipe_y = T.y, pipe_z = T.z, ct = T.ct, st = T.st;
    const float dy = -sp, dz = cp, ly = cp, lz = sp;
    // broad phase (pipe_broad_phase below): (1) bounding circles of the link and of the tube, (2) the rectangle's box in the
    // pipe frame against each wall's box
    if (nearbits == PIPE_BROAD_HERE) nearbits = pipe_broad_phase(z0, z1, py, pz, sp, cp, T);
    const bool near0 = (nearbits & 1u) != 0, near1 = (nearbits & 2u) != 0;
    if (!(near0 || near1)) return false;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float yl = e ? LINK_Y1 : LINK_Y0;
            const float zl = (t == 0) ? z0 : (t == 1 ? 0.5f * (z0 + z1) : z1);
            const float ry = zl * dy + yl * ly, rz = zl * dz + yl * lz;
            const float gy = py + ry - pipe_y, gz = pz + rz - pipe_z;
            const float pyl = gy * ct + gz * st, pzl = -gy * st + gz * ct;
            // the walls are disjoint in the pipe frame's y ((0, 0.00525) and (0.15015, 0.1554)): a point can be inside
            // the one on its side of the tube's axis only, so ONE test body per point instead of one per (point, wall) --
            // same arithmetic for the wall that is tested, half the worst-case instruction stream of a wave
            {
                const bool w1 = pyl > 0.5f * PIPE_OUTER;
                const float cw = w1 ? (wall_lo[1] + 0.5f * PIPE_WALL) : (wall_lo[0] + 0.5f * PIPE_WALL);
                if (w1 ? near1 : near0) {
                    const float ddy = pyl - cw, ddz = pzl - 0.5f * PIPE_LEN;
                    const float ey = 0.5f * PIPE_WALL - fabsf(ddy), ez = 0.5f * PIPE_LEN - fabsf(ddz);
                    if (ey > 0.0f && ez > 0.0f) {
                        const float vy = pvy - om * rz, vz = pvz + om * ry;
                        const float vyl = vy * ct + vz * st, vzl = -vy * st + vz * ct;
                        float fyl = 0.0f, fzl = 0.0f;
                        if (ey < ez) {
                            const float sg = (ddy > 0.0f) ? 1.0f : -1.0f;
                            fyl = sg * fmaxf(CONTACT_K * ey - CONTACT_C * sg * vyl, 0.0f);
                        } else {
                            const float sg = (ddz > 0.0f) ? 1.0f : -1.0f;
                            fzl = sg * fmaxf(CONTACT_K * ez - CONTACT_C * sg * vzl, 0.0f);
                        }
                        const float fy = fyl * ct - fzl * st, fz = fyl * st + fzl * ct;
                        fy_tot += fy; fz_tot += fz;
                        mom += -rz * fy + ry * fz;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        if (w == 0 ? near0 : near1) {
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) {
                if (!(nearbits & ((cidx & 2) ? 8u : 4u))) continue;      // this end of the tube is outside the link's box
                const float pyl = wall_lo[w] + ((cidx & 1) ? PIPE_WALL : 0.0f), pzl = (cidx & 2) ? PIPE_LEN : 0.0f;
                const float ry = pipe_y + pyl * ct - pzl * st - py, rz = pipe_z + pyl * st + pzl * ct - pz;
                const float zl = ry * dy + rz * dz, yl = ry * ly + rz * lz;
                if (zl > z0 && zl < z1 && yl > LINK_Y0 && yl < LINK_Y1) {
                    float dep = zl - z0, ny = -dy, nz = -dz;
                    if (z1 - zl < dep) { dep = z1 - zl; ny = dy; nz = dz; }
                    if (yl - LINK_Y0 < dep) { dep = yl - LINK_Y0; ny = -ly; nz = -lz; }
                    if (LINK_Y1 - yl < dep) { dep = LINK_Y1 - yl; ny = ly; nz = lz; }
                    const float vy = pvy - om * rz, vz = pvz + om * ry;
                    const float f = fmaxf(CONTACT_K * dep + CONTACT_C * (vy * ny + vz * nz), 0.0f);
                    const float fy = -f * ny, fz = -f * nz;
                    fy_tot += fy; fz_tot += fz;
                    mom += -rz * fy + ry * fz;
                }
            }
        }
    }
    return true;
}

__device__ __forceinline__ void pipe_contact(const DevParams& P, const Dyn& s, float pipe_y, float pipe_z, float ct,
                                             float st, float (&qa)[ND]) {
    float py = s.y, pz = P.z1, pvy = s.vy, pvz = 0.0f;
    float Fy[NL], Fz[NL], ny_[NL], nz_[NL], mom_[NL];
    bool any = false;
    const PipePose T = pipe_pose(pipe_y, pipe_z, ct, st);
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const float sp = s.sn[k], cp = s.cs[k];
        const float om = s.w[k];
        const float z0 = (k == 0) ? -0.00575f : 0.0f, z1 = (k == 0) ? 0.09425f : P.L;
        float fy_tot = 0.0f, fz_tot = 0.0f, mom = 0.0f;
        ny_[k] = -cp; nz_[k] = -sp;
        any |= pipe_link_contact(P, z0, z1, py, pz, pvy, pvz, sp, cp, om, T, fy_tot, fz_tot, mom);
        Fy[k] = fy_tot; Fz[k] = fz_tot; mom_[k] = mom;
        py += P.L * (-sp); pz += P.L * cp;
        pvy += P.L * om * (-cp); pvz += P.L * om * (-sp);
    }
    if (!any) return;        // nothing to add
    float sy = 0.0f, sz = 0.0f;
#pragma unroll
    for (int i = NL - 1; i >= 0; --i) {
        qa[i + 1] += mom_[i] + P.L * (ny_[i] * sy + nz_[i] * sz);
        sy += Fy[i]; sz += Fz[i];
    }
    qa[0] += sy;
}

// ---- cooperative narrow phases for the four-lanes-per-env kernel.  Under a trained policy the distal links sit AT the
// obstacle (inside the tube, at the shelf's edge): their narrow phases run in nearly every substep, and a wave pays for
// a whole link whenever one lane needs it.  For links 3 and 4 (whose state every lane of the quad holds: link 4 is
// replicated, link 3 is broadcast) the 6 rectangle points and the obstacle's corners are therefore SPLIT over the four
// lanes -- lane t takes point t (and point t + 4 on lanes 0, 1) and corner t of every wall / strip corner t -- and the
// partial forces are added with quad sums by the caller.  Same arithmetic per point / corner as the functions above; the
// broad-phase decisions are uniform over the quad (same inputs on all four lanes).
__device__ __forceinline__ void coop_point(int t, int slot, float z0, float z1, float& yl, float& zl, bool& valid) {
    // slot 0: point t  -> (edge, position) = (0, t) for t < 3, (1, 0) for t = 3;  slot 1: point t + 4 -> (1, t + 1), lanes 0, 1
    const int e = slot ? 1 : (t == 3), pos = slot ? t + 1 : (t == 3 ? 0 : t);
    valid = slot ? t < 2 : true;
    yl = e ? LINK_Y1 : LINK_Y0;
    zl = pos == 0 ? z0 : (pos == 1 ? 0.5f * (z0 + z1) : z1);
}
__device__ __forceinline__ void shelf_link_contact_coop(const DevParams& P, int t, float z0, float z1, float py, float pz,
                                                        float pvy, float pvz, float sp, float cp, float om, float shelf_y,
                                                        float shelf_z, float& fy_tot, float& fz_tot, float& mom,
                                                        float& strip_fy, float& strip_fz) {
    const float board[2][4] = {{-0.001f, 0.0f, 0.1995f, 0.005f}, {0.0f, 0.2f, 0.2f, 0.005f}};
    const float dy = -sp, dz = cp, ly = cp, lz = sp;
    const float ycut = shelf_y + 0.2f;
    const float a_lo = shelf_z - 0.005f, a_hi = shelf_z + 0.005f, b_lo = shelf_z + 0.195f, b_hi = shelf_z + 0.205f;
    const float qy = py + P.L * dy, qz = pz + P.L * dz;
    const float ylo = fminf(py, qy) - LINK_REACH;
    const float zlo = fminf(pz, qz) - LINK_REACH, zhi = fmaxf(pz, qz) + LINK_REACH;
    const bool near_a = ylo < ycut && zlo < a_hi && zhi > a_lo;
    const bool near_b = ylo < ycut && zlo < b_hi && zhi > b_lo;
    if (!(near_a || near_b)) return;
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        float yl, zl;
        bool valid;
        coop_point(t, slot, z0, z1, yl, zl, valid);
        const float ry = zl * dy + yl * ly, rz = zl * dz + yl * lz;
        const float wy = py + ry, wz = pz + rz;
        {   // one test body per point: the board on the point's side of z = shelf_z + 0.1 (see shelf_link_contact)
            const bool bsel = wz - shelf_z > 0.1f;
            const float ddy = wy - (shelf_y + (bsel ? board[1][0] : board[0][0]));
            const float ddz = wz - (shelf_z + (bsel ? board[1][1] : board[0][1]));
            const float ey = (bsel ? board[1][2] : board[0][2]) - fabsf(ddy), ez = board[0][3] - fabsf(ddz);
            if (valid && (bsel ? near_b : near_a) && ey > 0.0f && ez > 0.0f) {
                const float vy = pvy - om * rz, vz = pvz + om * ry;
                float fy = 0.0f, fz = 0.0f;
                if (ey < ez) {
                    const float sg = (ddy > 0.0f) ? 1.0f : -1.0f;
                    fy = sg * fmaxf(CONTACT_K * ey - CONTACT_C * sg * vy, 0.0f);
                } else {
                    const float sg = (ddz > 0.0f) ? 1.0f : -1.0f;
                    fz = sg * fmaxf(CONTACT_K * ez - CONTACT_C * sg * vz, 0.0f);
                }
                fy_tot += fy; fz_tot += fz;
                mom += -rz * fy + ry * fz;
            }
        }
    }
    if (near_a && t < 2) {                                   // strip corner t
        const float wy = shelf_y + 0.2f, wz = shelf_z + (t ? 0.005f : -0.005f);
        const float ry = wy - py, rz = wz - pz;
        const float zl = ry * dy + rz * dz, yl = ry * ly + rz * lz;
        if (zl > z0 && zl < z1 && yl > LINK_Y0 && yl < LINK_Y1) {
            float dep = zl - z0, ny = -dy, nz = -dz;
            if (z1 - zl < dep) { dep = z1 - zl; ny = dy; nz = dz; }
            if (yl - LINK_Y0 < dep) { dep = yl - LINK_Y0; ny = -ly; nz = -lz; }
            if (LINK_Y1 - yl < dep) { dep = LINK_Y1 - yl; ny = ly; nz = lz; }
            const float vy = pvy - om * rz, vz = pvz + om * ry;
            const float f = fmaxf(CONTACT_K * dep + CONTACT_C * (vy * ny + vz * nz), 0.0f);
            strip_fy += f * ny; strip_fz += f * nz;
            const float fy = -f * ny, fz = -f * nz;
            fy_tot += fy; fz_tot += fz;
            mom += -rz * fy + ry * fz;
        }
    }
}
// The broad phase of one (link, tube) pair -- levels 1 and 2 of pipe_link_contact, the same expressions -- as bit 0 / bit 1 =
// the link's box reaches wall 0 / wall 1.  Round 5: the four-lane kernel evaluates it ONCE per link, on the lane that owns
// the link (link 4: on every lane), and broadcasts the two bits to the lanes that share the link's narrow phase; the
// cooperative form used to repeat it on all four lanes for each of its links (~45 instructions x 3 links per substep).
__device__ __forceinline__ unsigned pipe_broad_phase(float z0, float z1, float py, float pz, float sp, float cp,
                                                     const PipePose& T) {
    const float pipe_y = T.y, pipe_z = T.z, ct = T.ct, st = T.st;
    const float dy = -sp, dz = cp;
    // radius of the tube's bounding circle + the link's (half-length 0.05, lateral reach 0.0719 about the axis midpoint)
    const float rsum = 0.18748f + 0.0877f + 1.0e-4f;       // hypot(0.0777, 0.170625) + hypot(0.05, 0.0719)
    const float my = py + 0.04425f * dy - T.ccy, mz = pz + 0.04425f * dz - T.ccz;     // axis midpoint - tube centre
    if (!(my * my + mz * mz < rsum * rsum)) return 0u;
    // level 2: the rectangle's box in the pipe frame.  Local axis / lateral directions, local joint position.
    const float dly = dy * ct + dz * st, dlz = -dy * st + dz * ct;          // d in the pipe frame; l = (dlz, -dly)
    const float gy0 = py - pipe_y, gz0 = pz - pipe_z;
    const float jy = gy0 * ct + gz0 * st, jz = -gy0 * st + gz0 * ct;
    const float ay0 = z0 * dly, ay1 = z1 * dly, by0 = LINK_Y0 * dlz, by1 = LINK_Y1 * dlz;
    const float az0 = z0 * dlz, az1 = z1 * dlz, bz0 = LINK_Y0 * -dly, bz1 = LINK_Y1 * -dly;
    const float ymin = jy + fminf(ay0, ay1) + fminf(by0, by1) - PIPE_CULL_EPS;
    const float ymax = jy + fmaxf(ay0, ay1) + fmaxf(by0, by1) + PIPE_CULL_EPS;
    const float zmin = jz + fminf(az0, az1) + fminf(bz0, bz1) - PIPE_CULL_EPS;
    const float zmax = jz + fmaxf(az0, az1) + fmaxf(bz0, bz1) + PIPE_CULL_EPS;
    const bool zin = zmin < PIPE_LEN && zmax > 0.0f;
    const bool near0 = zin && ymin < PIPE_WALL && ymax > 0.0f;
    const bool near1 = zin && ymin < PIPE_OUTER && ymax > PIPE_OUTER - PIPE_WALL;
    // bits 2, 3: the box reaches the tube's end z = 0 / z = PIPE_LEN.  A wall's corners lie on those two lines, and a corner
    // inside the link's rectangle lies inside the rectangle's box: a link deep inside the tube (a trained vine's links 3 and
    // 4, most of the time) cannot contain any corner, and its corner tests -- a third of a cooperative link's narrow phase --
    // are skipped.  Exact: the box carries the same rounding margin as above.
    return (near0 ? 1u : 0u) | (near1 ? 2u : 0u) | (zmin < 0.0f ? 4u : 0u) | (zmax > PIPE_LEN ? 8u : 0u);
}
__device__ __forceinline__ void pipe_link_contact_coop(const DevParams& P, int t, float z0, float z1, float py, float pz,
                                                       float pvy, float pvz, float sp, float cp, float om, const PipePose& T,
                                                       float& fy_tot, float& fz_tot, float& mom, unsigned nearbits) {
    const float wall_lo[2] = {0.0f, PIPE_OUTER - PIPE_WALL};
    const float pipe_y = T.y, pipe_z = T.z, ct = T.ct, st = T.st;
    const float dy = -sp, dz = cp, ly = cp, lz = sp;
    const bool near0 = (nearbits & 1u) != 0, near1 = (nearbits & 2u) != 0;      // (pipe_broad_phase of this link)
    if (!(near0 || near1)) return;
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        float yl, zl;
        bool valid;
        coop_point(t, slot, z0, z1, yl, zl, valid);
        const float ry = zl * dy + yl * ly, rz = zl * dz + yl * lz;
        const float gy = py + ry - pipe_y, gz = pz + rz - pipe_z;
        const float pyl = gy * ct + gz * st, pzl = -gy * st + gz * ct;
        {   // one test body per point: the wall on the point's side of the tube's axis (see pipe_link_contact)
            const bool w1 = pyl > 0.5f * PIPE_OUTER;
            const float cw = w1 ? (wall_lo[1] + 0.5f * PIPE_WALL) : (wall_lo[0] + 0.5f * PIPE_WALL);
            const float ddy = pyl - cw, ddz = pzl - 0.5f * PIPE_LEN;
            const float ey = 0.5f * PIPE_WALL - fabsf(ddy), ez = 0.5f * PIPE_LEN - fabsf(ddz);
            if (valid && (w1 ? near1 : near0) && ey > 0.0f && ez > 0.0f) {
                const float vy = pvy - om * rz, vz = pvz + om * ry;
                const float vyl = vy * ct + vz * st, vzl = -vy * st + vz * ct;
                float fyl = 0.0f, fzl = 0.0f;
                if (ey < ez) {
                    const float sg = (ddy > 0.0f) ? 1.0f : -1.0f;
                    fyl = sg * fmaxf(CONTACT_K * ey - CONTACT_C * sg * vyl, 0.0f);
                } else {
                    const float sg = (ddz > 0.0f) ? 1.0f : -1.0f;
                    fzl = sg * fmaxf(CONTACT_K * ez - CONTACT_C * sg * vzl, 0.0f);
                }
                const float fy = fyl * ct - fzl * st, fz = fyl * st + fzl * ct;
                fy_tot += fy; fz_tot += fz;
                mom += -rz * fy + ry * fz;
            }
        }
    }
    const bool my_end = (nearbits & ((t & 2) ? 8u : 4u)) != 0;      // can this lane's corner (end t >> 1 of the tube) lie in the link's box?
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        if ((w == 0 ? near0 : near1) && my_end) {            // corner t of wall w
            const float pyl = wall_lo[w] + ((t & 1) ? PIPE_WALL : 0.0f), pzl = (t & 2) ? PIPE_LEN : 0.0f;
            const float ry = pipe_y + pyl * ct - pzl * st - py, rz = pipe_z + pyl * st + pzl * ct - pz;
            const float zl = ry * dy + rz * dz, yl = ry * ly + rz * lz;
            if (zl > z0 && zl < z1 && yl > LINK_Y0 && yl < LINK_Y1) {
                float dep = zl - z0, ny = -dy, nz = -dz;
                if (z1 - zl < dep) { dep = z1 - zl; ny = dy; nz = dz; }
                if (yl - LINK_Y0 < dep) { dep = yl - LINK_Y0; ny = -ly; nz = -lz; }
                if (LINK_Y1 - yl < dep) { dep = LINK_Y1 - yl; ny = ly; nz = lz; }
                const float vy = pvy - om * rz, vz = pvz + om * ry;
                const float f = fmaxf(CONTACT_K * dep + CONTACT_C * (vy * ny + vz * nz), 0.0f);
                const float fy = -f * ny, fz = -f * nz;
                fy_tot += fy; fz_tot += fz;
                mom += -rz * fy + ry * fz;
            }
        }
    }
}

__device__ __forceinline__ float clampf(float v, float lim) { return fminf(fmaxf(v, -lim), lim); }

#define ST(f) st[(size_t)(f) * n + e]

// reset_idx for one env (Vine5LinkMovingBase.py:774-839, 887-914).  Writes the persistent fields and
// returns the new relative joint positions / target in registers.
__device__ __forceinline__ void reset_env(const DevParams& P, float* __restrict__ st, int n, int e,
                                          unsigned long long step, const float* __restrict__ reset_values,
                                          float (&qn)[ND], float& ty, float& tz) {
    const float ten = 0.17453292519943295f;  // math.radians(10)
    float depth, pdepth;
    if (reset_values) {
        const float* v = reset_values + (size_t)e * 10;
#pragma unroll
        for (int k = 0; k < NL; ++k) qn[k + 1] = v[k];
        qn[0] = v[5]; pdepth = v[6]; ty = v[7]; tz = v[8]; depth = v[9];
    } else {
        unsigned r0[4], r1[4], r2[4];
        rng4(P, (unsigned)e, step, RNG_RESET, 0, r0);
        rng4(P, (unsigned)e, step, RNG_RESET, 1, r1);
        rng4(P, (unsigned)e, step, RNG_RESET, 2, r2);
        qn[1] = -ten + (2.0f * ten) * u01(r0[0]);
        qn[2] = -ten + (2.0f * ten) * u01(r0[1]);
        qn[3] = -ten + (2.0f * ten) * u01(r0[2]);
        qn[4] = -ten + (2.0f * ten) * u01(r0[3]);
        qn[5] = -ten + (2.0f * ten) * u01(r1[0]);
        qn[0] = P.cart_min + P.cart_span * u01(r1[1]);
        ty = P.ty_min + P.ty_span * u01(r1[3]);
        tz = P.tz_min + P.tz_span * u01(r2[0]);
        depth = P.depth_min + P.depth_span * u01(r2[1]);
        pdepth = P.depth_min + P.depth_span * u01(r1[2]);
    }
    if (!(P.flags & VINE_FLAG_RANDOMIZE_DOF_INIT)) {
#pragma unroll
        for (int i = 0; i < ND; ++i) qn[i] = 0.0f;
    }
    if (!(P.flags & VINE_FLAG_RANDOMIZE_TARGETS)) { ty = P.ty_max; tz = P.tz_fixed; }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        ST(VF_Q0 + i) = qn[i];
        ST(VF_QD0 + i) = 0.0f;
        ST(VF_PREV_Q0 + i) = qn[i];
    }
    ST(VF_TARGET_Y) = ty;
    ST(VF_TARGET_Z) = tz;
    if (P.flags & VINE_FLAG_CREATE_SHELF) {
        ST(VF_SHELF_Y) = ty + (-0.2f + depth);
        ST(VF_SHELF_Z) = tz - 0.01f;
        ST(VF_OBJ_DEPTH) = depth;
    }
    if (P.flags & VINE_FLAG_CREATE_PIPE) {   // V5:841-885
        const float R = 0.0735f;             // PIPE_RADIUS = 0.07 * 1.05 (V5:88)
        const float ez = 1.0f - tz;
        const float deg = ((13199.0f * ez - 12276.0f) * ez + 4045.0f) * ez - 447.0f;
        const float tp = deg * 0.017453292519943295f;
        float stp, ctp;
        sincosf(tp, &stp, &ctp);
        ST(VF_PIPE_Y) = ty + pdepth * ctp + R * stp;
        ST(VF_PIPE_Z) = tz + pdepth * stp - R * ctp;
        ST(VF_OBJ_DEPTH) = pdepth;
        ST(VF_OBJ_ANGLE) = tp;
    }
}

// Forward kinematics of the tip body from absolute angles.
__device__ __forceinline__ void tip_fk_sc(const DevParams& P, float y, float vy, const float (&sn)[NL],
                                          const float (&cs)[NL], const float (&w)[NL], float (&tip)[4]) {
    float ty = y, tz = P.z1, tvy = vy, tvz = 0.0f;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const float sp = sn[k], cp = cs[k];      // already sin/cos of the world angle
        ty -= P.L * sp; tz += P.L * cp;
        tvy -= P.L * w[k] * cp; tvz -= P.L * w[k] * sp;
    }
    tip[0] = ty; tip[1] = tz; tip[2] = tvy; tip[3] = tvz;
}
__device__ __forceinline__ void tip_fk(const DevParams& P, float y, float vy, const float (&th)[NL],
                                       const float (&w)[NL], float (&tip)[4]) {
    float ty = y, tz = P.z1, tvy = vy, tvz = 0.0f;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        float s, c;
        sincosf(th[k], &s, &c);
        float sp = P.s0 * c + P.c0 * s, cp = P.c0 * c - P.s0 * s;
        ty -= P.L * sp; tz += P.L * cp;
        tvy -= P.L * w[k] * cp; tvz -= P.L * w[k] * sp;
    }
    tip[0] = ty; tip[1] = tz; tip[2] = tvy; tip[3] = tvz;
}

template <int OBS_TYPE, bool RANDOMIZE, int OBST>   // OBST bit 0: shelf, bit 1: pipe
__global__ __launch_bounds__(VINE_STEP_THREADS) void vine_step_kernel(const DevParams P, float* __restrict__ st,
                                                       const float* __restrict__ actions, float* __restrict__ obs,
                                                       float* __restrict__ rew, long long* __restrict__ reset,
                                                       long long* __restrict__ progress,
                                                       unsigned char* __restrict__ timeouts,
                                                       float* __restrict__ reward_matrix,
                                                       const float* __restrict__ reset_values,
                                                       unsigned long long* __restrict__ counters) {
    constexpr bool SHELF = (OBST & 1) != 0, PIPE = (OBST & 2) != 0, CONTACT = OBST != 0;
    const int n = P.n;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long step = step_of(P, counters);
    if (e < n) {
        // ---- VecTask.step: clamp actions (vec_task.py:333) ----
        const float2 act = reinterpret_cast<const float2*>(actions)[e];
        float a0 = clampf(act.x, P.clip_act), a1 = clampf(act.y, P.clip_act);
        // ---- pre_physics_step (V5:922-945) ----
        if (RANDOMIZE && P.act_noise != 0.0f) {
            unsigned r[4];
            float n0, n1;
            rng4(P, (unsigned)e, step, RNG_ACTION_NOISE, 0, r);
            normal2(r[0], r[1], n0, n1);
            a0 += P.act_noise * n0;
            a1 += P.act_noise * n1;
        }
        float new_rail = a0 * P.rail_scale;
        float new_fpam = (a1 + 1.0f) * 0.5f * P.fpam_span + P.fpam_min;   // /2 == *0.5 exactly
        float u_rail = new_rail, u_fpam = new_fpam;
        if (P.delay > 0) {
            int slot = (int)(step % (unsigned long long)P.delay);
            u_rail = ST(VF_FIFO0 + 2 * slot);
            u_fpam = ST(VF_FIFO0 + 2 * slot + 1);
            ST(VF_FIFO0 + 2 * slot) = new_rail;
            ST(VF_FIFO0 + 2 * slot + 1) = new_fpam;
        }
        if (P.flags & VINE_FLAG_FORCE_U_FPAM) u_fpam = 0.0f;
        if (P.flags & VINE_FLAG_FORCE_U_RAIL_VELOCITY) u_rail = 0.0f;
        float smoothed = ST(VF_SMOOTHED_U);
        {
            float alpha = (u_fpam > smoothed) ? P.alpha_inf : P.alpha_def;
            smoothed = alpha * smoothed + (1.0f - alpha) * u_fpam;
        }
        float q[ND], qd[ND], prev_q[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            q[i] = ST(VF_Q0 + i);
            qd[i] = ST(VF_QD0 + i);
            prev_q[i] = q[i];
        }
        float tip[4] = {ST(VF_TIP_Y), ST(VF_TIP_Z), 0.0f, 0.0f};   // (the velocities are recomputed before their first use)
        float prev_tip_y = tip[0], prev_tip_z = tip[1];
        float prev_u_rail = u_rail;
        float cart_y = ST(VF_CART_Y), cart_vy = ST(VF_CART_VY);
        float pcv = ST(VF_PREV_CART_VEL), pce = ST(VF_PREV_CART_VEL_ERR);
        float rail_force = 0.0f;
        float contact = SHELF ? ST(VF_CONTACT) : 0.0f, contact_sum = 0.0f;
        const float shelf_y = SHELF ? ST(VF_SHELF_Y) : 0.0f, shelf_z = SHELF ? ST(VF_SHELF_Z) : 0.0f;
        const float pipe_y = PIPE ? ST(VF_PIPE_Y) : 0.0f, pipe_z = PIPE ? ST(VF_PIPE_Z) : 0.0f;
        float pipe_ct = 1.0f, pipe_st = 0.0f;
        if (PIPE) sincosf(ST(VF_OBJ_ANGLE) + 1.5707963267948966f, &pipe_st, &pipe_ct);
        const float u_used = (P.flags & VINE_FLAG_USE_SMOOTHED_FPAM) ? smoothed : u_fpam;
        const bool held = (P.flags & VINE_FLAG_FPAM_DAMPING_HELD) != 0;
        // fields nothing in the step reads back (attributes / dashboard inputs of the reference) are stored on request only
        const bool introspect = (P.flags & VINE_FLAG_INTROSPECT) != 0;

        Dyn s;
        s.y = q[0];
        s.vy = qd[0];
        {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                a += q[k + 1];
                b += qd[k + 1];
                s.th[k] = a;
                s.w[k] = b;
            }
        }
        dyn_sync_trig(P, s);
        // ---- control_freq_inv x [refresh, actuation (V5:1028-1106), simulate] (vec_task.py:338-356) ----
        for (int it = 0; it < P.cfi; ++it) {
            float sc[20];
            if (RANDOMIZE && P.dyn_span != 0.0f) {
                // 20 factors from 3 Philox calls: 16-bit uniforms, two per 32-bit word
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    unsigned r[4];
                    rng4(P, (unsigned)e, step, RNG_DYN_SCALE, (unsigned)(it * 3 + g), r);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (g * 8 + k < 20) {
                            const float u = (float)((r[k >> 1] >> (16 * (k & 1))) & 0xffffu) * (1.0f / 65536.0f);
                            sc[g * 8 + k] = P.dyn_min + P.dyn_span * u;
                        }
                    }
                }
            } else if (RANDOMIZE) {
#pragma unroll
                for (int k = 0; k < 20; ++k) sc[k] = P.dyn_min;
            } else {
#pragma unroll
                for (int k = 0; k < 20; ++k) sc[k] = 1.0f;
            }
            float eff[ND], cj[ND], hc[ND];
            cj[0] = P.damping;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                float qj = (j == 0) ? s.th[0] : s.th[j] - s.th[j - 1];
                float qdj = (j == 0) ? s.w[0] : s.w[j] - s.w[j - 1];
                float t = P.K[j] * sc[j] * qj;
                float cv = P.C[j] * sc[5 + j];
                if (held) t += cv * qdj;
                t += P.bb[j] * sc[10 + j];
                t += P.B[j] * sc[15 + j] * u_used;
                eff[j + 1] = (P.eff_lim > 0.0f) ? clampf(-t, P.eff_lim) : -t;
                cj[j + 1] = P.damping + (held ? 0.0f : cv);
            }
            {
                float err = u_rail - cart_vy;
                float fmax = P.rail_acc * 0.5f;
                float minmax = (err > 0.0f) ? fmax : -fmax;
                float accel = (cart_vy - pcv) * P.inv_dt;
                float accel_target = (err > 0.0f) ? P.rail_acc : -P.rail_acc;
                minmax += 0.30f * (accel_target - accel);
                float pid = P.p_gain * err + P.d_gain * (err - pce);
                eff[0] = (fabsf(err) > 0.1f) ? minmax : pid;
                pce = err;
                pcv = cart_vy;
                rail_force = eff[0];
            }
#pragma unroll
            for (int i = 0; i < ND; ++i) hc[i] = P.hsub * cj[i];
            float qa[ND] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            float csum = 0.0f;
            if (SHELF) contact_sum += contact;            // vec_task.py:348-351: force left by the previous simulate
            const bool extras = (P.kq != 0.0f) || (P.cad != 0.0f);
            const bool implicit = (P.flags & VINE_FLAG_IMPLICIT_JOINT_DAMPING) != 0;
#define VINE_SUBSTEP_LOOP(IMPL, EXTR)                                                  \
    SimConst K;                                                                        \
    make_sim_const<IMPL, EXTR>(P, hc, K);                                              \
    for (int k = 0; k < P.substeps; ++k) {                                             \
        if (SHELF) csum += shelf_contact(P, s, shelf_y, shelf_z, qa);                  \
        if (PIPE && !SHELF) {                                                          \
            _Pragma("unroll") for (int i = 0; i < ND; ++i) qa[i] = 0.0f;               \
        }                                                                              \
        if (PIPE) pipe_contact(P, s, pipe_y, pipe_z, pipe_ct, pipe_st, qa);            \
        substep<IMPL, CONTACT, EXTR>(P, s, eff, cj, K, qa);                            \
    }
            if (implicit && !extras) { VINE_SUBSTEP_LOOP(true, false) }
            else if (implicit) { VINE_SUBSTEP_LOOP(true, true) }
            else if (!extras) { VINE_SUBSTEP_LOOP(false, false) }
            else { VINE_SUBSTEP_LOOP(false, true) }
#undef VINE_SUBSTEP_LOOP
            if (SHELF) contact = csum / (float)P.substeps;
            cart_y = s.y;
            cart_vy = s.vy;
        }
        // refreshed rigid-body states after the last simulate
        tip_fk_sc(P, s.y, s.vy, s.sn, s.cs, s.w, tip);
        q[0] = s.y;
        qd[0] = s.vy;
        q[1] = s.th[0];
        qd[1] = s.w[0];
#pragma unroll
        for (int k = 1; k < NL; ++k) {
            q[k + 1] = s.th[k] - s.th[k - 1];
            qd[k + 1] = s.w[k] - s.w[k - 1];
        }

        // ---- post_physics_step (V5:1110-1120) ----
        long long prog = progress[e] + 1;
        long long rst = reset[e];
        float agg = ST(VF_AGG_REW);
        float ty = ST(VF_TARGET_Y), tz = ST(VF_TARGET_Z);
        if (rst != 0) {  // reset requested by the previous step's reward pass (V5:1114-1116)
            float qn[ND];
            reset_env(P, st, n, e, step, reset_values, qn, ty, tz);
            rst = 0;
            prog = 0;
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                q[i] = qn[i];
                qd[i] = 0.0f;
                prev_q[i] = qn[i];
            }
            if (P.flags & VINE_FLAG_STALE_BODY_STATE_AFTER_RESET) {
                // tip/cart rigid-body states keep their pre-reset values (V5:796-797 TODO)
                prev_tip_y = tip[0];
                prev_tip_z = tip[1];
            } else {
                float th[NL], w[NL] = {0, 0, 0, 0, 0};
                float a = 0.0f;
#pragma unroll
                for (int k = 0; k < NL; ++k) { a += qn[k + 1]; th[k] = a; }
                tip_fk(P, qn[0], 0.0f, th, w, tip);
                prev_tip_y = tip[0];
                prev_tip_z = tip[1];
                cart_y = qn[0];
                cart_vy = 0.0f;
            }
            prev_u_rail = 0.0f;
            pce = 0.0f;
            agg = 0.0f;
        } else {
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                ST(VF_Q0 + i) = q[i];
                ST(VF_QD0 + i) = qd[i];
            }
            if (introspect) {
#pragma unroll
                for (int i = 0; i < ND; ++i) ST(VF_PREV_Q0 + i) = prev_q[i];
            }
        }
        const float obj_depth = ST(VF_OBJ_DEPTH), obj_angle = ST(VF_OBJ_ANGLE);

        // compute_observations (V5:1339-1390)
        // OBS_TYPE: the two scalable layouts are specialised; VINE_OBS_POS_ONLY has its own 14-column layout;
        // VINE_OBS_POS_AND_VEL stands for the 26-column family (V5:1357-1368), whose middle blocks are picked at
        // run time from P.obs_type (uniform branch).  Those four are unscaled by construction (V5:267-268).
        float o[VINE_MAX_OBS];
        int k = 0;
        constexpr int NOBS = (OBS_TYPE == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) ? 28
                           : (OBS_TYPE == VINE_OBS_TIP_AND_CART_AND_OBJ_INFO) ? 18
                           : (OBS_TYPE == VINE_OBS_POS_ONLY) ? 14 : 26;
        const float fd_tip_y = (tip[0] - prev_tip_y) * P.inv_cdt, fd_tip_z = (tip[1] - prev_tip_z) * P.inv_cdt;
        if (OBS_TYPE == VINE_OBS_POS_ONLY) {
#pragma unroll
            for (int i = 0; i < ND; ++i) o[k++] = q[i];
            o[k++] = 0.0f; o[k++] = tip[0]; o[k++] = tip[1];
            o[k++] = 0.0f; o[k++] = ty; o[k++] = tz;
            o[k++] = smoothed; o[k++] = prev_u_rail;
        } else if (OBS_TYPE == VINE_OBS_POS_AND_VEL) {
            const bool sim_vel = P.obs_type == VINE_OBS_POS_AND_VEL, fd_vel = P.obs_type == VINE_OBS_POS_AND_FD_VEL;
#pragma unroll
            for (int i = 0; i < ND; ++i) o[k++] = q[i];
#pragma unroll
            for (int i = 0; i < ND; ++i)
                o[k++] = sim_vel ? qd[i] : fd_vel ? (q[i] - prev_q[i]) * P.inv_cdt : prev_q[i];
            o[k++] = 0.0f; o[k++] = tip[0]; o[k++] = tip[1];
            o[k++] = 0.0f;
            o[k++] = sim_vel ? tip[2] : fd_vel ? fd_tip_y : prev_tip_y;
            o[k++] = sim_vel ? tip[3] : fd_vel ? fd_tip_z : prev_tip_z;
            o[k++] = 0.0f; o[k++] = ty; o[k++] = tz;
            o[k++] = 0.0f; o[k++] = 0.0f; o[k++] = 0.0f;
            o[k++] = smoothed; o[k++] = prev_u_rail;
        } else {
            if (OBS_TYPE == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) {
#pragma unroll
                for (int i = 0; i < ND; ++i) o[k++] = q[i];
#pragma unroll
                for (int i = 0; i < ND; ++i) o[k++] = (q[i] - prev_q[i]) * P.inv_cdt;
            } else {
                o[k++] = q[0];
                o[k++] = (q[0] - prev_q[0]) * P.inv_cdt;
            }
            o[k++] = 0.0f; o[k++] = tip[0]; o[k++] = tip[1];
            o[k++] = 0.0f; o[k++] = fd_tip_y; o[k++] = fd_tip_z;
            o[k++] = 0.0f; o[k++] = ty; o[k++] = tz;
            o[k++] = 0.0f; o[k++] = 0.0f; o[k++] = 0.0f;
            o[k++] = smoothed; o[k++] = prev_u_rail; o[k++] = obj_depth; o[k++] = obj_angle;
#pragma unroll
            for (int i = 0; i < NOBS; ++i) o[i] = o[i] * P.inv_obs_scale[i];
        }
        if (RANDOMIZE && P.obs_noise != 0.0f) {
#pragma unroll
            for (int i = 0; i < NOBS; i += 4) {
                unsigned r[4];
                float nn[4];
                rng4(P, (unsigned)e, step, RNG_OBS_NOISE, (unsigned)(i / 4), r);
                normal2(r[0], r[1], nn[0], nn[1]);
                normal2(r[2], r[3], nn[2], nn[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j < NOBS) o[i + j] += P.obs_noise * nn[j];
            }
        }
        // compute_reward (V5:1218-1331) + compute_reward_jit (V5:1470-1537)
        const float dy = tip[0] - ty, dz = tip[1] - tz;
        const float dist = sqrtf(dy * dy + dz * dz);
        const bool reached = dist < P.success_dist;
        const bool limit_hit = (cart_y > P.soft_limit) || (cart_y < -P.soft_limit);
        const bool tip_limit_hit = tip[0] < ty;
        const float cmean = SHELF ? contact_sum / (float)P.cfi : 0.0f;      // V5:1242-1248
        const float vnorm = sqrtf(tip[2] * tip[2] + tip[3] * tip[3]);
        float rm[VINE_NUM_REWARDS];
        rm[0] = -dist;
        rm[1] = -1.0f;
        rm[2] = reached ? 1000.0f : 0.0f;
        rm[3] = -(reached ? vnorm : 0.0f);
        rm[4] = vnorm;
        rm[5] = -fabsf(u_rail);
        rm[6] = -fabsf(u_fpam);
        rm[7] = -fabsf(u_rail - prev_u_rail);
        rm[8] = -fabsf(u_fpam - smoothed);
        rm[9] = limit_hit ? -100.0f : 0.0f;
        rm[10] = -fabsf(cart_y);
        rm[11] = tip_limit_hit ? -100.0f : 0.0f;
        rm[12] = -((cmean > 0.0f) ? cmean : 0.0f);
        float total = 0.0f;
#pragma unroll
        for (int i = 0; i < VINE_NUM_REWARDS; ++i) total += rm[i] * P.rw[i];
        agg += total;
        // compute_reset_jit (V5:1540-1558)
        if (prog >= (long long)P.max_len - 1) rst = 1;
        if (reached && (P.flags & VINE_FLAG_USE_TARGET_REACHED_RESET)) rst = 1;
        if (tip_limit_hit && (P.flags & VINE_FLAG_USE_TIP_LIMIT_HIT_RESET)) rst = 1;
        if (limit_hit) rst = 1;
        if (SHELF && cmean > 0.0f && (P.flags & VINE_FLAG_USE_NONZERO_CONTACT_FORCE_RESET)) rst = 1;
        // ---- VecTask.step epilogue (vec_task.py:366-380) ----
        const unsigned char to = (prog >= (long long)P.max_len - 1) && (rst != 0);
        float* orow = obs + (size_t)e * NOBS;
        if (NOBS % 4 == 0) {
#pragma unroll
            for (int i = 0; i < NOBS; i += 4)
                reinterpret_cast<float4*>(orow)[i / 4] = make_float4(clampf(o[i], P.clip_obs), clampf(o[i + 1], P.clip_obs),
                                                                     clampf(o[i + 2], P.clip_obs), clampf(o[i + 3], P.clip_obs));
        } else {
#pragma unroll
            for (int i = 0; i < NOBS; i += 2)
                reinterpret_cast<float2*>(orow)[i / 2] = make_float2(clampf(o[i], P.clip_obs), clampf(o[i + 1], P.clip_obs));
        }
        rew[e] = total;
        reset[e] = rst;
        progress[e] = prog;
        timeouts[e] = to;
        if (reward_matrix) {
#pragma unroll
            for (int i = 0; i < VINE_NUM_REWARDS; ++i) reward_matrix[(size_t)e * VINE_NUM_REWARDS + i] = rm[i];
        }
        // persistent state
        ST(VF_TIP_Y) = tip[0]; ST(VF_TIP_Z) = tip[1];
        ST(VF_CART_Y) = cart_y; ST(VF_CART_VY) = cart_vy;
        ST(VF_SMOOTHED_U) = smoothed;
        ST(VF_PREV_CART_VEL) = pcv; ST(VF_PREV_CART_VEL_ERR) = pce;
        ST(VF_AGG_REW) = agg;
        if (SHELF) ST(VF_CONTACT) = contact;
        if (introspect) {
            ST(VF_TIP_VY) = tip[2]; ST(VF_TIP_VZ) = tip[3];
            ST(VF_PREV_TIP_Y) = prev_tip_y; ST(VF_PREV_TIP_Z) = prev_tip_z;
            ST(VF_U_FPAM) = u_fpam; ST(VF_U_RAIL) = u_rail; ST(VF_PREV_U_RAIL) = prev_u_rail;
            ST(VF_RAIL_FORCE) = rail_force;
            if (SHELF) ST(VF_CONTACT_MEAN) = cmean;
        }
    }
    step_arrive(counters);
}

// ================================================================================================================
// Lane-cooperative step kernel: FOUR lanes per env (one DPP quad).  At the metric's 16384 envs per GPU the kernel above
// fills 256 of the chip's 1024 SIMDs with one wave each and sits on the single-wave VALU issue rate; spread over a
// quad the same envs are 1024 waves, each executing ~170 instead of ~275 instructions per substep.
//   * lane t of a quad owns link t (angle, rate, sin / cos of the world angle); link 4 (the 100 g end link) and the
//     cart are replicated on all four lanes;
//   * every cross-lane operand is a DPP quad_perm source (broadcast of lane k, rotation by +-1, xor 1 / 2 for the
//     quad sums): no LDS, no ds_bpermute;
//   * the 6x6 system is solved by eliminating the two replicated rows first (cart: constant pivot; link 4), which
//     leaves a 4x4 system with ONE ROW PER LANE, solved by Gauss-Jordan with the pivot row broadcast from lane j;
//   * the glue is parallelised where it is expensive: lane i draws the dynamics-scaling factors of control iteration i
//     (3 Philox calls instead of 12), lane t assembles, perturbs and stores observation columns [4t, 4t+4) and
//     [16+4t, 16+4t+4) (2 Philox calls + 4 Box-Muller pairs instead of 7 + 14), lanes 0..2 draw one reset word each.
// Same arithmetic model, same RNG keys and draw-to-variable mapping as vine_step_kernel (and the oracle): results
// differ by round-off only (another elimination order).  Covers the configurations the rollout actually runs at that
// size -- no obstacle, implicit joint damping, no joint stiffness / link damping, the two scalable observation
// layouts; everything else takes the one-lane kernel.
template <int CTRL>
__device__ __forceinline__ float qperm(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int K>
__device__ __forceinline__ float qbcast(float v) { return qperm<K * 0x55>(v); }            // lane K of the quad
__device__ __forceinline__ float qprev(float v) { return qperm<0x93>(v); }                 // lane t reads lane t - 1
__device__ __forceinline__ float qnext(float v) { return qperm<0x39>(v); }                 // lane t reads lane t + 1
__device__ __forceinline__ float quad_sum(float v) {
    const float s = v + qperm<0xB1>(v);                                                    // xor 1
    return s + qperm<0x4E>(s);                                                             // xor 2
}
// NOTE on selects: a DPP read must execute with the whole quad active -- a source lane that EXEC has switched off
// delivers 0 -- and `c ? dpp(x) : y` is a branch around the DPP (C++ evaluates only the chosen operand, and the
// compiler may not speculate a convergent operation).  Every cross-lane value is therefore produced unconditionally
// and selected afterwards: pick() / sel4() take their operands by value.
__device__ __forceinline__ float pick(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ float sel4(int t, float v0, float v1, float v2, float v3) {
    return t == 0 ? v0 : (t == 1 ? v1 : (t == 2 ? v2 : v3));
}
// a wave-uniform value (a kernel-argument word) pinned into a scalar register HERE: left alone, the compiler sinks the
// argument load into the one arm of a select that uses it, and sel4(t, P.x[0], ..) becomes four branches with a scalar
// load and a wait each
__device__ __forceinline__ float pin_s(float x) {
    asm volatile("" : "+s"(x));
    return x;
}
template <int K>
__device__ __forceinline__ unsigned qbcast_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, true);
}

// acc += (value of `src` on lane K of the quad) * b as ONE instruction, v_fmac_f32_dpp.  The compiler folds a DPP read into
// v_mul / v_add / v_sub consumers by itself but not into v_fmac (it leaves a v_mov_b32_dpp in front of it: 30 of the 178
// instructions of a substep).  A DPP read of a VGPR needs two wait states after the VALU instruction that wrote it, which the
// compiler cannot see inside an asm statement: NOP = true puts an `s_nop 1` in front (for operands that may have been
// produced by the instruction just before); scripts/check_dpp_hazards.py scans the generated assembly of every
// instantiation for an unprotected read (run it after touching this kernel).
template <int K, bool NOP>
__device__ __forceinline__ void fmac_bcast(float& acc, float src, float b) {
    static_assert(K >= 0 && K < 4, "lane of the quad");
    if (K == 0) {
        if (NOP) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(b));
        else asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(b));
    } else if (K == 1) {
        if (NOP) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(b));
        else asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(b));
    } else if (K == 2) {
        if (NOP) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(b));
        else asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(b));
    } else {
        if (NOP) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(b));
        else asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(b));
    }
}

// inclusive prefix sum over the quad (lane t: v_0 + ... + v_t): two DPP steps
__device__ __forceinline__ float quad_scan_incl(float v, int t) {
    const float s = v + pick(t > 0, qprev(v), 0.0f);
    return s + pick(t >= 2, qperm<0x4E>(s), 0.0f);
}

#ifndef VSQ_PIPE_COOP_FROM
#define VSQ_PIPE_COOP_FROM 2      // first link whose pipe contacts are split over the quad (A/B: 1, 2, 3; see §4.1c of DESIGN.md)
#endif
#ifndef VSQ_SHELF_COOP_FROM
#define VSQ_SHELF_COOP_FROM 4     // first link whose shelf contacts are split over the quad (A/B: 3, 4)
#endif
#ifdef VSQ_TIMING
__device__ unsigned long long vsq_t[1024 * 8];      // (debug build: 8 time stamps per wave, scripts/ubench/step_phases.py)
#endif
// ROLL (round 5, vine_step_rollout): the rollout step's policy head in front of the step and its bookkeeping behind it, in
// this launch -- see include/vine.h.  The plain step (ROLL = false) is a separate instantiation and is not touched.
struct RollArgs {
    const float* y; const float* hw; const float* hc; const float* logstd;
    const double* vmean; const double* vvar;
    float ln_eps, veps;
    unsigned seed_lo, seed_hi; const long long* counter;
    float* mu_out; float* sigma_out; float* value_out; float* action_out; float* neglogp_out;
    float shift, scale, gamma_b;
    float* shaped; unsigned char* dones; float* cur_r; float* cur_l;
    float* h_state; float* c_state; float* h_op; long long h_op_stride;
    float* partial;
};
template <int OBS_TYPE, bool RANDOMIZE, int OBST, bool ROLL = false>   // OBST bit 0: shelf, bit 1: pipe
__global__ __launch_bounds__(256) void vine_step_quad_kernel(const DevParams P, float* __restrict__ st,
                                                             const float* __restrict__ actions, float* __restrict__ obs,
                                                             float* __restrict__ rew, long long* __restrict__ reset,
                                                             long long* __restrict__ progress,
                                                             unsigned char* __restrict__ timeouts,
                                                             float* __restrict__ reward_matrix,
                                                             const float* __restrict__ reset_values,
                                                             unsigned long long* __restrict__ counters, const RollArgs R) {
    static_assert(OBS_TYPE == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO || OBS_TYPE == VINE_OBS_TIP_AND_CART_AND_OBJ_INFO, "");
    constexpr int NOBS = OBS_TYPE == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO ? 28 : 18;
    constexpr bool SHELF = (OBST & 1) != 0, PIPE = (OBST & 2) != 0, CONTACT = OBST != 0;
    const int n = P.n;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = gid >> 2, t = threadIdx.x & 3;
#ifdef VSQ_TIMING
    if ((threadIdx.x & 63) == 0) vsq_t[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 0] = wall_clock64();
#endif
    const unsigned long long step = step_of(P, counters);
    float roll_sr = 0.0f, roll_sl = 0.0f, roll_cnt = 0.0f;      // (ROLL) this lane's finished-episode totals
    if (e < n) {
        // ---- everything the step needs from memory is requested FIRST, in one batch, and waited for once (behind the
        // pure-ALU random-number work below).  The prologue used to be four dependent stages -- per-lane constants fetched
        // one kernel-argument word at a time behind branches (the compiler sinks an unpinned P.x[i] into the arm of the
        // select that needs it: ~45 scalar loads, each with its own wait), the action load, the action-noise Philox, then
        // the state loads, with the body-state fields behind `if (progress == 0)` -- and took 6 of the kernel's 24 us.
        float2 act = make_float2(0.0f, 0.0f);
        float4 yq[ROLL ? 16 : 1];      // ROLL: this lane's 64 units {16 i + 4 t .. + 3 : i < 16} of the env's LSTM output row
        float roll_cr = 0.0f, roll_cl = 0.0f;
        if (ROLL) {
            // (interleaved: the four lanes of a quad read 64 contiguous bytes per load -- 16 lines per wave instruction; with
            // 64 consecutive units per lane every lane of the wave sat on its own line)
            const float4* yr = reinterpret_cast<const float4*>(R.y + (size_t)e * 256) + t;
#pragma unroll
            for (int i = 0; i < 16; ++i) yq[i] = yr[4 * i];
            roll_cr = R.cur_r[e]; roll_cl = R.cur_l[e];
        } else {
            act = reinterpret_cast<const float2*>(actions)[e];
        }
        float smoothed = ST(VF_SMOOTHED_U);
        // relative joint coordinates: lane t holds joint t (dof 1 + t), everyone dof 5 and the cart
        float q_own = ST(VF_Q0 + 1 + t), qd_own = ST(VF_QD0 + 1 + t);
        float q5 = ST(VF_Q0 + 5), qd5 = ST(VF_QD0 + 5);
        float y = ST(VF_Q0), vy = ST(VF_QD0);
        const long long prog_in = progress[e];
        long long rst = reset[e];
        // Rigid-body states of the tip and the cart at the start of the step.  They equal the forward kinematics of the
        // DOF state EXCEPT in the step after a reset (P5: reset_idx does not move bodies, so they are stale), and they are
        // then the only copy; progress == 0 marks exactly those envs (and freshly initialised ones, whose body states
        // vine_init / vine_reset_idx stored).  Everyone else re-derives them (below, once sin / cos exist) and, at the end,
        // stores them only when the env was reset in this step.  (Loaded unconditionally -- 16 B per env and step -- so
        // that they do not wait for progress[e]; with introspection on they are stored every step, as attribute views.)
        const float m_tip_y = ST(VF_TIP_Y), m_tip_z = ST(VF_TIP_Z), m_cart_y = ST(VF_CART_Y), m_cart_vy = ST(VF_CART_VY);
        float pcv = ST(VF_PREV_CART_VEL), pce = ST(VF_PREV_CART_VEL_ERR);
        float agg = ST(VF_AGG_REW);
        float ty = ST(VF_TARGET_Y), tz = ST(VF_TARGET_Z);
        float obj_depth = ST(VF_OBJ_DEPTH), obj_angle = ST(VF_OBJ_ANGLE);
        // obstacles (replicated on the quad): the shelf's force on its front strip as the last simulate left it, the
        // obstacle poses of THIS step (a reset below moves them for the next one)
        float contact = SHELF ? ST(VF_CONTACT) : 0.0f, contact_sum = 0.0f;
        const float shelf_y = SHELF ? ST(VF_SHELF_Y) : 0.0f, shelf_z = SHELF ? ST(VF_SHELF_Z) : 0.0f;
        const float m_pipe_y = PIPE ? ST(VF_PIPE_Y) : 0.0f, m_pipe_z = PIPE ? ST(VF_PIPE_Z) : 0.0f;
        float fifo_rail = 0.0f, fifo_fpam = 0.0f;
        int fifo_slot = 0;
        if (P.delay > 0) {
            fifo_slot = (int)(step % (unsigned long long)P.delay);
            fifo_rail = ST(VF_FIFO0 + 2 * fifo_slot);
            fifo_fpam = ST(VF_FIFO0 + 2 * fifo_slot + 1);
        }
#ifdef VSQ_TIMING
        if ((threadIdx.x & 63) == 0) vsq_t[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 1] = wall_clock64();
#endif
        // ---- per-lane constants of link t / joint t.  The 44 kernel-argument words they are selected from are pinned into
        // scalar registers by TWO asm statements (24 + 20 operands): the loads of each group are issued together and waited
        // for once.  Unpinned, the compiler sinks each P.x[i] into the arm of the select that uses it (45 scalar loads, a
        // wait each: 1.1 us); pinned one select at a time it still waited eleven times.
        float cb[4] = {P.b[0], P.b[1], P.b[2], P.b[3]}, cgb[4] = {P.gb[0], P.gb[1], P.gb[2], P.gb[3]};
        float ca[4][4] = {{P.a[0][0], P.a[1][0], P.a[2][0], P.a[3][0]}, {P.a[0][1], P.a[1][1], P.a[2][1], P.a[3][1]},
                          {P.a[0][2], P.a[1][2], P.a[2][2], P.a[3][2]}, {P.a[0][3], P.a[1][3], P.a[2][3], P.a[3][3]}};
        float ca4[4] = {P.a[0][4], P.a[1][4], P.a[2][4], P.a[3][4]};
        float cK[4] = {P.K[0], P.K[1], P.K[2], P.K[3]}, cC[4] = {P.C[0], P.C[1], P.C[2], P.C[3]};
        float cbb[4] = {P.bb[0], P.bb[1], P.bb[2], P.bb[3]}, cB[4] = {P.B[0], P.B[1], P.B[2], P.B[3]};
        asm volatile("" : "+s"(cb[0]), "+s"(cb[1]), "+s"(cb[2]), "+s"(cb[3]), "+s"(cgb[0]), "+s"(cgb[1]), "+s"(cgb[2]), "+s"(cgb[3]),
                          "+s"(ca[0][0]), "+s"(ca[0][1]), "+s"(ca[0][2]), "+s"(ca[0][3]), "+s"(ca[1][0]), "+s"(ca[1][1]),
                          "+s"(ca[1][2]), "+s"(ca[1][3]), "+s"(ca[2][0]), "+s"(ca[2][1]), "+s"(ca[2][2]), "+s"(ca[2][3]),
                          "+s"(ca[3][0]), "+s"(ca[3][1]), "+s"(ca[3][2]), "+s"(ca[3][3]));
        asm volatile("" : "+s"(ca4[0]), "+s"(ca4[1]), "+s"(ca4[2]), "+s"(ca4[3]), "+s"(cK[0]), "+s"(cK[1]), "+s"(cK[2]), "+s"(cK[3]),
                          "+s"(cC[0]), "+s"(cC[1]), "+s"(cC[2]), "+s"(cC[3]), "+s"(cbb[0]), "+s"(cbb[1]), "+s"(cbb[2]), "+s"(cbb[3]),
                          "+s"(cB[0]), "+s"(cB[1]), "+s"(cB[2]), "+s"(cB[3]));
        const float b_t = sel4(t, cb[0], cb[1], cb[2], cb[3]);
        const float gb_t = sel4(t, cgb[0], cgb[1], cgb[2], cgb[3]);
        const float nb_t = -b_t;
        float a_k[4], as_k[4];                           // a_tk (cos terms; a_tt on the diagonal), the same with 0 on the diagonal
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a_k[k] = sel4(t, ca[k][0], ca[k][1], ca[k][2], ca[k][3]);
            as_k[k] = t == k ? 0.0f : a_k[k];
        }
        const float a_t4 = sel4(t, ca4[0], ca4[1], ca4[2], ca4[3]);
        const float K_t = sel4(t, cK[0], cK[1], cK[2], cK[3]), C_t = sel4(t, cC[0], cC[1], cC[2], cC[3]);
        const float bb_t = sel4(t, cbb[0], cbb[1], cbb[2], cbb[3]), B_t = sel4(t, cB[0], cB[1], cB[2], cB[3]);
        // ---- random numbers of the step that do not depend on its state (pure ALU: the loads above are in flight)
        float an0 = 0.0f, an1 = 0.0f;
        if (RANDOMIZE && P.act_noise != 0.0f) {
            unsigned r[4];
            rng4(P, (unsigned)e, step, RNG_ACTION_NOISE, 0, r);
            normal2(r[0], r[1], an0, an1);
        }
        // observation noise of this lane's 8 columns (it does not depend on the state either: drawn here, under the loads,
        // instead of as 2 Philox calls + 4 Box-Muller pairs at the end of the kernel, where nothing covers them)
        float onoise[8];
        if (RANDOMIZE && P.obs_noise != 0.0f) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // Philox call index = first column / 4: t for columns [4t, 4t+4), 4 + t for [16+4t, ...)
                unsigned r[4];
                rng4(P, (unsigned)e, step, RNG_OBS_NOISE, (unsigned)(4 * half + t), r);
                normal2(r[0], r[1], onoise[4 * half], onoise[4 * half + 1]);
                normal2(r[2], r[3], onoise[4 * half + 2], onoise[4 * half + 3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) onoise[j] = 0.0f;
        }
        // dynamics-scaling factors: lane i draws the 20 factors of control iteration i (V5:1053-1055)
        float scl[20];
        if (RANDOMIZE && P.dyn_span != 0.0f) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                unsigned r[4];
                rng4(P, (unsigned)e, step, RNG_DYN_SCALE, (unsigned)(t * 3 + g), r);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (g * 8 + k < 20) {
                        const float u = (float)((r[k >> 1] >> (16 * (k & 1))) & 0xffffu) * (1.0f / 65536.0f);
                        scl[g * 8 + k] = P.dyn_min + P.dyn_span * u;
                    }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 20; ++k) scl[k] = RANDOMIZE ? P.dyn_min : 1.0f;
        }
#ifdef VSQ_TIMING
        if ((threadIdx.x & 63) == 0) vsq_t[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 3] = wall_clock64();
#endif
        float roll_value = 0.0f;
        if (ROLL) {
            // ---- policy head (vine_policy_head's formulas, ppo_kernels.hip): LayerNorm of the row (two passes: mean, then
            // centred second moment and the three centred dot products with gamma_u w_k[u]), mu / value, sampling
            float s1 = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s1 += (yq[i].x + yq[i].y) + (yq[i].z + yq[i].w);
            const float mean = quad_sum(s1) * (1.0f / 256.0f);
            float q2 = 0.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
            const float4* hw0 = reinterpret_cast<const float4*>(R.hw) + t;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float4 w0 = hw0[4 * i], w1 = hw0[64 + 4 * i], w2 = hw0[128 + 4 * i];
                const float c0 = yq[i].x - mean, c1 = yq[i].y - mean, c2 = yq[i].z - mean, c3 = yq[i].w - mean;
                q2 += (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3);
                d0 += (c0 * w0.x + c1 * w0.y) + (c2 * w0.z + c3 * w0.w);
                d1 += (c0 * w1.x + c1 * w1.y) + (c2 * w1.z + c3 * w1.w);
                d2 += (c0 * w2.x + c1 * w2.y) + (c2 * w2.z + c3 * w2.w);
            }
            const float rstd = rsqrtf(quad_sum(q2) * (1.0f / 256.0f) + R.ln_eps);
            const float m0 = rstd * quad_sum(d0) + R.hc[0], m1 = rstd * quad_sum(d1) + R.hc[1];
            float v = rstd * quad_sum(d2) + R.hc[2];
            if (R.vmean) {      // RunningMeanStd's own float64 statistics: mean.float(), sqrt(var.float() + eps)
                const float vm = (float)R.vmean[0], vs = sqrtf((float)R.vvar[0] + R.veps);
                v = fminf(fmaxf(v, -5.0f), 5.0f) * vs + vm;
            }
            roll_value = v;
            unsigned r[4];
            philox4x32_10((unsigned)e, (unsigned)(unsigned long long)R.counter[0], 0x504f4c59u, 0u, R.seed_lo, R.seed_hi, r);
            const float u1 = 1.0f - (float)(r[0] >> 8) * (1.0f / 16777216.0f);
            const float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.0f * __logf(u1));
            float sn_, cs_;
            __sincosf(6.283185307179586f * u2, &sn_, &cs_);
            const float e0 = rad * cs_, e1 = rad * sn_;
            const float ls0 = R.logstd[0], ls1 = R.logstd[1], sg0 = __expf(ls0), sg1 = __expf(ls1);
            act = make_float2(m0 + sg0 * e0, m1 + sg1 * e1);
            if (t == 0) {
                float nlp = 0.9189385332046727f * 2.0f;
                nlp += 0.5f * e0 * e0 + ls0;
                nlp += 0.5f * e1 * e1 + ls1;
                reinterpret_cast<float2*>(R.mu_out)[e] = make_float2(m0, m1);
                reinterpret_cast<float2*>(R.sigma_out)[e] = make_float2(sg0, sg1);
                reinterpret_cast<float2*>(R.action_out)[e] = act;
                R.value_out[e] = v;
                R.neglogp_out[e] = nlp;
            }
        }
        // ---- VecTask.step: clamp actions (vec_task.py:333); pre_physics_step (V5:922-945), replicated on the quad
        float a0 = clampf(act.x, P.clip_act), a1 = clampf(act.y, P.clip_act);
        if (RANDOMIZE && P.act_noise != 0.0f) {
            a0 += P.act_noise * an0;
            a1 += P.act_noise * an1;
        }
        const float new_rail = a0 * P.rail_scale;
        const float new_fpam = (a1 + 1.0f) * 0.5f * P.fpam_span + P.fpam_min;
        float u_rail = new_rail, u_fpam = new_fpam;
        if (P.delay > 0) {
            u_rail = fifo_rail;
            u_fpam = fifo_fpam;
            if (t == 0) {
                ST(VF_FIFO0 + 2 * fifo_slot) = new_rail;
                ST(VF_FIFO0 + 2 * fifo_slot + 1) = new_fpam;
            }
        }
        if (P.flags & VINE_FLAG_FORCE_U_FPAM) u_fpam = 0.0f;
        if (P.flags & VINE_FLAG_FORCE_U_RAIL_VELOCITY) u_rail = 0.0f;
        {
            const float alpha = (u_fpam > smoothed) ? P.alpha_inf : P.alpha_def;
            smoothed = alpha * smoothed + (1.0f - alpha) * u_fpam;
        }
        const float prev_q_own = q_own, prev_q5 = q5, prev_y = y;
        const bool introspect = (P.flags & VINE_FLAG_INTROSPECT) != 0;
        const bool body_from_mem = introspect || prog_in == 0;
        float tip_y = 0.0f, tip_z = 0.0f, tip_vy = 0.0f, tip_vz = 0.0f;
        float cart_y = y, cart_vy = vy;
        if (body_from_mem) {
            tip_y = m_tip_y; tip_z = m_tip_z;
            cart_y = m_cart_y; cart_vy = m_cart_vy;
        }
        float prev_u_rail = u_rail;
        float rail_force = 0.0f;
        PipePose pipeT = PipePose{0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f};
        if (PIPE) {
            float pst, pct;
            sincosf(obj_angle + 1.5707963267948966f, &pst, &pct);
            pipeT = pipe_pose(m_pipe_y, m_pipe_z, pct, pst);
        }
        const float u_used = (P.flags & VINE_FLAG_USE_SMOOTHED_FPAM) ? smoothed : u_fpam;
        const bool held = (P.flags & VINE_FLAG_FPAM_DAMPING_HELD) != 0;
        // absolute angles / rates: inclusive prefix sums over the quad, then link 4 on top of lane 3's
        float th, w;
        {
            float s = q_own + pick(t > 0, qprev(q_own), 0.0f);
            th = s + pick(t >= 2, qperm<0x4E>(s), 0.0f);
            float v = qd_own + pick(t > 0, qprev(qd_own), 0.0f);
            w = v + pick(t >= 2, qperm<0x4E>(v), 0.0f);
        }
        float th4 = qbcast<3>(th) + q5, w4 = qbcast<3>(w) + qd5;
        float sn, cs, sn4, cs4;
        {
            float s_, c_;
            sincosf(th, &s_, &c_);
            sn = P.s0 * c_ + P.c0 * s_; cs = P.c0 * c_ - P.s0 * s_;
            sincosf(th4, &s_, &c_);
            sn4 = P.s0 * c_ + P.c0 * s_; cs4 = P.c0 * c_ - P.s0 * s_;
        }
        {   // forward kinematics of the start pose (quad sums outside any branch), used unless the body states are stale
            const float fk_y = y - P.L * (quad_sum(sn) + sn4), fk_z = P.z1 + P.L * (quad_sum(cs) + cs4);
            tip_y = body_from_mem ? tip_y : fk_y;
            tip_z = body_from_mem ? tip_z : fk_z;
        }
        float prev_tip_y = tip_y, prev_tip_z = tip_z;
#ifdef VSQ_TIMING
        if (q_own + y + smoothed + pcv + agg + ty + obj_depth + m_tip_y + (float)rst + (float)prog_in == 1234.5f) rew[e] = 1.0f;   // (all loads back)
        if ((threadIdx.x & 63) == 0) vsq_t[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 4] = wall_clock64();
#endif
        const float h = P.hsub;
        const float gb4 = P.gb[4], b4 = P.b[4], a44 = P.a[4][4];
        // The cart row has a constant pivot (a00 = total mass + h * DOF damping), so its elimination is folded into the
        // CONSTANTS of the remaining 5x5 system: row t, column k becomes (a_tk - b_t b_k / a00) cos cos + a_tk sin sin.
        const float a00 = P.mtot + h * P.damping, inv_a00 = 1.0f / a00;
        float al_k[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) al_k[k] = a_k[k] - b_t * (P.b[k] * inv_a00);
        const float al_4 = a_t4 - b_t * (b4 * inv_a00);
        const float nm_t = t > 0 ? -1.0f : 0.0f;
        const float be_t = b_t * inv_a00, be4 = b4 * inv_a00, b4sq = b4 * b4 * inv_a00;
        // ---- control_freq_inv x [actuation (V5:1028-1106), simulate]: iterations unrolled by 4 (the broadcasting lane
        // of the scaling factors is a compile-time constant), any further ones reuse the pattern
#define VINE_QUAD_ITER(IT)                                                                                               \
        {                                                                                                                \
            float sK, sC, sb, sB, sK4, sC4, sb4, sB4;                                                                    \
            if (RANDOMIZE && P.dyn_span != 0.0f) {                                                                       \
                sK = sel4(t, qbcast<IT>(scl[0]), qbcast<IT>(scl[1]), qbcast<IT>(scl[2]), qbcast<IT>(scl[3]));            \
                sC = sel4(t, qbcast<IT>(scl[5]), qbcast<IT>(scl[6]), qbcast<IT>(scl[7]), qbcast<IT>(scl[8]));            \
                sb = sel4(t, qbcast<IT>(scl[10]), qbcast<IT>(scl[11]), qbcast<IT>(scl[12]), qbcast<IT>(scl[13]));        \
                sB = sel4(t, qbcast<IT>(scl[15]), qbcast<IT>(scl[16]), qbcast<IT>(scl[17]), qbcast<IT>(scl[18]));        \
                sK4 = qbcast<IT>(scl[4]); sC4 = qbcast<IT>(scl[9]); sb4 = qbcast<IT>(scl[14]); sB4 = qbcast<IT>(scl[19]); \
            } else {                                                                                                     \
                sK = sC = sb = sB = sK4 = sC4 = sb4 = sB4 = scl[0];                                                      \
            }                                                                                                            \
            quad_simulate(sK, sC, sb, sB, sK4, sC4, sb4, sB4);                                                           \
        }
        auto quad_simulate = [&](float sK, float sC, float sb, float sB, float sK4, float sC4, float sb4, float sB4) {
            // FPAM torque model of joint t (own) and joint 4 (replicated)
            const float qj = th - pick(t > 0, qprev(th), 0.0f), qdj = w - pick(t > 0, qprev(w), 0.0f);
            float tq = K_t * sK * qj;
            const float cv = C_t * sC;
            if (held) tq += cv * qdj;
            tq += bb_t * sb;
            tq += B_t * sB * u_used;
            const float eff_t = (P.eff_lim > 0.0f) ? clampf(-tq, P.eff_lim) : -tq, cj_t = P.damping + (held ? 0.0f : cv);
            const float q5r = th4 - qbcast<3>(th), qd5r = w4 - qbcast<3>(w);
            float tq4 = P.K[4] * sK4 * q5r;
            const float cv4 = P.C[4] * sC4;
            if (held) tq4 += cv4 * qd5r;
            tq4 += P.bb[4] * sb4;
            tq4 += P.B[4] * sB4 * u_used;
            const float eff4 = (P.eff_lim > 0.0f) ? clampf(-tq4, P.eff_lim) : -tq4, cj4 = P.damping + (held ? 0.0f : cv4);
            float eff0;
            {   // rail controller (V5:1069-1098), replicated
                const float err = u_rail - cart_vy;
                const float fmax = P.rail_acc * 0.5f;
                float minmax = (err > 0.0f) ? fmax : -fmax;
                const float accel = (cart_vy - pcv) * P.inv_dt;
                const float accel_target = (err > 0.0f) ? P.rail_acc : -P.rail_acc;
                minmax += 0.30f * (accel_target - accel);
                const float pid = P.p_gain * err + P.d_gain * (err - pce);
                eff0 = (fabsf(err) > 0.1f) ? minmax : pid;
                pce = err;
                pcv = cart_vy;
                rail_force = eff0;
            }
            // constants of this simulate: implicit damping folded into the matrix (tridiagonal in the absolute angles)
            const float cj0 = P.damping;
            const float hc_t = h * cj_t, hc4 = h * cj4, hc0 = h * cj0;
            const float hc_n = pick(t == 3, hc4, qnext(hc_t));                 // damping of the joint BEYOND link t
            float nbase[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) nbase[k] = t == k ? hc_t + hc_n : (t == k + 1 ? -hc_t : (t == k - 1 ? -hc_n : 0.0f));
            const float nb4 = t == 3 ? -hc4 : 0.0f;
            const float adiag4 = a44 + hc4;
            (void)hc0;
            if (SHELF) contact_sum += contact;            // vec_task.py:348-351: force left by the previous simulate
            float csum = 0.0f;
            for (int ss = 0; ss < P.substeps; ++ss) {
                // ---- obstacle contacts: lane t evaluates link t (broad phase + narrow phase of shelf_link_contact /
                // pipe_link_contact, the same code as the one-lane kernel), link 4 is evaluated on every lane; joint
                // positions / velocities are prefix sums over the quad, the forces on the links beyond link t a suffix sum
                float Qt = 0.0f, Q4 = 0.0f, Qc = 0.0f;
                if (CONTACT) {
                    const float L = P.L;
                    const float dyo = -sn, dzo = cs, vyo = -(w * cs), vzo = -(w * sn);
                    const float iy = quad_scan_incl(dyo, t), iz = quad_scan_incl(dzo, t);
                    const float ivy = quad_scan_incl(vyo, t), ivz = quad_scan_incl(vzo, t);
                    const float py = y + L * (iy - dyo), pz = P.z1 + L * (iz - dzo);
                    const float pvy = vy + L * (ivy - vyo), pvz = L * (ivz - vzo);
                    const float p4y = y + L * qbcast<3>(iy), p4z = P.z1 + L * qbcast<3>(iz);
                    const float pv4y = vy + L * qbcast<3>(ivy), pv4z = L * qbcast<3>(ivz);
                    const float z0 = t == 0 ? -0.00575f : 0.0f, z1 = t == 0 ? 0.09425f : L;
                    float fy = 0.0f, fz = 0.0f, mom = 0.0f, f4y = 0.0f, f4z = 0.0f, mom4 = 0.0f;
                    float sfy = 0.0f, sfz = 0.0f, s4y = 0.0f, s4z = 0.0f;
                    // proximal links: by their own lanes (rarely anywhere near the obstacle); distal links (pipe: 3 and 4,
                    // shelf: 4): their points and the obstacle's corners split over the quad (*_link_contact_coop), the
                    // partial sums added below
                    const float p3y = qbcast<3>(py), p3z = qbcast<3>(pz), pv3y = qbcast<3>(pvy), pv3z = qbcast<3>(pvz);
                    const float sn3 = qbcast<3>(sn), cs3 = qbcast<3>(cs), w3 = qbcast<3>(w);
                    float c3y = 0.0f, c3z = 0.0f, cm3 = 0.0f, c4y = 0.0f, c4z = 0.0f, cm4 = 0.0f;
                    // (measured at 16384 envs: the split pays for the pipe -- a reaching vine has links 3 and 4 INSIDE the tube:
                    // default-config training rollout 5.06 -> 4.66 ms -- while the shelf is touched by link 4 almost alone:
                    // there link 3 stays with its own lane (random policy 85 us; with link 3 split as well 95 us))
                    float cs3y = 0.0f, cs3z = 0.0f, cms3 = 0.0f, s3y = 0.0f, s3z = 0.0f;      // link 3's shelf share (cooperative form)
                    if (SHELF) {
#if VSQ_SHELF_COOP_FROM <= 3
                        if (t < 3) shelf_link_contact(P, z0, z1, py, pz, pvy, pvz, sn, cs, w, shelf_y, shelf_z, fy, fz, mom, sfy, sfz);
                        shelf_link_contact_coop(P, t, 0.0f, L, p3y, p3z, pv3y, pv3z, sn3, cs3, w3, shelf_y, shelf_z, cs3y, cs3z, cms3, s3y, s3z);
#else
                        shelf_link_contact(P, z0, z1, py, pz, pvy, pvz, sn, cs, w, shelf_y, shelf_z, fy, fz, mom, sfy, sfz);
#endif
                        shelf_link_contact_coop(P, t, 0.0f, L, p4y, p4z, pv4y, pv4z, sn4, cs4, w4, shelf_y, shelf_z, c4y, c4z, cm4, s4y, s4z);
                    }
                    float c2y = 0.0f, c2z = 0.0f, cm2 = 0.0f, c1y = 0.0f, c1z = 0.0f, cm1 = 0.0f;
                    if (PIPE) {
                        // broad phase: link t's on lane t (z0 = 0, z1 = L for t >= 1), link 4's on every lane; two bits per
                        // link, broadcast over the quad to the lanes that share a cooperative link's narrow phase
                        const unsigned nb_own = pipe_broad_phase(z0, z1, py, pz, sn, cs, pipeT);
                        const unsigned nb4 = pipe_broad_phase(0.0f, L, p4y, p4z, sn4, cs4, pipeT);
                        if (t < VSQ_PIPE_COOP_FROM) pipe_link_contact(P, z0, z1, py, pz, pvy, pvz, sn, cs, w, pipeT, fy, fz, mom, nb_own);
#if VSQ_PIPE_COOP_FROM <= 1
                        pipe_link_contact_coop(P, t, 0.0f, L, qbcast<1>(py), qbcast<1>(pz), qbcast<1>(pvy), qbcast<1>(pvz),
                                               qbcast<1>(sn), qbcast<1>(cs), qbcast<1>(w), pipeT, c1y, c1z, cm1, qbcast_u<1>(nb_own));
#endif
#if VSQ_PIPE_COOP_FROM <= 2
                        pipe_link_contact_coop(P, t, 0.0f, L, qbcast<2>(py), qbcast<2>(pz), qbcast<2>(pvy), qbcast<2>(pvz),
                                               qbcast<2>(sn), qbcast<2>(cs), qbcast<2>(w), pipeT, c2y, c2z, cm2, qbcast_u<2>(nb_own));
#endif
                        pipe_link_contact_coop(P, t, 0.0f, L, p3y, p3z, pv3y, pv3z, sn3, cs3, w3, pipeT, c3y, c3z, cm3, qbcast_u<3>(nb_own));
                        pipe_link_contact_coop(P, t, 0.0f, L, p4y, p4z, pv4y, pv4z, sn4, cs4, w4, pipeT, c4y, c4z, cm4, nb4);
                    }
                    {   // fold the cooperative partial sums: link 3's totals go to lane 3's slots, link 4's to every lane
                        f4y = quad_sum(c4y); f4z = quad_sum(c4z); mom4 = quad_sum(cm4);
#if VSQ_SHELF_COOP_FROM <= 3
                        if (SHELF) {
                            const float g3y = quad_sum(cs3y), g3z = quad_sum(cs3z), gm3 = quad_sum(cms3);
                            fy += pick(t == 3, g3y, 0.0f); fz += pick(t == 3, g3z, 0.0f); mom += pick(t == 3, gm3, 0.0f);
                        }
#endif
                        if (PIPE) {
                            const float f3y = quad_sum(c3y), f3z = quad_sum(c3z), m3 = quad_sum(cm3);
                            fy += pick(t == 3, f3y, 0.0f); fz += pick(t == 3, f3z, 0.0f); mom += pick(t == 3, m3, 0.0f);
#if VSQ_PIPE_COOP_FROM <= 2
                            const float f2y = quad_sum(c2y), f2z = quad_sum(c2z), m2 = quad_sum(cm2);
                            fy += pick(t == 2, f2y, 0.0f); fz += pick(t == 2, f2z, 0.0f); mom += pick(t == 2, m2, 0.0f);
#endif
#if VSQ_PIPE_COOP_FROM <= 1
                            const float f1y = quad_sum(c1y), f1z = quad_sum(c1z), m1 = quad_sum(cm1);
                            fy += pick(t == 1, f1y, 0.0f); fz += pick(t == 1, f1z, 0.0f); mom += pick(t == 1, m1, 0.0f);
#endif
                        }
                    }
                    if (SHELF) {
                        // (sfy / sfz: own-lane strip reactions of links 0..3; s4y / s4z: this lane's share of link 4's)
                        const float ty_ = quad_sum(sfy + s4y + s3y), tz_ = quad_sum(sfz + s4z + s3z);
                        csum += sqrtf(ty_ * ty_ + tz_ * tz_);
                    }
                    const float ify = quad_scan_incl(fy, t), ifz = quad_scan_incl(fz, t);
                    const float toty = qbcast<3>(ify), totz = qbcast<3>(ifz);
                    const float sy = (toty - ify) + f4y, sz = (totz - ifz) + f4z;      // forces on the links beyond link t
                    Qt = mom + L * (-cs * sy - sn * sz);                               // lever L n_t, n_t = (-cos, -sin)
                    Q4 = mom4;
                    Qc = toty + f4y;
                }
                // Every cross-lane operand below is a DPP read with ONE consumer of VOP2 shape (v_mul / v_fmac / v_add /
                // v_subrev with the permuted value as src0): the compiler then folds the permutation into the consumer
                // instead of issuing a v_mov_b32_dpp in front of it (45 of the 184 instructions of the round-2 substep were
                // such moves).  Hence the operand orders, the pre-negated factors and the products formed on the SOURCE
                // lane (U, V) rather than on the consumer.
                const float w2 = w * w, w24 = w4 * w4;
                // right-hand sides
                const float dwm = fmaf(nm_t, qprev(w), w);                   // w_t - w_{t-1}  (w_{-1} = 0)
                const float T = eff_t - cj_t * dwm;
                const float T4 = eff4 - cj4 * (w4 - qbcast<3>(w));
                const float Tn = pick(t == 3, T4, qnext(T));
                float r = T - Tn + gb_t * sn;
                float r4 = T4 + gb4 * sn4;
                float rc = eff0 - cj0 * vy - quad_sum(b_t * sn * w2) - b4 * sn4 * w24;
                if (CONTACT) { r += Qt; r4 += Q4; rc += Qc; }
                // centrifugal terms: sum_k a_tk sin(th_t - th_k) w_k^2 = sn_t sum_k a_tk (cs_k w_k^2) - cs_t sum_k a_tk (sn_k w_k^2)
                const float U = cs * w2, V = sn * w2;
                float accU = qbcast<0>(U) * as_k[0], accV = qbcast<0>(V) * as_k[0];
                fmac_bcast<1, true>(accU, U, as_k[1]); fmac_bcast<1, false>(accV, V, as_k[1]);
                fmac_bcast<2, false>(accU, U, as_k[2]); fmac_bcast<2, false>(accV, V, as_k[2]);
                fmac_bcast<3, false>(accU, U, as_k[3]); fmac_bcast<3, false>(accV, V, as_k[3]);
                r = fmaf(cs, accV, fmaf(-sn, accU, r));
                // rows of the 4x4 block (absolute column index k) with the cart already eliminated (constants al_k), link-4 column
                float R[4];
                const float Ac = nb_t * cs, A4c = -b4 * cs4;
#define VINE_QUAD_COL(KK)                                                                                \
                {                                                                                        \
                    float v = qbcast<KK>(cs) * (al_k[KK] * cs);                                          \
                    fmac_bcast<KK, false>(v, sn, a_k[KK] * sn);    /* (sn: written a substep ago) */       \
                    R[KK] = v + nbase[KK];                                                               \
                }
                VINE_QUAD_COL(0) VINE_QUAD_COL(1) VINE_QUAD_COL(2) VINE_QUAD_COL(3)
#undef VINE_QUAD_COL
                const float sin4 = fmaf(sn, cs4, -(cs * sn4));
                float A4 = fmaf(al_4 * cs, cs4, fmaf(a_t4 * sn, sn4, nb4));
                const float as4 = a_t4 * sin4;
                r = fmaf(-as4, w24, r);
                r4 += quad_sum(as4 * w2);
                r = fmaf(be_t * cs, rc, r);
                r4 = fmaf(be4 * cs4, rc, r4);
                const float A44 = fmaf(-b4sq, cs4 * cs4, adiag4);
                // eliminate link 4
                const float ni44 = __builtin_amdgcn_rcpf(-A44);            // -1 / A44
                const float nf4 = A4 * ni44;
                fmac_bcast<0, true>(R[0], A4, nf4); fmac_bcast<1, false>(R[1], A4, nf4);
                fmac_bcast<2, false>(R[2], A4, nf4); fmac_bcast<3, false>(R[3], A4, nf4);
                r = fmaf(nf4, r4, r);
                // Gauss-Jordan on the 4x4 system, one row per lane, pivot row broadcast from lane j
                float pown = 0.0f;
#define VINE_QUAD_PIVOT(J)                                                                               \
                {                                                                                        \
                    const float pinv = __builtin_amdgcn_rcpf(R[J]);                                      \
                    const float m = qbcast<J>(pinv) * R[J];                                              \
                    const float nf = pick(t == J, 0.0f, -m);                                             \
                    /* (the permuted operands were written by the previous pivot step: the checker confirms the distance) */ \
                    if (J < 1) fmac_bcast<J, false>(R[1], R[1], nf);                                     \
                    if (J < 2) fmac_bcast<J, false>(R[2], R[2], nf);                                     \
                    if (J < 3) fmac_bcast<J, false>(R[3], R[3], nf);                                     \
                    fmac_bcast<J, false>(r, r, nf);                                                      \
                    pown = t == J ? pinv : pown;                                                         \
                }
                VINE_QUAD_PIVOT(0) VINE_QUAD_PIVOT(1) VINE_QUAD_PIVOT(2) VINE_QUAD_PIVOT(3)
#undef VINE_QUAD_PIVOT
                const float x = r * pown;                                   // angular acceleration of link t
                const float x4 = (quad_sum(A4 * x) - r4) * ni44;
                const float ydd = (rc - A4c * x4 - quad_sum(Ac * x)) * inv_a00;
                // semi-implicit Euler + incremental rotation of (sin, cos) (see substep() above)
                vy += h * ydd;
                y += h * vy;
                {
                    w += h * x;
                    const float d = h * w;
                    th += d;
                    const float d2 = d * d;
                    const float cd = fmaf(d2, fmaf(d2, 1.0f / 24.0f, -0.5f), 1.0f);
                    const float sd = d * fmaf(d2, -1.0f / 6.0f, 1.0f);
                    const float s_old = sn, c_old = cs;
                    sn = fmaf(s_old, cd, c_old * sd);
                    cs = fmaf(c_old, cd, -(s_old * sd));
                }
                {
                    w4 += h * x4;
                    const float d = h * w4;
                    th4 += d;
                    const float d2 = d * d;
                    const float cd = fmaf(d2, fmaf(d2, 1.0f / 24.0f, -0.5f), 1.0f);
                    const float sd = d * fmaf(d2, -1.0f / 6.0f, 1.0f);
                    const float s_old = sn4, c_old = cs4;
                    sn4 = fmaf(s_old, cd, c_old * sd);
                    cs4 = fmaf(c_old, cd, -(s_old * sd));
                }
            }
            if (SHELF) contact = csum / (float)P.substeps;
            cart_y = y;
            cart_vy = vy;
        };
        {
            int it = 0;
            for (; it + 4 <= P.cfi; it += 4) { VINE_QUAD_ITER(0) VINE_QUAD_ITER(1) VINE_QUAD_ITER(2) VINE_QUAD_ITER(3) }
        }
#undef VINE_QUAD_ITER
#ifdef VSQ_TIMING
        if ((threadIdx.x & 63) == 0) vsq_t[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 5] = wall_clock64();
#endif
        // ---- refreshed rigid-body states: tip = joint 1 + L sum d_k (quad sums over links 0..3, link 4 on top)
        {
            const float L = P.L;
            tip_y = y - L * (quad_sum(sn) + sn4);
            tip_z = P.z1 + L * (quad_sum(cs) + cs4);
            tip_vy = vy - L * (quad_sum(w * cs) + w4 * cs4);
            tip_vz = -L * (quad_sum(w * sn) + w4 * sn4);
        }
        q_own = th - pick(t > 0, qprev(th), 0.0f);
        qd_own = w - pick(t > 0, qprev(w), 0.0f);
        q5 = th4 - qbcast<3>(th);
        qd5 = w4 - qbcast<3>(w);
        float q0 = y, qd0 = vy;
        float prev_q0 = prev_y, prev_qo = prev_q_own, prev_q5v = prev_q5;
        // ---- post_physics_step (V5:1110-1120)
        long long prog = prog_in + 1;
        const bool was_reset = rst != 0;
        if (rst != 0) {      // reset_idx (V5:774-839, 887-914): a quad-uniform branch
            const float ten = 0.17453292519943295f;
            float qn1_4[4], qn5, qn0, depth, pdepth;
            if (reset_values) {
                const float* v = reset_values + (size_t)e * 10;
                qn1_4[0] = v[0]; qn1_4[1] = v[1]; qn1_4[2] = v[2]; qn1_4[3] = v[3];
                qn5 = v[4]; qn0 = v[5]; pdepth = v[6]; ty = v[7]; tz = v[8]; depth = v[9];
            } else {
                // lanes 0..2 draw one Philox word quadruple each; the ten uniforms are then broadcast
                unsigned r[4];
                rng4(P, (unsigned)e, step, RNG_RESET, (unsigned)(t < 3 ? t : 0), r);
                qn1_4[0] = -ten + (2.0f * ten) * u01(qbcast_u<0>(r[0]));
                qn1_4[1] = -ten + (2.0f * ten) * u01(qbcast_u<0>(r[1]));
                qn1_4[2] = -ten + (2.0f * ten) * u01(qbcast_u<0>(r[2]));
                qn1_4[3] = -ten + (2.0f * ten) * u01(qbcast_u<0>(r[3]));
                qn5 = -ten + (2.0f * ten) * u01(qbcast_u<1>(r[0]));
                qn0 = P.cart_min + P.cart_span * u01(qbcast_u<1>(r[1]));
                pdepth = P.depth_min + P.depth_span * u01(qbcast_u<1>(r[2]));
                ty = P.ty_min + P.ty_span * u01(qbcast_u<1>(r[3]));
                tz = P.tz_min + P.tz_span * u01(qbcast_u<2>(r[0]));
                depth = P.depth_min + P.depth_span * u01(qbcast_u<2>(r[1]));
            }
            if (!(P.flags & VINE_FLAG_RANDOMIZE_DOF_INIT)) {
                qn1_4[0] = qn1_4[1] = qn1_4[2] = qn1_4[3] = 0.0f; qn5 = 0.0f; qn0 = 0.0f;
            }
            if (!(P.flags & VINE_FLAG_RANDOMIZE_TARGETS)) { ty = P.ty_max; tz = P.tz_fixed; }
            (void)depth; (void)pdepth;
            if (SHELF) obj_depth = depth;
            if (SHELF && t == 0) {             // obstacle poses of the new episode (reset_env's arithmetic, V5:818-885)
                ST(VF_SHELF_Y) = ty + (-0.2f + depth);
                ST(VF_SHELF_Z) = tz - 0.01f;
                ST(VF_OBJ_DEPTH) = depth;
            }
            if (PIPE) {
                const float R = 0.0735f;             // PIPE_RADIUS = 0.07 * 1.05 (V5:88)
                const float ez = 1.0f - tz;
                const float deg = ((13199.0f * ez - 12276.0f) * ez + 4045.0f) * ez - 447.0f;
                const float tp = deg * 0.017453292519943295f;
                float stp, ctp;
                sincosf(tp, &stp, &ctp);
                obj_depth = pdepth; obj_angle = tp;      // (what the observation row shows: the new episode's)
                if (t == 0) {
                    ST(VF_PIPE_Y) = ty + pdepth * ctp + R * stp;
                    ST(VF_PIPE_Z) = tz + pdepth * stp - R * ctp;
                    ST(VF_OBJ_DEPTH) = pdepth;
                    ST(VF_OBJ_ANGLE) = tp;
                }
            }
            rst = 0;
            prog = 0;
            q_own = sel4(t, qn1_4[0], qn1_4[1], qn1_4[2], qn1_4[3]);
            q5 = qn5; q0 = qn0;
            qd_own = 0.0f; qd5 = 0.0f; qd0 = 0.0f;
            prev_qo = q_own; prev_q5v = q5; prev_q0 = q0;
            ST(VF_Q0 + 1 + t) = q_own; ST(VF_QD0 + 1 + t) = 0.0f; ST(VF_PREV_Q0 + 1 + t) = q_own;
            if (t == 0) {
                ST(VF_Q0) = q0; ST(VF_QD0) = 0.0f; ST(VF_PREV_Q0) = q0;
                ST(VF_Q0 + 5) = q5; ST(VF_QD0 + 5) = 0.0f; ST(VF_PREV_Q0 + 5) = q5;
                ST(VF_TARGET_Y) = ty; ST(VF_TARGET_Z) = tz;
            }
            if (P.flags & VINE_FLAG_STALE_BODY_STATE_AFTER_RESET) {
                prev_tip_y = tip_y;        // tip/cart rigid-body states keep their pre-reset values (V5:796-797 TODO)
                prev_tip_z = tip_z;
            } else {
                // forward kinematics of the reset pose (zero rates): prefix sums of the new joint angles
                float s = q_own + pick(t > 0, qprev(q_own), 0.0f);
                const float thn = s + pick(t >= 2, qperm<0x4E>(s), 0.0f);
                const float th4n = qbcast<3>(thn) + q5;
                float s_, c_;
                sincosf(thn, &s_, &c_);
                const float snn = P.s0 * c_ + P.c0 * s_, csn = P.c0 * c_ - P.s0 * s_;
                sincosf(th4n, &s_, &c_);
                tip_y = q0 - P.L * (quad_sum(snn) + (P.s0 * c_ + P.c0 * s_));
                tip_z = P.z1 + P.L * (quad_sum(csn) + (P.c0 * c_ - P.s0 * s_));
                tip_vy = 0.0f; tip_vz = 0.0f;
                prev_tip_y = tip_y; prev_tip_z = tip_z;
                cart_y = q0; cart_vy = 0.0f;
            }
            prev_u_rail = 0.0f;
            pce = 0.0f;
            agg = 0.0f;
        } else {
            ST(VF_Q0 + 1 + t) = q_own; ST(VF_QD0 + 1 + t) = qd_own;
            if (t == 0) { ST(VF_Q0) = q0; ST(VF_QD0) = qd0; ST(VF_Q0 + 5) = q5; ST(VF_QD0 + 5) = qd5; }
            if (introspect) {
                ST(VF_PREV_Q0 + 1 + t) = prev_qo;
                if (t == 0) { ST(VF_PREV_Q0) = prev_q0; ST(VF_PREV_Q0 + 5) = prev_q5v; }
            }
        }
        // ---- compute_observations (V5:1339-1390): the row is assembled replicated, then lane t keeps, perturbs, clamps
        // and stores columns [4t, 4t+4) and [16+4t, 16+4t+4)
        float o[32];
        {
            const float fd_tip_y = (tip_y - prev_tip_y) * P.inv_cdt, fd_tip_z = (tip_z - prev_tip_z) * P.inv_cdt;
            int k = 0;
            if (OBS_TYPE == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) {
                const float fd_own = (q_own - prev_qo) * P.inv_cdt;
                o[k++] = q0;
                o[k++] = qbcast<0>(q_own); o[k++] = qbcast<1>(q_own); o[k++] = qbcast<2>(q_own); o[k++] = qbcast<3>(q_own);
                o[k++] = q5;
                o[k++] = (q0 - prev_q0) * P.inv_cdt;
                o[k++] = qbcast<0>(fd_own); o[k++] = qbcast<1>(fd_own); o[k++] = qbcast<2>(fd_own); o[k++] = qbcast<3>(fd_own);
                o[k++] = (q5 - prev_q5v) * P.inv_cdt;
            } else {
                o[k++] = q0;
                o[k++] = (q0 - prev_q0) * P.inv_cdt;
            }
            o[k++] = 0.0f; o[k++] = tip_y; o[k++] = tip_z;
            o[k++] = 0.0f; o[k++] = fd_tip_y; o[k++] = fd_tip_z;
            o[k++] = 0.0f; o[k++] = ty; o[k++] = tz;
            o[k++] = 0.0f; o[k++] = 0.0f; o[k++] = 0.0f;
            o[k++] = smoothed; o[k++] = prev_u_rail; o[k++] = obj_depth; o[k++] = obj_angle;
#pragma unroll
            for (int i = 0; i < NOBS; ++i) o[i] = o[i] * P.inv_obs_scale[i];
#pragma unroll
            for (int i = NOBS; i < 32; ++i) o[i] = 0.0f;
        }
        float mine[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mine[c] = sel4(t, o[c], o[4 + c], o[8 + c], o[12 + c]);
            mine[4 + c] = sel4(t, o[16 + c], o[20 + c], o[24 + c], o[28 + c]);
        }
        if (RANDOMIZE && P.obs_noise != 0.0f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) mine[j] += P.obs_noise * onoise[j];      // (drawn in the prologue)
        }
        {
            float* orow = obs + (size_t)e * NOBS;
#pragma unroll
            for (int j = 0; j < 8; ++j) mine[j] = clampf(mine[j], P.clip_obs);
            if (NOBS % 4 == 0) {
                reinterpret_cast<float4*>(orow)[t] = make_float4(mine[0], mine[1], mine[2], mine[3]);
                if (16 + 4 * t < NOBS) reinterpret_cast<float4*>(orow)[4 + t] = make_float4(mine[4], mine[5], mine[6], mine[7]);
            } else {       // 18 columns: 72-B rows are 8-B aligned only
                reinterpret_cast<float2*>(orow)[2 * t] = make_float2(mine[0], mine[1]);
                reinterpret_cast<float2*>(orow)[2 * t + 1] = make_float2(mine[2], mine[3]);
                if (t == 0) reinterpret_cast<float2*>(orow)[8] = make_float2(mine[4], mine[5]);
            }
        }
#ifdef VSQ_TIMING
        if ((threadIdx.x & 63) == 0) vsq_t[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 6] = wall_clock64();
#endif
        // ---- compute_reward (V5:1218-1331, 1470-1537), compute_reset (V5:1540-1558): replicated, lane 0 stores
        const float dy = tip_y - ty, dz = tip_z - tz;
        const float dist = sqrtf(dy * dy + dz * dz);
        const bool reached = dist < P.success_dist;
        const bool limit_hit = (cart_y > P.soft_limit) || (cart_y < -P.soft_limit);
        const bool tip_limit_hit = tip_y < ty;
        const float vnorm = sqrtf(tip_vy * tip_vy + tip_vz * tip_vz);
        const float cmean = SHELF ? contact_sum / (float)P.cfi : 0.0f;      // V5:1242-1248
        float rm[VINE_NUM_REWARDS];
        rm[0] = -dist;
        rm[1] = -1.0f;
        rm[2] = reached ? 1000.0f : 0.0f;
        rm[3] = -(reached ? vnorm : 0.0f);
        rm[4] = vnorm;
        rm[5] = -fabsf(u_rail);
        rm[6] = -fabsf(u_fpam);
        rm[7] = -fabsf(u_rail - prev_u_rail);
        rm[8] = -fabsf(u_fpam - smoothed);
        rm[9] = limit_hit ? -100.0f : 0.0f;
        rm[10] = -fabsf(cart_y);
        rm[11] = tip_limit_hit ? -100.0f : 0.0f;
        rm[12] = -((cmean > 0.0f) ? cmean : 0.0f);
        float total = 0.0f;
#pragma unroll
        for (int i = 0; i < VINE_NUM_REWARDS; ++i) total += rm[i] * P.rw[i];
        agg += total;
        if (prog >= (long long)P.max_len - 1) rst = 1;
        if (reached && (P.flags & VINE_FLAG_USE_TARGET_REACHED_RESET)) rst = 1;
        if (tip_limit_hit && (P.flags & VINE_FLAG_USE_TIP_LIMIT_HIT_RESET)) rst = 1;
        if (limit_hit) rst = 1;
        if (SHELF && cmean > 0.0f && (P.flags & VINE_FLAG_USE_NONZERO_CONTACT_FORCE_RESET)) rst = 1;
        const unsigned char to = (prog >= (long long)P.max_len - 1) && (rst != 0);
        if (t == 0) {
            rew[e] = total;
            reset[e] = rst;
            progress[e] = prog;
            timeouts[e] = to;
            if (reward_matrix) {
#pragma unroll
                for (int i = 0; i < VINE_NUM_REWARDS; ++i) reward_matrix[(size_t)e * VINE_NUM_REWARDS + i] = rm[i];
            }
            if (introspect || was_reset) {      // (else: re-derived by the next step, see the top of the kernel)
                ST(VF_TIP_Y) = tip_y; ST(VF_TIP_Z) = tip_z;
                ST(VF_CART_Y) = cart_y; ST(VF_CART_VY) = cart_vy;
            }
            ST(VF_SMOOTHED_U) = smoothed;
            ST(VF_PREV_CART_VEL) = pcv; ST(VF_PREV_CART_VEL_ERR) = pce;
            ST(VF_AGG_REW) = agg;
            if (SHELF) ST(VF_CONTACT) = contact;
            if (introspect) {
                ST(VF_TIP_VY) = tip_vy; ST(VF_TIP_VZ) = tip_vz;
                ST(VF_PREV_TIP_Y) = prev_tip_y; ST(VF_PREV_TIP_Z) = prev_tip_z;
                ST(VF_U_FPAM) = u_fpam; ST(VF_U_RAIL) = u_rail; ST(VF_PREV_U_RAIL) = prev_u_rail;
                ST(VF_RAIL_FORCE) = rail_force;
                if (SHELF) ST(VF_CONTACT_MEAN) = cmean;
            }
        }
        if (ROLL) {
            // ---- vine_rollout_post's bookkeeping (play_steps_rnn; common_agent.py:293-306): shaped reward with the time-out
            // bootstrap, done flag, episode accumulators; the LSTM-state rows of a finished env are cleared by its four lanes
            const bool done = rst != 0;
            const float cr = roll_cr + total, cl = roll_cl + 1.0f;
            if (t == 0) {
                float sh = (total + R.shift) * R.scale;
                if (R.gamma_b != 0.0f && to) sh += R.gamma_b * roll_value;
                R.shaped[e] = sh;
                R.dones[e] = done ? 1 : 0;
                R.cur_r[e] = done ? 0.0f : cr;
                R.cur_l[e] = done ? 0.0f : cl;
                if (done) { roll_sr = cr; roll_sl = cl; roll_cnt = 1.0f; }
            }
            if (done) {
                const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                float4* hr = reinterpret_cast<float4*>(R.h_state + (size_t)e * 256) + t;
                float4* cr4 = reinterpret_cast<float4*>(R.c_state + (size_t)e * 256) + t;
#pragma unroll
                for (int i = 0; i < 16; ++i) { hr[4 * i] = z; cr4[4 * i] = z; }
                if (R.h_op) {
                    float4* orow = reinterpret_cast<float4*>(R.h_op + (size_t)e * R.h_op_stride) + t;
#pragma unroll
                    for (int i = 0; i < 16; ++i) orow[4 * i] = z;
                }
            }
        }
    }
    if (ROLL) {
        // one {sum of finished returns, sum of finished lengths, count} row per workgroup, fixed order (no atomics)
        __shared__ float roll_red[4][3];
        float v3[3] = {roll_sr, roll_sl, roll_cnt};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v3[k] += __shfl_xor(v3[k], off, 64);
        }
        if ((threadIdx.x & 63) == 0) { roll_red[threadIdx.x >> 6][0] = v3[0]; roll_red[threadIdx.x >> 6][1] = v3[1]; roll_red[threadIdx.x >> 6][2] = v3[2]; }
        __syncthreads();
        if (threadIdx.x < 3)
            R.partial[blockIdx.x * 3 + threadIdx.x] = (roll_red[0][threadIdx.x] + roll_red[1][threadIdx.x]) +
                                                      (roll_red[2][threadIdx.x] + roll_red[3][threadIdx.x]);
    }
#ifdef VSQ_TIMING
    if ((threadIdx.x & 63) == 0) vsq_t[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 7] = wall_clock64();
#endif
    step_arrive(counters);
}

// Tip / cart rigid-body fields of env e from its DOF state (forward kinematics): what a `refresh_rigid_body_state_tensor`
// after the last simulate left there.
__device__ __forceinline__ void refresh_body_from_dofs(const DevParams& P, float* __restrict__ st, int n, int e) {
    float th[NL], w[NL], tip[4];
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int k = 0; k < NL; ++k) { a += ST(VF_Q0 + 1 + k); b += ST(VF_QD0 + 1 + k); th[k] = a; w[k] = b; }
    const float y = ST(VF_Q0), vy = ST(VF_QD0);
    tip_fk(P, y, vy, th, w, tip);
    ST(VF_TIP_Y) = tip[0]; ST(VF_TIP_Z) = tip[1]; ST(VF_TIP_VY) = tip[2]; ST(VF_TIP_VZ) = tip[3];
    ST(VF_CART_Y) = y; ST(VF_CART_VY) = vy;
}

// Launched once by the first vine_step after introspection was switched ON mid-run: from then on the step kernels load the
// body states from memory again, and the copies of envs that were not reset in the last step are out of date.
__global__ void vine_refresh_body_kernel(const DevParams P, float* __restrict__ st, const long long* __restrict__ progress) {
    const int n = P.n;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || progress[e] == 0) return;
    refresh_body_from_dofs(P, st, n, e);
}

// reset_idx(env_ids) from outside the step (vec_task.py:412-427; V5:715-718).
__global__ void vine_reset_idx_kernel(const DevParams P, float* __restrict__ st, const long long* __restrict__ env_ids,
                                      long long count, float* __restrict__ rew, long long* __restrict__ reset,
                                      long long* __restrict__ progress, const float* __restrict__ reset_values,
                                      const unsigned long long* __restrict__ counters) {
    const int n = P.n;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    long long id = env_ids[i];
    if (id < 0 || id >= n) return;
    const int e = (int)id;
    const unsigned long long step = step_of(P, counters) | (1ull << 62);
    // Without introspection the four-lanes-per-env kernel stores the tip / cart rigid-body states only in the step in which
    // an env was reset (it re-derives them from the DOF state otherwise): bring the memory copies up to date from the
    // pre-reset DOF state before they become the stale-by-design body state (P5) of the reset env.  progress == 0 marks
    // envs whose stored body state IS the only copy (reset in the last step, or never stepped).
    if (!(P.flags & VINE_FLAG_INTROSPECT) && progress && progress[e] != 0) refresh_body_from_dofs(P, st, n, e);
    float qn[ND], ty, tz;
    reset_env(P, st, n, e, step, reset_values, qn, ty, tz);
    ST(VF_PREV_TIP_Y) = ST(VF_TIP_Y);
    ST(VF_PREV_TIP_Z) = ST(VF_TIP_Z);
    ST(VF_PREV_U_RAIL) = 0.0f;
    ST(VF_PREV_CART_VEL_ERR) = 0.0f;
    ST(VF_AGG_REW) = 0.0f;
    if (!(P.flags & VINE_FLAG_STALE_BODY_STATE_AFTER_RESET)) {
        float th[NL], w[NL] = {0, 0, 0, 0, 0}, tip[4];
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < NL; ++k) { a += qn[k + 1]; th[k] = a; }
        tip_fk(P, qn[0], 0.0f, th, w, tip);
        ST(VF_TIP_Y) = tip[0]; ST(VF_TIP_Z) = tip[1]; ST(VF_TIP_VY) = 0.0f; ST(VF_TIP_VZ) = 0.0f;
        ST(VF_PREV_TIP_Y) = tip[0]; ST(VF_PREV_TIP_Z) = tip[1];
        ST(VF_CART_Y) = qn[0]; ST(VF_CART_VY) = 0.0f;
    }
    if (reset) reset[e] = 0;
    if (progress) progress[e] = 0;
    if (rew) rew[e] = 0.0f;
}

// Initial asset pose: all DOFs zero (V5:440-445), body states from FK, shelf at (0, 0.2, 0) (V5:468-470).
__global__ void vine_init_kernel(const DevParams P, float* __restrict__ st) {
    const int n = P.n;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float th[NL] = {0, 0, 0, 0, 0}, w[NL] = {0, 0, 0, 0, 0}, tip[4];
    tip_fk(P, 0.0f, 0.0f, th, w, tip);
    ST(VF_TIP_Y) = tip[0]; ST(VF_TIP_Z) = tip[1];
    ST(VF_PREV_TIP_Y) = tip[0]; ST(VF_PREV_TIP_Z) = tip[1];
    ST(VF_SHELF_Y) = 0.2f;
    if (P.flags & VINE_FLAG_CREATE_PIPE) {   // V5:482-484 initial pose, identity orientation
        ST(VF_PIPE_Y) = -0.4f; ST(VF_PIPE_Z) = 0.5f; ST(VF_OBJ_ANGLE) = -1.5707963267948966f;
    }
}


// ---- vine_stats: the dashboard scalars of compute_reward (V5:1250-1322) as one two-stage reduction ----
// Stage 1: STATS_BLOCKS workgroups, each folds its grid-strided share of the envs into one row of partial sums /
// maxima (double sums: the result does not depend on the launch geometry beyond round-off of doubles, and the same
// inputs give the same bits every time -- no atomics).  Stage 2: one workgroup folds the rows in a fixed order and
// writes the VineStat vector.  The variance of aggregated_rew_buf is accumulated about a pilot value (env 0) so
// that sum / sum-of-squares in double is exact enough for torch.std's unbiased estimate.
#define STATS_BLOCKS 64
#define STATS_NSUM (17 + VINE_NUM_REWARDS)        // sums: 16 plain + agg (shifted) + agg^2 (shifted) ... see enum
#define STATS_NMAX (4 + 2 * VINE_NUM_REWARDS)      // maxima: |tip_y|, tip_z, |v_tip|, rew, then max / -min per term
enum { SS_DIST = 0, SS_REACHED, SS_LIMIT, SS_TIPLIM, SS_ABSTIPY, SS_TIPZ, SS_TIPV, SS_URAIL, SS_PURAIL, SS_RFORCE, SS_UFPAM,
       SS_SMOOTH, SS_PROG, SS_CONTACT, SS_NONZERO, SS_REW, SS_AGG, SS_AGG2, SS_TERM0 };
static_assert(SS_TERM0 + VINE_NUM_REWARDS == STATS_NSUM + 1, "stat slots");

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(256) void vine_stats_partial_kernel(const DevParams P, const float* __restrict__ st,
                                                                 const float* __restrict__ rew,
                                                                 const long long* __restrict__ progress,
                                                                 const float* __restrict__ reward_matrix,
                                                                 double* __restrict__ psum, float* __restrict__ pmax) {
    constexpr int NS = STATS_NSUM + 1, NM = STATS_NMAX;
    const int n = P.n;
    double s[NS];
    float m[NM];
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NM; ++i) m[i] = -3.0e38f;
    const float pilot = st[(size_t)VF_AGG_REW * n];        // env 0's value: shift for the variance
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const float ty = ST(VF_TIP_Y), tz = ST(VF_TIP_Z), gy = ST(VF_TARGET_Y), gz = ST(VF_TARGET_Z);
        const float dy = ty - gy, dz = tz - gz;
        const float dist = sqrtf(dy * dy + dz * dz);
        const float cy = ST(VF_CART_Y);
        const float vy = ST(VF_TIP_VY), vz = ST(VF_TIP_VZ);
        const float tv = sqrtf(vy * vy + vz * vz);
        const float cm = ST(VF_CONTACT_MEAN);
        const float r = rew[e];
        const float ag = ST(VF_AGG_REW) - pilot;
        s[SS_DIST] += dist; s[SS_REACHED] += dist < P.success_dist ? 1.0 : 0.0;
        s[SS_LIMIT] += ((cy > P.soft_limit) || (cy < -P.soft_limit)) ? 1.0 : 0.0;
        s[SS_TIPLIM] += ty < gy ? 1.0 : 0.0; s[SS_ABSTIPY] += fabsf(ty); s[SS_TIPZ] += tz; s[SS_TIPV] += tv;
        s[SS_URAIL] += fabsf(ST(VF_U_RAIL)); s[SS_PURAIL] += fabsf(ST(VF_PREV_U_RAIL)); s[SS_RFORCE] += fabsf(ST(VF_RAIL_FORCE));
        s[SS_UFPAM] += fabsf(ST(VF_U_FPAM)); s[SS_SMOOTH] += fabsf(ST(VF_SMOOTHED_U)); s[SS_PROG] += (double)progress[e];
        s[SS_CONTACT] += cm; s[SS_NONZERO] += cm > 0.0f ? 1.0 : 0.0; s[SS_REW] += r;
        s[SS_AGG] += ag; s[SS_AGG2] += (double)ag * (double)ag;
        m[0] = fmaxf(m[0], fabsf(ty)); m[1] = fmaxf(m[1], tz); m[2] = fmaxf(m[2], tv); m[3] = fmaxf(m[3], r);
        if (reward_matrix) {
#pragma unroll
            for (int k = 0; k < VINE_NUM_REWARDS; ++k) {
                const float v = reward_matrix[(size_t)e * VINE_NUM_REWARDS + k];
                s[SS_TERM0 + k] += v;
                m[4 + 2 * k] = fmaxf(m[4 + 2 * k], v);
                m[5 + 2 * k] = fmaxf(m[5 + 2 * k], -v);
            }
        }
    }
    __shared__ double sred[4][NS];
    __shared__ float mred[4][NM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) { const double v = wave_sum_d(s[i]); if (lane == 0) sred[wave][i] = v; }
#pragma unroll
    for (int i = 0; i < NM; ++i) { const float v = wave_max_f(m[i]); if (lane == 0) mred[wave][i] = v; }
    __syncthreads();
    if (threadIdx.x < NS)
        psum[(size_t)blockIdx.x * NS + threadIdx.x] = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x];
    if (threadIdx.x < NM)
        pmax[(size_t)blockIdx.x * NM + threadIdx.x] = fmaxf(fmaxf(mred[0][threadIdx.x], mred[1][threadIdx.x]), fmaxf(mred[2][threadIdx.x], mred[3][threadIdx.x]));
}

__global__ __launch_bounds__(64) void vine_stats_finalize_kernel(const DevParams P, const float* __restrict__ st, int blocks,
                                                                 const double* __restrict__ psum, const float* __restrict__ pmax,
                                                                 int has_terms, long long view, float* __restrict__ out) {
    constexpr int NS = STATS_NSUM + 1, NM = STATS_NMAX;
    __shared__ double S[NS];
    __shared__ float M[NM];
    const int t = threadIdx.x, n = P.n;
    if (t < NS) { double v = 0.0; for (int b = 0; b < blocks; ++b) v += psum[(size_t)b * NS + t]; S[t] = v; }
    if (t < NM) { float v = -3.0e38f; for (int b = 0; b < blocks; ++b) v = fmaxf(v, pmax[(size_t)b * NM + t]); M[t] = v; }
    __syncthreads();
    for (int i = t; i < VINE_NUM_STATS; i += 64) out[i] = 0.0f;
    __syncthreads();
    const double inv = 1.0 / (double)n;
    if (t == 0) {
        static const int dst[16] = {VS_DIST_MEAN, VS_TARGET_REACHED, VS_LIMIT_HIT, VS_TIP_LIMIT_HIT, VS_ABS_TIP_Y, VS_TIP_Z, VS_TIP_VEL_MEAN,
                                    VS_U_RAIL_ABS, VS_PREV_U_RAIL_ABS, VS_RAIL_FORCE_ABS, VS_U_FPAM_ABS, VS_SMOOTHED_ABS,
                                    VS_PROGRESS_MEAN, VS_CONTACT_MEAN, VS_CONTACT_NONZERO, VS_REW_MEAN};
        for (int i = 0; i < 16; ++i) out[dst[i]] = (float)(S[i] * inv);
        out[VS_MAX_ABS_TIP_Y] = M[0]; out[VS_MAX_TIP_Z] = M[1]; out[VS_TIP_VEL_MAX] = M[2]; out[VS_REW_MAX] = M[3];
        const double pilot = (double)st[(size_t)VF_AGG_REW * n];
        const double mean_s = S[SS_AGG] * inv;                       // mean of the shifted values
        double var = n > 1 ? (S[SS_AGG2] - (double)n * mean_s * mean_s) / (double)(n - 1) : 0.0;
        if (var < 0.0) var = 0.0;
        out[VS_AGG_MEAN] = (float)(pilot + mean_s);
        out[VS_AGG_STD] = (float)sqrt(var);
    }
    const int e = (int)view;
    if (t < 6) { out[VS_VIEW0 + t] = ST(VF_Q0 + t); out[VS_VIEW0 + 6 + t] = ST(VF_QD0 + t); out[VS_VIEW0 + 12 + t] = ST(VF_PREV_Q0 + t); }
    if (t == 6) {
        const int vf[10] = {VF_TIP_Y, VF_TIP_Z, VF_TIP_VY, VF_TIP_VZ, VF_PREV_TIP_Y, VF_PREV_TIP_Z, VF_CART_Y, VF_CART_VY, VF_TARGET_Y, VF_TARGET_Z};
        for (int i = 0; i < 10; ++i) out[VS_VIEW0 + 18 + i] = ST(vf[i]);
        const int vu[5] = {VF_U_FPAM, VF_SMOOTHED_U, VF_U_RAIL, VF_RAIL_FORCE, VF_CONTACT_MEAN};
        for (int i = 0; i < 5; ++i) out[VS_VIEW_U + i] = ST(vu[i]);
    }
    if (has_terms && t >= 16 && t < 16 + VINE_NUM_REWARDS) {
        const int k = t - 16;
        out[VS_TERM0 + 3 * k] = (float)(S[SS_TERM0 + k] * inv);
        out[VS_TERM0 + 3 * k + 1] = M[4 + 2 * k];
        out[VS_TERM0 + 3 * k + 2] = -M[5 + 2 * k];
    }
}

int validate(const VineConfig* c) {
    if (!c) return fail(VINE_ERR_INVALID_ARG, "cfg is NULL");
    if (c->abi_version != VINE_ABI_VERSION) return fail(VINE_ERR_INVALID_ARG, "abi_version mismatch");
    if (c->num_envs <= 0) return fail(VINE_ERR_INVALID_ARG, "num_envs must be positive");
    if (c->control_freq_inv <= 0 || c->substeps <= 0)
        return fail(VINE_ERR_INVALID_ARG, "control_freq_inv/substeps must be positive");
    if (c->action_delay < 0 || c->action_delay > VINE_MAX_DELAY)
        return fail(VINE_ERR_INVALID_ARG, "ACTION_DELAY out of range");
    if (vine_num_obs(c) < 0) return VINE_ERR_UNSUPPORTED;
    return VINE_OK;
}

void make_params(const VineConfig& c, DevParams& P) {
    memset(&P, 0, sizeof P);
    P.n = c.num_envs; P.num_obs = vine_num_obs(&c); P.obs_type = c.obs_type; P.cfi = c.control_freq_inv;
    P.substeps = c.substeps; P.max_len = c.max_episode_length; P.delay = c.action_delay; P.flags = c.flags;
    P.seed_lo = (unsigned)c.seed; P.seed_hi = (unsigned)(c.seed >> 32); P.env_off = (unsigned)c.env_id_offset;
    P.dt = c.dt; P.hsub = c.dt / (float)c.substeps; P.cdt = c.dt * (float)c.control_freq_inv;
    P.inv_dt = (float)(1.0 / (double)P.dt); P.inv_cdt = (float)(1.0 / (double)P.cdt);
    P.clip_obs = c.clip_observations; P.clip_act = c.clip_actions;
    P.fpam_min = c.fpam_min; P.fpam_span = (float)((double)c.fpam_max - (double)c.fpam_min);
    P.rail_scale = c.rail_velocity_scale; P.damping = c.damping; P.kq = c.stiffness; P.cad = c.link_angular_damping;
    P.eff_lim = c.effort_limit;
    P.soft_limit = c.rail_soft_limit; P.p_gain = c.rail_p_gain; P.d_gain = c.rail_d_gain; P.rail_acc = c.rail_acceleration;
    P.alpha_inf = c.smoothing_alpha_inflate; P.alpha_def = c.smoothing_alpha_deflate; P.success_dist = c.success_dist;
    P.cart_min = c.random_init_cart_min_y; P.cart_span = c.random_init_cart_max_y - c.random_init_cart_min_y;
    P.ty_min = c.min_target_y; P.ty_span = c.max_target_y - c.min_target_y; P.ty_max = c.max_target_y;
    P.tz_min = c.min_target_z; P.tz_span = c.max_target_z - c.min_target_z; P.tz_fixed = c.min_target_z;
    P.depth_min = c.min_target_depth; P.depth_span = c.max_target_depth - c.min_target_depth;
    P.dyn_min = c.dyn_scale_min; P.dyn_span = c.dyn_scale_max - c.dyn_scale_min;
    P.obs_noise = c.obs_noise_std; P.act_noise = c.action_noise_std;
    P.g = c.gravity; P.L = c.link_length; P.z1 = c.joint1_z;
    P.s0 = (float)sin((double)c.phi0); P.c0 = (float)cos((double)c.phi0);
    // composite constants of the absolute-angle Lagrangian, accumulated in double
    double m[NL], mt = c.cart_mass, L = c.link_length, l = c.link_com;
    for (int i = 0; i < NL; ++i) { m[i] = c.link_mass[i]; mt += m[i]; }
    P.mtot = (float)mt;
    double b[NL];
    for (int i = 0; i < NL; ++i) {
        double distal = 0;
        for (int k = i + 1; k < NL; ++k) distal += m[k];
        b[i] = m[i] * l + L * distal;
        P.b[i] = (float)b[i];
        P.gb[i] = (float)((double)c.gravity * b[i]);
        P.I[i] = c.link_inertia[i];
        P.a[i][i] = (float)(m[i] * l * l + L * L * distal + (double)c.link_inertia[i]);
    }
    for (int i = 0; i < NL; ++i)
        for (int j = 0; j < NL; ++j)
            if (i != j) P.a[i][j] = (float)(L * b[i > j ? i : j]);
    for (int i = 0; i < NL; ++i) { P.K[i] = c.fpam_K[i]; P.C[i] = c.fpam_C[i]; P.bb[i] = c.fpam_b[i]; P.B[i] = c.fpam_B[i]; }
    for (int i = 0; i < VINE_NUM_REWARDS; ++i) P.rw[i] = c.reward_weights[i];
    for (int i = 0; i < VINE_MAX_OBS; ++i) P.inv_obs_scale[i] = (float)(1.0 / (double)c.obs_scaling[i]);
}

}  // namespace

struct VineHandle {
    VineConfig cfg;
    DevParams P;
    int device;
    float* state;
    bool owns_state;
    unsigned long long* counters;  // [0] step-count base, [1] finished workgroups of step launches since (step_of)
    const float* reset_values;
    float* reward_matrix;
    bool refresh_body;             // introspection was switched on since the last step: the next vine_step refreshes the lazily
                                   // stored tip / cart body states first (vine_refresh_body_kernel)
    int step_kernel;               // 0 = by size, 1 = one lane per env, 2 = four lanes per env where it applies (VINE_STEP_KERNEL)
    double* stats_psum;            // [STATS_BLOCKS][STATS_NSUM + 1] partial sums of vine_stats (allocated on first use)
    float* stats_pmax;             // [STATS_BLOCKS][STATS_NMAX]
};

namespace {
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
}  // namespace

extern "C" {

const char* vine_last_error(void) { return g_err; }
const char* vine_backend_name(void) { return "hip-gfx950"; }

int vine_config_default(VineConfig* c) {
    if (!c) return fail(VINE_ERR_INVALID_ARG, "cfg is NULL");
    memset(c, 0, sizeof *c);
    c->abi_version = VINE_ABI_VERSION;
    c->num_envs = 4096;
    c->control_freq_inv = 4;
    c->substeps = 10;
    c->max_episode_length = 500;
    c->action_delay = 1;
    c->flags = VINE_FLAG_USE_SMOOTHED_FPAM | VINE_FLAG_RANDOMIZE_DOF_INIT | VINE_FLAG_RANDOMIZE_TARGETS |
               VINE_FLAG_USE_TARGET_REACHED_RESET | VINE_FLAG_VINE_RANDOMIZE | VINE_FLAG_STALE_BODY_STATE_AFTER_RESET |
               VINE_FLAG_IMPLICIT_JOINT_DAMPING;
    c->seed = 42;
    c->dt = 0.00833f; c->gravity = 9.81f; c->clip_observations = 5.0f; c->clip_actions = 1.0f;
    c->fpam_min = -0.1f; c->fpam_max = 3.0f; c->rail_velocity_scale = 1.0f;
    c->damping = 2e-2f; c->stiffness = 0.0f;
    c->rail_soft_limit = 0.3f; c->rail_p_gain = 10.0f; c->rail_d_gain = 0.0f; c->rail_acceleration = 8.0f;
    c->smoothing_alpha_inflate = 0.81f; c->smoothing_alpha_deflate = 0.86f;
    c->random_init_cart_min_y = (float)(-0.1 * 0.3); c->random_init_cart_max_y = 0.3f;
    c->success_dist = 0.08f;
    c->min_target_depth = -0.05f; c->max_target_depth = 0.2f;
    c->min_target_y = -0.48f; c->max_target_y = -0.4f; c->min_target_z = 0.58f; c->max_target_z = 0.67f;
    const float w[VINE_NUM_REWARDS] = {0, 0, 1.0f, 0, 0.1f, 0, 0, 0, 0, 1.0f, 0, 0, 0.10f};
    memcpy(c->reward_weights, w, sizeof w);
    c->dyn_scale_min = 0.999f; c->dyn_scale_max = 1.001f;
    c->cart_mass = 0.4f;
    for (int i = 0; i < NL; ++i) { c->link_mass[i] = 0.005f; c->link_inertia[i] = 0.00000689246f; }
    c->link_mass[4] = 0.1f; c->link_inertia[4] = 0.000101559f;
    c->link_length = 0.0885f; c->link_com = 0.04425f;
    c->joint1_z = (float)(1.0 - 0.025 - 0.01); c->phi0 = 3.1415f;
    const float K[NL] = {0.8385f, 1.5400f, 1.5109f, 1.2887f, 0.4347f};
    const float C[NL] = {0.0178f, 0.0304f, 0.0528f, 0.0367f, 0.0223f};
    const float b[NL] = {0.0007f, 0.0062f, 0.0402f, 0.0160f, 0.0133f};
    const float B[NL] = {0.0247f, 0.0616f, 0.0779f, 0.0498f, 0.0268f};
    memcpy(c->fpam_K, K, sizeof K); memcpy(c->fpam_C, C, sizeof C);
    memcpy(c->fpam_b, b, sizeof b); memcpy(c->fpam_B, B, sizeof B);
    return vine_config_set_obs_type(c, VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO, 1);
}

int vine_config_set_obs_type(VineConfig* c, int obs_type, int scale_observations) {
    if (!c) return fail(VINE_ERR_INVALID_ARG, "cfg is NULL");
    // per-column scales of the two observation layouts the reference can scale (V5:246-266)
    const float joint_pos[6] = {0.12f, 0.269f, 0.148f, 0.249f, 0.148f, 0.344f};
    const float joint_vel[6] = {0.67f, 2.22f, 1.47f, 1.14f, 0.903f, 0.716f};
    const float tail[16] = {0.0656f, 0.238f, 0.0656f, 0.732f, 2.0f, 0.732f, 0.02f, 0.0235f,
                            0.02f, 0.732f, 2.0f, 0.732f, 0.845f, 0.86f, 0.0385f, 0.5f};
    if (obs_type < VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO || obs_type > VINE_OBS_POS_AND_PREV_POS)
        return fail(VINE_ERR_INVALID_ARG, "unknown observation type");
    const bool scalable = obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO || obs_type == VINE_OBS_TIP_AND_CART_AND_OBJ_INFO;
    if (scale_observations && !scalable)   // the reference raises NotImplementedError here (V5:267-268)
        return fail(VINE_ERR_UNSUPPORTED, "observation scaling not implemented for this observation type");
    c->obs_type = obs_type;
    for (int i = 0; i < VINE_MAX_OBS; ++i) c->obs_scaling[i] = 1.0f;
    if (scale_observations) {
        c->flags |= VINE_FLAG_SCALE_OBSERVATIONS;
        int k = 0;
        const int nj = (obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) ? 6 : 1;
        for (int i = 0; i < nj; ++i) c->obs_scaling[k++] = joint_pos[i];
        for (int i = 0; i < nj; ++i) c->obs_scaling[k++] = joint_vel[i];
        for (int i = 0; i < 16; ++i) c->obs_scaling[k++] = tail[i];
    } else {
        c->flags &= ~(uint32_t)VINE_FLAG_SCALE_OBSERVATIONS;
    }
    return VINE_OK;
}

int vine_num_obs(const VineConfig* c) {
    if (!c) return fail(VINE_ERR_INVALID_ARG, "cfg is NULL");
    if (c->obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) return 28;
    if (c->obs_type == VINE_OBS_TIP_AND_CART_AND_OBJ_INFO) return 18;
    if (c->obs_type == VINE_OBS_POS_ONLY) return 14;
    if (c->obs_type >= VINE_OBS_POS_AND_VEL && c->obs_type <= VINE_OBS_POS_AND_PREV_POS) return 26;
    return fail(VINE_ERR_INVALID_ARG, "unknown observation type");
}

#ifndef VSQ_THREADS
#define VSQ_THREADS 256      // threads per workgroup of the four-lane step kernel (A/B: 64 / 128)
#endif
static int step_grid_log2(const VineHandle* h);

int vine_create(const VineConfig* cfg, int device_id, float* state_storage, VineHandle** out) {
    int rc = validate(cfg);
    if (rc) return rc;
    if (!out) return fail(VINE_ERR_INVALID_ARG, "out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(VINE_ERR_NO_DEVICE, "no HIP device: libvine_hip has no CPU path (MI355X/gfx950 only)");
    if (device_id < 0 || device_id >= ndev) return fail(VINE_ERR_INVALID_ARG, "device_id out of range");
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail(VINE_ERR_DEVICE, "hipSetDevice failed");
    VineHandle* h = new (std::nothrow) VineHandle();
    if (!h) return fail(VINE_ERR_ALLOC, "out of host memory");
    h->cfg = *cfg;
    h->device = device_id;
    make_params(*cfg, h->P);
    h->reset_values = nullptr;
    h->reward_matrix = nullptr;
    h->stats_psum = nullptr;
    h->stats_pmax = nullptr;
    h->step_kernel = 0;
    h->refresh_body = false;
    if (const char* k = getenv("VINE_STEP_KERNEL")) h->step_kernel = !strcmp(k, "lane") ? 1 : (!strcmp(k, "quad") ? 2 : 0);
    const size_t bytes = (size_t)VF_COUNT * cfg->num_envs * sizeof(float);
    if (state_storage) {
        h->state = state_storage;
        h->owns_state = false;
    } else {
        hipError_t e = hipMalloc(&h->state, bytes);
        if (e != hipSuccess) { delete h; return hip_fail(e, "hipMalloc(state)"); }
        h->owns_state = true;
    }
    hipError_t e = hipMalloc(&h->counters, 2 * sizeof(unsigned long long));
    if (e != hipSuccess) { if (h->owns_state) (void)hipFree(h->state); delete h; return hip_fail(e, "hipMalloc(counters)"); }
    HIP_TRY(hipMemset(h->state, 0, bytes));
    HIP_TRY(hipMemset(h->counters, 0, 2 * sizeof(unsigned long long)));
    h->P.glog = step_grid_log2(h);
    const int threads = 256, blocks = (cfg->num_envs + threads - 1) / threads;
    hipLaunchKernelGGL(vine_init_kernel, dim3(blocks), dim3(threads), 0, 0, h->P, h->state);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    *out = h;
    return VINE_OK;
}

void vine_destroy(VineHandle* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    if (h->owns_state) (void)hipFree(h->state);
    (void)hipFree(h->counters);
    if (h->stats_psum) (void)hipFree(h->stats_psum);
    if (h->stats_pmax) (void)hipFree(h->stats_pmax);
    delete h;
}

// Four lanes per env (vine_step_quad_kernel) where the chip would otherwise be three quarters empty, for the
// configurations that kernel covers.  Measured (profiles/r02/step_kernels.txt): 4096 envs 30.7 -> 23.6 us, 16384 envs
// 32.5 -> 30.2 us, 32768 envs 35.3 -> 43.4 us: up to 16384 envs the quad kernel, beyond one lane per env.
static bool use_quad_kernel(const VineHandle* h) {
    const int obst = ((h->P.flags & VINE_FLAG_CREATE_SHELF) ? 1 : 0) | ((h->P.flags & VINE_FLAG_CREATE_PIPE) ? 2 : 0);
    (void)obst;      // (obstacles are covered since round 3: lane t evaluates link t's contacts)
    const bool quad_ok = h->P.cfi == 4 && (h->P.flags & VINE_FLAG_IMPLICIT_JOINT_DAMPING) && h->P.kq == 0.0f &&
                         h->P.cad == 0.0f && (h->P.obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO ||
                                              h->P.obs_type == VINE_OBS_TIP_AND_CART_AND_OBJ_INFO);
    return quad_ok && (h->step_kernel == 2 || (h->step_kernel == 0 && h->P.n <= 16384));
}

// log2 of the step launch's grid: the workgroups the kernel needs, rounded up to a power of two (see step_of(); the
// workgroups past the last env find no live lane and only report their arrival)
static int step_grid_log2(const VineHandle* h) {
    const long long blocks = use_quad_kernel(h) ? ((long long)h->P.n * 4 + VSQ_THREADS - 1) / VSQ_THREADS
                                                : ((long long)h->P.n + VINE_STEP_THREADS - 1) / VINE_STEP_THREADS;
    int l = 0;
    while ((1ll << l) < blocks) ++l;
    return l;
}

const char* vine_step_kernel_name(VineHandle* h) {
    if (!h) return "";
    return use_quad_kernel(h) ? "vine_step_quad_kernel" : "vine_step_kernel";
}

int vine_step(VineHandle* h, const float* actions, float* obs, float* rew, int64_t* reset, int64_t* progress,
              uint8_t* timeouts, void* stream) {
    if (!h || !actions || !obs || !rew || !reset || !progress || !timeouts)
        return fail(VINE_ERR_INVALID_ARG, "null argument to vine_step");
    DeviceGuard guard(h->device);
    const int threads = VINE_STEP_THREADS;
    hipStream_t s = (hipStream_t)stream;
    if (step_grid_log2(h) != h->P.glog) {
        // (the kernel choice changed since the counters were last normalised -- it cannot with the present switches, all of
        // which are fixed at creation: re-base the step count on the new grid size)
        const int64_t now = vine_get_step_count(h);
        if (now < 0) return fail(VINE_ERR_DEVICE, "step count unreadable");
        h->P.glog = step_grid_log2(h);
        const int rc = vine_set_step_count(h, now);
        if (rc != VINE_OK) return rc;
    }
    const int blocks = 1 << h->P.glog;
    const bool rnd = (h->P.flags & VINE_FLAG_VINE_RANDOMIZE) != 0;
    const int obst = ((h->P.flags & VINE_FLAG_CREATE_SHELF) ? 1 : 0) | ((h->P.flags & VINE_FLAG_CREATE_PIPE) ? 2 : 0);
    if (h->refresh_body) {
        h->refresh_body = false;
        hipLaunchKernelGGL(vine_refresh_body_kernel, dim3((h->P.n + 255) / 256), dim3(256), 0, s, h->P, h->state,
                           (const long long*)progress);
    }
    if (use_quad_kernel(h)) {
        const int qblocks = 1 << h->P.glog;
#define LAUNCH_QUAD_O(OT, RND, OB)                                                                                         \
    hipLaunchKernelGGL((vine_step_quad_kernel<OT, RND, OB, false>), dim3(qblocks), dim3(VSQ_THREADS), 0, s, h->P, h->state, actions, obs, rew, \
                       (long long*)reset, (long long*)progress, (unsigned char*)timeouts, h->reward_matrix,              \
                       h->reset_values, h->counters, RollArgs{})
#define LAUNCH_QUAD(OT, RND)                        \
    do {                                            \
        if (obst == 0) LAUNCH_QUAD_O(OT, RND, 0);   \
        else if (obst == 1) LAUNCH_QUAD_O(OT, RND, 1); \
        else if (obst == 2) LAUNCH_QUAD_O(OT, RND, 2); \
        else LAUNCH_QUAD_O(OT, RND, 3);             \
    } while (0)
        if (h->P.obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) {
            if (rnd) LAUNCH_QUAD(VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO, true);
            else LAUNCH_QUAD(VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO, false);
        } else {
            if (rnd) LAUNCH_QUAD(VINE_OBS_TIP_AND_CART_AND_OBJ_INFO, true);
            else LAUNCH_QUAD(VINE_OBS_TIP_AND_CART_AND_OBJ_INFO, false);
        }
#undef LAUNCH_QUAD
#undef LAUNCH_QUAD_O
        HIP_TRY(hipGetLastError());
        return VINE_OK;
    }
#define LAUNCH(OT, RND, SH)                                                                                      \
    hipLaunchKernelGGL((vine_step_kernel<OT, RND, SH>), dim3(blocks), dim3(threads), 0, s, h->P, h->state, actions, \
                       obs, rew, (long long*)reset, (long long*)progress, (unsigned char*)timeouts,                 \
                       h->reward_matrix, h->reset_values, h->counters)
#define LAUNCH_RND(OT, RND)                  \
    do {                                     \
        if (obst == 0) LAUNCH(OT, RND, 0);   \
        else if (obst == 1) LAUNCH(OT, RND, 1); \
        else if (obst == 2) LAUNCH(OT, RND, 2); \
        else LAUNCH(OT, RND, 3);             \
    } while (0)
#define LAUNCH_OT(OT)                        \
    do {                                     \
        if (rnd) LAUNCH_RND(OT, true);       \
        else LAUNCH_RND(OT, false);          \
    } while (0)
    if (h->P.obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) LAUNCH_OT(VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO);
    else if (h->P.obs_type == VINE_OBS_TIP_AND_CART_AND_OBJ_INFO) LAUNCH_OT(VINE_OBS_TIP_AND_CART_AND_OBJ_INFO);
    else if (h->P.obs_type == VINE_OBS_POS_ONLY) LAUNCH_OT(VINE_OBS_POS_ONLY);
    else LAUNCH_OT(VINE_OBS_POS_AND_VEL);    // the 26-column family, resolved inside the kernel
#undef LAUNCH_OT
#undef LAUNCH_RND
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return VINE_OK;
}

// ---- vine_step_rollout: policy head + step + rollout bookkeeping in one launch of the four-lane kernel (include/vine.h)
__global__ __launch_bounds__(256) void rollout_head_prep_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ w_mu, const float* __restrict__ b_mu,
                                                                const float* __restrict__ w_v, const float* __restrict__ b_v,
                                                                float* __restrict__ hw, float* __restrict__ hc) {
    // hw[k][u] = gamma_u w_k[u], hc[k] = sum_u beta_u w_k[u] + b_k  (k = mu_0, mu_1, value; 256 units, one per thread)
    __shared__ float red[3][4];
    const int u = threadIdx.x;
    const float g = gamma[u], b = beta[u];
    const float w[3] = {w_mu[u], w_mu[256 + u], w_v[u]};
    float part[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        hw[k * 256 + u] = g * w[k];
        float v = b * w[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        part[k] = v;
    }
    if ((u & 63) == 0) { red[0][u >> 6] = part[0]; red[1][u >> 6] = part[1]; red[2][u >> 6] = part[2]; }
    __syncthreads();
    if (u < 3) hc[u] = ((red[u][0] + red[u][1]) + (red[u][2] + red[u][3])) + (u < 2 ? b_mu[u] : b_v[0]);
}

int vine_rollout_head_prep(const float* ln_gamma, const float* ln_beta, const float* w_mu, const float* b_mu, const float* w_v,
                           const float* b_v, float* hw, float* hc, void* stream) {
    if (!ln_gamma || !ln_beta || !w_mu || !b_mu || !w_v || !b_v || !hw || !hc) return fail(VINE_ERR_INVALID_ARG, "null argument to vine_rollout_head_prep");
    hipLaunchKernelGGL(rollout_head_prep_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ln_gamma, ln_beta, w_mu, b_mu, w_v, b_v,
                       hw, hc);
    HIP_TRY(hipGetLastError());
    return VINE_OK;
}

int32_t vine_step_rollout_args_size(void) { return (int32_t)sizeof(VineRolloutArgs); }

int32_t vine_step_rollout_blocks(VineHandle* h) {
    if (!h || !use_quad_kernel(h)) return 0;
    return (int32_t)(((long long)h->P.n * 4 + VSQ_THREADS - 1) / VSQ_THREADS);      // the workgroups that own envs
}

int vine_step_rollout(VineHandle* h, const VineRolloutArgs* a, float* obs, float* rew, int64_t* reset, int64_t* progress,
                      uint8_t* timeouts, void* stream) {
    if (!h || !a || !obs || !rew || !reset || !progress || !timeouts || !a->y || !a->hw || !a->hc || !a->logstd || !a->counter ||
        !a->mu_out || !a->sigma_out || !a->value_out || !a->action_out || !a->neglogp_out || !a->shaped_out || !a->dones_out ||
        !a->cur_rewards || !a->cur_lengths || !a->h_state || !a->c_state || !a->partial ||
        ((a->value_mean == nullptr) != (a->value_var == nullptr)) || (a->h_op && a->h_op_stride < 256))
        return fail(VINE_ERR_INVALID_ARG, "null or inconsistent argument to vine_step_rollout");
    if (((uintptr_t)a->y | (uintptr_t)a->hw | (uintptr_t)a->h_state | (uintptr_t)a->c_state | (uintptr_t)a->h_op) & 15 ||
        ((uintptr_t)a->mu_out | (uintptr_t)a->sigma_out | (uintptr_t)a->action_out) & 7 || (a->h_op && (a->h_op_stride & 3)))
        return fail(VINE_ERR_INVALID_ARG, "vine_step_rollout: misaligned row pointer");
    if (!use_quad_kernel(h)) return fail(VINE_ERR_UNSUPPORTED, "vine_step_rollout needs the four-lanes-per-env step kernel");
    DeviceGuard guard(h->device);
    hipStream_t s = (hipStream_t)stream;
    if (step_grid_log2(h) != h->P.glog) {
        const int64_t now = vine_get_step_count(h);
        if (now < 0) return fail(VINE_ERR_DEVICE, "step count unreadable");
        h->P.glog = step_grid_log2(h);
        const int rc = vine_set_step_count(h, now);
        if (rc != VINE_OK) return rc;
    }
    if (h->refresh_body) {
        h->refresh_body = false;
        hipLaunchKernelGGL(vine_refresh_body_kernel, dim3((h->P.n + 255) / 256), dim3(256), 0, s, h->P, h->state,
                           (const long long*)progress);
    }
    RollArgs R;
    R.y = a->y; R.hw = a->hw; R.hc = a->hc; R.logstd = a->logstd; R.vmean = a->value_mean; R.vvar = a->value_var;
    R.ln_eps = a->ln_eps; R.veps = a->value_eps; R.seed_lo = (unsigned)a->seed; R.seed_hi = (unsigned)(a->seed >> 32);
    R.counter = (const long long*)a->counter; R.mu_out = a->mu_out; R.sigma_out = a->sigma_out; R.value_out = a->value_out;
    R.action_out = a->action_out; R.neglogp_out = a->neglogp_out; R.shift = a->reward_shift; R.scale = a->reward_scale;
    R.gamma_b = a->gamma_bootstrap; R.shaped = a->shaped_out; R.dones = a->dones_out; R.cur_r = a->cur_rewards;
    R.cur_l = a->cur_lengths; R.h_state = a->h_state; R.c_state = a->c_state; R.h_op = a->h_op; R.h_op_stride = a->h_op_stride;
    R.partial = a->partial;
    const int qblocks = 1 << h->P.glog;
    const bool rnd = (h->P.flags & VINE_FLAG_VINE_RANDOMIZE) != 0;
    const int obst = ((h->P.flags & VINE_FLAG_CREATE_SHELF) ? 1 : 0) | ((h->P.flags & VINE_FLAG_CREATE_PIPE) ? 2 : 0);
#define LAUNCH_ROLL_O(OT, RND, OB)                                                                                         \
    hipLaunchKernelGGL((vine_step_quad_kernel<OT, RND, OB, true>), dim3(qblocks), dim3(VSQ_THREADS), 0, s, h->P, h->state, (const float*)nullptr, obs, rew, \
                       (long long*)reset, (long long*)progress, (unsigned char*)timeouts, h->reward_matrix,              \
                       h->reset_values, h->counters, R)
#define LAUNCH_ROLL(OT, RND)                        \
    do {                                            \
        if (obst == 0) LAUNCH_ROLL_O(OT, RND, 0);   \
        else if (obst == 1) LAUNCH_ROLL_O(OT, RND, 1); \
        else if (obst == 2) LAUNCH_ROLL_O(OT, RND, 2); \
        else LAUNCH_ROLL_O(OT, RND, 3);             \
    } while (0)
    if (h->P.obs_type == VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO) {
        if (rnd) LAUNCH_ROLL(VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO, true);
        else LAUNCH_ROLL(VINE_OBS_POS_AND_FD_VEL_AND_OBJ_INFO, false);
    } else {
        if (rnd) LAUNCH_ROLL(VINE_OBS_TIP_AND_CART_AND_OBJ_INFO, true);
        else LAUNCH_ROLL(VINE_OBS_TIP_AND_CART_AND_OBJ_INFO, false);
    }
#undef LAUNCH_ROLL
#undef LAUNCH_ROLL_O
    HIP_TRY(hipGetLastError());
    return VINE_OK;
}


int vine_reset_idx(VineHandle* h, const int64_t* env_ids, int64_t n, float* rew, int64_t* reset, int64_t* progress,
                   void* stream) {
    if (!h || (!env_ids && n > 0)) return fail(VINE_ERR_INVALID_ARG, "null argument to vine_reset_idx");
    if (n <= 0) return VINE_OK;
    DeviceGuard guard(h->device);
    const int threads = 256;
    const int blocks = (int)((n + threads - 1) / threads);
    hipLaunchKernelGGL(vine_reset_idx_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, h->P, h->state,
                       (const long long*)env_ids, (long long)n, rew, (long long*)reset, (long long*)progress,
                       h->reset_values, h->counters);
    HIP_TRY(hipGetLastError());
    return VINE_OK;
}

int vine_bind_reset_values(VineHandle* h, const float* values) {
    if (!h) return fail(VINE_ERR_INVALID_ARG, "handle is NULL");
    h->reset_values = values;
    return VINE_OK;
}

int vine_bind_reward_matrix(VineHandle* h, float* reward_matrix) {
    if (!h) return fail(VINE_ERR_INVALID_ARG, "handle is NULL");
    h->reward_matrix = reward_matrix;
    if (reward_matrix) {                                        // whoever asks for the reward matrix reads the dashboard fields too
        if (!(h->P.flags & VINE_FLAG_INTROSPECT)) h->refresh_body = true;
        h->P.flags |= VINE_FLAG_INTROSPECT;
    }
    return VINE_OK;
}

int vine_set_introspection(VineHandle* h, int on) {
    if (!h) return fail(VINE_ERR_INVALID_ARG, "handle is NULL");
    if (on && !(h->P.flags & VINE_FLAG_INTROSPECT)) h->refresh_body = true;
    if (on) h->P.flags |= VINE_FLAG_INTROSPECT;
    else { h->P.flags &= ~(unsigned)VINE_FLAG_INTROSPECT; h->refresh_body = false; }
    return VINE_OK;
}

int vine_stats(VineHandle* h, const float* rew, const int64_t* progress, int64_t index_to_view, float* out, void* stream) {
    if (!h || !rew || !progress || !out || index_to_view < 0 || index_to_view >= h->P.n)
        return fail(VINE_ERR_INVALID_ARG, "bad argument to vine_stats");
    if (!(h->P.flags & VINE_FLAG_INTROSPECT))
        return fail(VINE_ERR_INVALID_ARG, "vine_stats needs VINE_FLAG_INTROSPECT (vine_set_introspection / vine_bind_reward_matrix) "
                                          "armed before the step whose state it summarises");
    DeviceGuard guard(h->device);
    if (!h->stats_psum) {
        HIP_TRY(hipMalloc(&h->stats_psum, sizeof(double) * STATS_BLOCKS * (STATS_NSUM + 1)));
        HIP_TRY(hipMalloc(&h->stats_pmax, sizeof(float) * STATS_BLOCKS * STATS_NMAX));
    }
    hipStream_t s = (hipStream_t)stream;
    int blocks = (h->P.n + 255) / 256;
    if (blocks > STATS_BLOCKS) blocks = STATS_BLOCKS;
    hipLaunchKernelGGL(vine_stats_partial_kernel, dim3(blocks), dim3(256), 0, s, h->P, h->state, rew, (const long long*)progress,
                       h->reward_matrix, h->stats_psum, h->stats_pmax);
    hipLaunchKernelGGL(vine_stats_finalize_kernel, dim3(1), dim3(64), 0, s, h->P, h->state, blocks, h->stats_psum, h->stats_pmax,
                       h->reward_matrix ? 1 : 0, (long long)index_to_view, out);
    HIP_TRY(hipGetLastError());
    return VINE_OK;
}

float* vine_state_ptr(VineHandle* h) { return h ? h->state : nullptr; }

int64_t vine_get_step_count(VineHandle* h) {
    if (!h) return -1;
    DeviceGuard guard(h->device);
    unsigned long long v[2] = {0ull, 0ull};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(v, h->counters, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)(v[0] + (v[1] >> h->P.glog));      // (step_of)
}

int vine_set_step_count(VineHandle* h, int64_t step_count) {
    if (!h || step_count < 0) return fail(VINE_ERR_INVALID_ARG, "bad step count");
    DeviceGuard guard(h->device);
    unsigned long long v[2] = {(unsigned long long)step_count, 0ull};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->counters, v, sizeof v, hipMemcpyHostToDevice));
    return VINE_OK;
}

#ifdef VSQ_TIMING
int vine_debug_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vsq_t), sizeof(unsigned long long) * 1024 * 8) == hipSuccess ? 0 : -1;
}
#endif
}  // extern "C"
