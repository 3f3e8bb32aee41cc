"""Row P1: the reference's physics is PhysX (closed, absent) => parity unpinned.  These tests pin the
oracle's rigid-body model against itself: three independent formulations, conservation laws and
closed-form small-motion results derived from the URDF constants."""
import ctypes as C

import numpy as np
import pytest

from oracle import vine_oracle as vo
from vine_robot_isaacgymenvs_amd import abi

H = 0.00833 / 10


def cfg_with(implicit=True, damping=None, cad=0.0):
    cfg = vo.default_config()
    cfg.set_flag(abi.FLAG_IMPLICIT_JOINT_DAMPING, implicit)
    if damping is not None:
        cfg.damping = damping
    cfg.link_angular_damping = cad
    return cfg


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    lib = vo.load("f64")
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, expect in kat:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        lib.vine_oracle_philox(c, k, o)
        assert list(o) == expect


@pytest.mark.parametrize("implicit", [False, True])
def test_three_formulations_agree(implicit):
    """Jacobian CRBA (relative coords), planar ABA, absolute-angle Lagrangian: same accelerations."""
    cfg = cfg_with(implicit)
    rng = np.random.default_rng(0)
    for _ in range(300):
        q = rng.uniform(-1.0, 1.0, 6)
        qd = rng.uniform(-5, 5, 6)
        eff = rng.uniform(-0.5, 0.5, 6)
        cj = rng.uniform(0.01, 0.08, 6)
        a = [vo.forward_dynamics(cfg, q, qd, eff, H, f, cj=cj) for f in (vo.FORM_CRBA, vo.FORM_ABA, vo.FORM_ABS)]
        scale = 1 + np.abs(a[0]).max()
        assert np.abs(a[0] - a[1]).max() / scale < 1e-10
        assert np.abs(a[0] - a[2]).max() / scale < 1e-10


def test_link_angular_damping_consistent():
    """The optional per-link angular damping switch: CRBA and absolute-angle forms agree."""
    for implicit in (False, True):
        cfg = cfg_with(implicit, cad=0.5)
        rng = np.random.default_rng(1)
        for _ in range(50):
            q, qd, eff = rng.uniform(-1, 1, 6), rng.uniform(-5, 5, 6), rng.uniform(-0.5, 0.5, 6)
            a0 = vo.forward_dynamics(cfg, q, qd, eff, H, vo.FORM_CRBA)
            a2 = vo.forward_dynamics(cfg, q, qd, eff, H, vo.FORM_ABS)
            assert np.abs(a0 - a2).max() / (1 + np.abs(a0).max()) < 1e-10


def test_f32_matches_f64():
    cfg = cfg_with(True)
    rng = np.random.default_rng(2)
    worst = 0
    for _ in range(200):
        q, qd, eff = rng.uniform(-0.6, 0.6, 6), rng.uniform(-3, 3, 6), rng.uniform(-0.3, 0.3, 6)
        a64 = vo.forward_dynamics(cfg, q, qd, eff, H, vo.FORM_ABS, "f64")
        a32 = vo.forward_dynamics(cfg, q, qd, eff, H, vo.FORM_ABS, "f32")
        worst = max(worst, np.abs(a64 - a32).max() / (1 + np.abs(a64).max()))
    assert worst < 5e-3      # M is ill-conditioned (5 g links carrying a 100 g link): fp32 solve, cond ~1e3-1e4


def test_energy_conservation_undamped():
    """tau = 0, damping = 0: semi-implicit Euler keeps the energy error O(h), with no drift."""
    cfg = cfg_with(False, damping=0.0)
    q = np.array([0.05, 0.3, -0.2, 0.25, -0.1, 0.2])
    qd = np.zeros(6)
    e0 = vo.energy(cfg, q, qd)
    swing = abs(vo.energy(cfg, q, qd) - vo.energy(cfg, np.zeros(6), qd))
    errs = []
    for h in (H / 8, H / 16):
        qq, qqd = q.copy(), qd.copy()
        worst = 0
        n = int(round(0.5 / h))
        for _ in range(20):
            qq, qqd = vo.simulate(cfg, qq, qqd, np.zeros(6), h, n // 20, vo.FORM_CRBA)
            worst = max(worst, abs(vo.energy(cfg, qq, qqd) - e0))
        errs.append(worst)
    assert errs[0] < 0.05 * swing
    assert errs[1] < 0.7 * errs[0]          # first-order convergence


def test_momentum_of_free_cart():
    """No rail force, no damping: horizontal momentum of cart + links is conserved (to O(h))."""
    cfg = cfg_with(False, damping=0.0)
    q = np.array([0.0, 0.4, 0.1, -0.3, 0.2, 0.1])
    qd = np.array([0.3, 0, 0, 0, 0, 0])

    def py(q, qd):   # d(E)/d(vy) at fixed rest = horizontal momentum; finite difference of the energy
        eps = 1e-6
        return (vo.energy(cfg, q, qd + np.array([eps, 0, 0, 0, 0, 0])) - vo.energy(cfg, q, qd - np.array([eps, 0, 0, 0, 0, 0]))) / (2 * eps)

    p0 = py(q, qd)
    errs = []
    for div in (4, 16):      # the integrator is first order: the momentum error shrinks with h
        q1, qd1 = vo.simulate(cfg, q, qd, np.zeros(6), H / div, 500 * div, vo.FORM_ABS)
        errs.append(abs(py(q1, qd1) - p0))
    assert errs[0] < 2e-3 * abs(p0)
    assert errs[1] < 0.5 * errs[0]


def test_static_equilibrium_and_rest_pose():
    """phi0 = 3.1415 (not pi): the hanging equilibrium sits at q1 ~ +9.27e-5 and the rest tip at
    z = 1 - 0.025 - 0.01 - 5*0.0885 = 0.5225."""
    cfg = cfg_with(True)
    q = np.zeros(6)
    q[1] = np.pi - 3.1415
    a = vo.forward_dynamics(cfg, q, np.zeros(6), np.zeros(6), H, vo.FORM_CRBA)
    assert np.abs(a).max() < 1e-3          # float32 constants in VineConfig leave a tiny residual
    t = vo.tip(cfg, q, np.zeros(6))
    assert abs(t[1] - 0.5225) < 1e-6 and abs(t[0]) < 1e-6
    t0 = vo.tip(cfg, np.zeros(6), np.zeros(6))
    assert abs(t0[0] - (-0.4425 * np.sin(3.1415))) < 1e-6


def test_total_mass_and_cart_acceleration():
    """With all revolute joints locked by symmetry (straight chain hanging at equilibrium) a rail force F
    accelerates the cart by more than F/m_total (the chain lags) and less than F/m_cart."""
    cfg = cfg_with(False, damping=0.0)
    q = np.zeros(6)
    q[1] = np.pi - 3.1415
    a = vo.forward_dynamics(cfg, q, np.zeros(6), np.array([1.0, 0, 0, 0, 0, 0]), H, vo.FORM_ABA)
    assert 1.0 / 0.52 < a[0] < 1.0 / 0.4


def test_small_angle_pendulum_frequency():
    """Lock the model into a single rigid pendulum by making joints 2..5 very stiff: the swing frequency of
    joint 1 must match sqrt(m g r / I_pivot) of the composite body computed from the URDF constants."""
    cfg = cfg_with(False, damping=0.0)
    m = np.array(cfg.link_mass[:])
    inertia = np.array(cfg.link_inertia[:])
    L, l = cfg.link_length, cfg.link_com
    r = np.array([k * L + l for k in range(5)])
    mt, rc = m.sum(), (m * r).sum() / m.sum()
    ip = (inertia + m * r ** 2).sum()
    omega = np.sqrt(mt * 9.81 * rc / ip)
    # analytic check through the mass matrix: d2q1/dt2 for a small straight-chain deflection with a fixed cart
    eps = 1e-4
    q = np.zeros(6)
    q[1] = (np.pi - 3.1415) + eps
    cfg.cart_mass = 1e6            # effectively fixed cart
    # a straight chain is not an eigenmode of the 5-link chain, so compare the energy Hessian instead:
    # potential energy of a rigid deflection eps must be 0.5 * (mt g rc) eps^2
    e0 = vo.energy(cfg, np.array([0, np.pi - 3.1415, 0, 0, 0, 0.0]), np.zeros(6))
    e1 = vo.energy(cfg, q, np.zeros(6))
    assert abs((e1 - e0) - 0.5 * mt * 9.81 * rc * eps ** 2) < 1e-3 * 0.5 * mt * 9.81 * rc * eps ** 2 + 1e-12
    # kinetic energy of a rigid rotation rate w about joint 1 must be 0.5 * ip * w^2
    w = 0.7
    ek = vo.energy(cfg, np.array([0, np.pi - 3.1415, 0, 0, 0, 0.0]), np.array([0, w, 0, 0, 0, 0.0])) - e0
    assert abs(ek - 0.5 * ip * w * w) < 1e-9
    assert 5.0 < omega < 7.0       # ~0.95 Hz: plausible for a 0.44 m hanging arm


def test_default_parameters_are_stable():
    """Random and bang-bang actions for 400 control steps stay bounded (the held-C variant does not)."""
    for mode in ("random", "bang"):
        cfg = vo.default_config(num_envs=32)
        cfg.set_flag(abi.FLAG_VINE_RANDOMIZE, False)
        env = vo.OracleEnv(cfg, "f64")
        rng = np.random.default_rng(3)
        a = np.zeros((32, 2))
        for s in range(400):
            if mode == "random":
                a = rng.uniform(-1, 1, (32, 2))
            elif s % 10 == 0:
                a = np.sign(rng.uniform(-1, 1, (32, 2)))
            env.step(a)
            st = env.state
            assert np.isfinite(st).all()
            assert np.abs(st[abi.VF_QD0 + 1:abi.VF_QD0 + 6]).max() < 30
            assert np.abs(st[abi.VF_Q0 + 1:abi.VF_Q0 + 6]).max() < 1.5


def test_held_velocity_feedback_is_unstable():
    """Documents assumption P4: with the reference's literal held C*qd torque the sampled
    system diverges in this integrator, which is why the default moves C into the DOF damping."""
    cfg = vo.default_config(num_envs=8)
    cfg.set_flag(abi.FLAG_VINE_RANDOMIZE, False)
    cfg.set_flag(abi.FLAG_FPAM_DAMPING_HELD, True)
    env = vo.OracleEnv(cfg, "f64")
    rng = np.random.default_rng(4)
    blew = False
    for s in range(50):
        env.step(rng.uniform(-1, 1, (8, 2)))
        if not np.isfinite(env.state).all() or np.abs(env.state[abi.VF_QD0 + 1:abi.VF_QD0 + 6]).max() > 1e3:
            blew = True
            break
    assert blew


def test_shelf_contact_model():
    """Config 5 contacts (solver unpinned): zero away from the shelf; the strip force is
    reported only for front-edge contacts; generalised forces equal J^T F (checked by virtual work)."""
    import ctypes as C
    lib = vo.load("f64")
    cfg = vo.default_config()
    D = C.POINTER(C.c_double)

    def contact(q, qd, sy, sz):
        q, qd = np.ascontiguousarray(q, np.float64), np.ascontiguousarray(qd, np.float64)
        Q = np.zeros(6)
        f = lib.vine_oracle_shelf_contact(C.byref(cfg), q.ctypes.data_as(D), qd.ctypes.data_as(D), sy, sz,
                                          Q.ctypes.data_as(D))
        return f, Q

    q0 = np.zeros(6)
    f, Q = contact(q0, np.zeros(6), 0.2 + 1.0, 0.0)          # shelf far away
    assert f == 0 and not Q.any()
    # hanging chain, tip at (0, 0.5225): put the strip's front corner 5 mm inside the last link, on its -y side
    t = vo.tip(cfg, q0, np.zeros(6))
    sy = t[0] - 0.2 - 0.0719 + 0.005        # front face x = shelf_y + 0.2 sits 5 mm inside the FPAM-side face (world -y)
    sz = t[1] + 0.03
    f, Q = contact(q0, np.zeros(6), sy, sz)
    assert f > 0                                              # strip is touched
    assert abs(f - 2 * 2000.0 * 0.005) < 0.5                  # two corners, 5 mm deep, k = 2000 N/m
    assert Q[0] > 0                                           # the link (and the cart) is pushed towards +y, out of the shelf
    # virtual work: Q . dq == sum F . dx ; here all force is along y on link 5: Q0 = Fy_total = f
    assert abs(Q[0] - f) < 1e-5 * f                           # phi0 = 3.1415 tilts the face normal by 9e-5 rad
    # moment about joint 1 = lever arm (vertical distance) x force
    lever = (0.965 - sz)
    assert abs(abs(Q[1]) - f * lever) < 0.02 * f * lever
    # board contact alone does not touch the strip: lay the tip corner on the top of board A
    sy2, sz2 = t[0] - 0.1, t[1] - 0.003                       # board top = sz2 + 0.005 is 2 mm above the link's end
    f2, Q2 = contact(q0, np.zeros(6), sy2, sz2)
    assert f2 == 0 and Q2.any()


def test_shelf_run_is_stable():
    cfg = vo.default_config(num_envs=64)
    cfg.set_flag(abi.FLAG_CREATE_SHELF, True)
    cfg.set_flag(abi.FLAG_VINE_RANDOMIZE, False)
    env = vo.OracleEnv(cfg, "f64")
    rng = np.random.default_rng(0)
    touched = 0
    a = np.zeros((64, 2))
    for s in range(300):
        if s % 8 == 0:
            a = np.sign(rng.uniform(-1, 1, (64, 2)))
        env.step(a)
        assert np.isfinite(env.state).all()
        assert np.abs(env.state[abi.VF_QD0 + 1:abi.VF_QD0 + 6]).max() < 60
        touched += int((env.state[abi.VF_CONTACT_MEAN] > 0).sum())
    assert touched > 0


def test_pipe_contact_model():
    """CREATE_PIPE cross-section (two walls in the pipe frame): no force away from the pipe; a link pressed into a
    wall is pushed back along the wall normal; the run with the obstacle stays bounded."""
    import ctypes as C
    lib = vo.load("f64")
    cfg = vo.default_config()
    D = C.POINTER(C.c_double)

    def contact(q, qd, py, pz, tp):
        q, qd = np.ascontiguousarray(q, np.float64), np.ascontiguousarray(qd, np.float64)
        Q = np.zeros(6)
        lib.vine_oracle_pipe_contact(C.byref(cfg), q.ctypes.data_as(D), qd.ctypes.data_as(D), py, pz, tp,
                                     Q.ctypes.data_as(D))
        return Q

    q0 = np.zeros(6)
    assert not contact(q0, np.zeros(6), -2.0, 0.5, 0.8).any()
    # theta' = -90 deg -> theta = 0: pipe frame == world frame; wall 1 occupies y in [py, py+0.00525], z in [pz, pz+0.34125].
    # Put it 1.5 mm inside the -y face of the hanging chain (face at y = -0.0719): it must push the chain towards +y
    # (walls are 5.25 mm thin: beyond half their thickness the nearest exit is the far side).
    t = vo.tip(cfg, q0, np.zeros(6))
    py = t[0] - 0.0719 - 0.00525 + 0.0015
    Q = contact(q0, np.zeros(6), py, 0.55, -np.pi / 2)
    assert Q[0] > 0 and Q[1] != 0
    # and the mirror image with the second wall on the +y face (face at y = +0.0381) pushes towards -y
    py2 = t[0] + 0.0381 - 0.0015 - (0.1554 - 0.00525)
    Q2 = contact(q0, np.zeros(6), py2, 0.55, -np.pi / 2)
    assert Q2[0] < 0
    cfg2 = vo.default_config(num_envs=64)
    cfg2.set_flag(abi.FLAG_CREATE_PIPE, True)
    cfg2.set_flag(abi.FLAG_VINE_RANDOMIZE, False)
    env = vo.OracleEnv(cfg2, "f64")
    rng = np.random.default_rng(0)
    a = np.zeros((64, 2))
    for s in range(300):
        if s % 8 == 0:
            a = np.sign(rng.uniform(-1, 1, (64, 2)))
        env.step(a)
        assert np.isfinite(env.state).all() and np.abs(env.state[abi.VF_QD0 + 1:abi.VF_QD0 + 6]).max() < 80
    assert np.abs(env.state[abi.VF_OBJ_ANGLE]).max() < 1.3 and (env.state[abi.VF_OBJ_DEPTH] != 0).all()


# --------------------------------------------------------------------------- literal-mode switch sweep
# The reference's own configuration -- the FPAM velocity term C_j*qd_j held over the 8.33 ms sim step (V5:1062) with
# DAMPING 0.02 (TY:49) -- diverges in this articulation model, while the authors trained with it in PhysX.  SURVEY 8(c)
# lists the Isaac Gym / PhysX defaults that cannot be verified here; each is a probe switch of the oracle.  The sweep
# records which single switch makes the literal configuration bounded, and compares the per-channel RMS of the
# unscaled observations under a random policy with the only physics data the reference holds: the empirical
# observation scales of V5:246-255 (joint angle 0.12-0.34 rad, finite-difference joint velocity 0.67-2.22 rad/s,
# tip y 0.238 m, finite-difference tip velocity 2.0 / 0.732 m/s).
OBS_SCALE_REF = np.array([0.12, 0.269, 0.148, 0.249, 0.148, 0.344, 0.67, 2.22, 1.47, 1.14, 0.903, 0.716])   # V5:246-249
TIP_SCALE_REF = np.array([0.238, 2.0, 0.732])     # tip y (V5:250), finite-difference tip velocity y, z (V5:251)

SWEEP_CASES = [
    # name, config changes, probe switches, expected class
    ("default: C_j in the implicit DOF damping (P4)", dict(), dict(), "stable"),
    ("literal: held C_j*qd_j, DAMPING 0.02", dict(held=True), dict(), "diverges"),
    ("literal + DAMPING 0.08 (round-1 fixtures)", dict(held=True, damping=0.08), dict(), "stable"),
    ("literal + link angular damping 0.5 1/s (AssetOptions default)", dict(held=True, cad=0.5), dict(), "diverges"),
    ("literal + explicit DOF damping", dict(held=True, explicit=True), dict(), "diverges"),
    ("literal + max link angular velocity 64 rad/s (AssetOptions default)", dict(held=True), dict(vmax_link=64.0), "chatters"),
    ("literal + max joint velocity 100 rad/s (PhysX joint default)", dict(held=True), dict(vmax_joint=100.0), "chatters"),
    ("literal + joint armature 1e-4 kg m^2", dict(held=True), dict(armature=1e-4), "diverges"),
    ("literal + joint armature 1e-3 kg m^2", dict(held=True), dict(armature=1e-3), "stable"),
    ("literal + efforts applied in the first substep only", dict(held=True), dict(effort_first_substep_only=True), "stable"),
    # Isaac Gym clamps set_dof_actuation_force_tensor to the DOF `effort` property; the URDF has no
    # <limit effort> (Vine5LinkMovingBase.urdf:278,292), so the value in force is the importer's default (unverifiable)
    ("literal + joint effort limit 0.05 N m", dict(held=True, effort_limit=0.05), dict(), "stable"),
    ("literal + joint effort limit 0.3 N m", dict(held=True, effort_limit=0.3), dict(), "stable"),
    ("literal + joint effort limit 1.0 N m", dict(held=True, effort_limit=1.0), dict(), "chatters"),
    ("literal + joint effort limit 5.0 N m", dict(held=True, effort_limit=5.0), dict(), "chatters"),
]


def run_sweep_case(cfg_changes, probe, n=48, steps=160, seed=11, precision="f64"):
    """Random-policy rollout through the oracle's full step; returns (class, max |qd|, RMS ratios) with the ratios =
    RMS of the unscaled observation channels / the reference's observation scale (12 joint channels, 3 tip channels)."""
    cfg = vo.default_config(num_envs=n)
    cfg.set_flag(abi.FLAG_VINE_RANDOMIZE, False)
    vo.load().vine_config_set_obs_type(cfg, abi.OBS_POS_AND_FD_VEL_AND_OBJ_INFO, 0)      # unscaled observations
    cfg.clip_observations = 1e9
    cfg.set_flag(abi.FLAG_FPAM_DAMPING_HELD, bool(cfg_changes.get("held", False)))
    cfg.set_flag(abi.FLAG_IMPLICIT_JOINT_DAMPING, not cfg_changes.get("explicit", False))
    cfg.damping = cfg_changes.get("damping", cfg.damping)
    cfg.link_angular_damping = cfg_changes.get("cad", 0.0)
    cfg.effort_limit = cfg_changes.get("effort_limit", 0.0)
    env = vo.OracleEnv(cfg, precision)
    env.set_probe(**probe)
    rng = np.random.default_rng(seed)
    rows, worst = [], 0.0
    for s in range(steps):
        fresh = env.reset_buf.astype(bool).copy()           # envs reset inside this step: their FD channels are meaningless
        obs, _, _, _ = env.step(rng.uniform(-1, 1, (n, 2)))
        st = env.state
        if not np.isfinite(st).all():
            return "diverges", np.inf, None
        worst = max(worst, float(np.abs(st[abi.VF_QD0 + 1:abi.VF_QD0 + 6]).max()))
        if worst > 1e3:
            return "diverges", worst, None
        if s >= steps // 4:
            rows.append(obs[~fresh].astype(np.float64))
    o = np.concatenate(rows, 0)
    rms = np.sqrt((o ** 2).mean(0))
    joint = rms[:12] / OBS_SCALE_REF
    tip = np.array([o[:, 13].std(), rms[16], rms[17]]) / TIP_SCALE_REF
    ratios = np.concatenate([joint, tip])
    cls = "chatters" if ratios[6:12].max() > 3.0 or worst > 40.0 else "stable"
    return cls, worst, ratios


def test_literal_mode_switch_sweep():
    """Which unverifiable simulator default makes the reference's literal configuration bounded (DESIGN.md section 3
    table).  None of the documented Isaac Gym / PhysX defaults does: the velocity clamps bound it, but as a chatter at
    several times the reference's own observation scales; joint armature >= ~1e-3 kg m^2 (not a reference setting: the
    asset properties printed at V5:560-583 show armature 0) or a 4x DAMPING do.  The default mode and every stable
    variant stay within a factor 10 of the reference's per-channel observation scales under a random policy."""
    table = []
    for name, changes, probe, expected in SWEEP_CASES:
        cls, worst, ratios = run_sweep_case(changes, probe)
        table.append((name, cls, worst, ratios))
        assert cls == expected, (name, cls, worst, ratios)
        if cls == "stable":
            assert 0.1 < ratios.min() and ratios.max() < 3.0, (name, ratios)
    # the effort clamp BOUNDS the literal mode but does not cure it: the peak joint rate of the bounded state grows in
    # proportion to the limit (a clamp-limited limit cycle), so no value of the limit is "the" missing simulator term,
    # and every limit that leaves the actuation its authority (K q + B u reaches ~0.45 N m) chatters
    peaks = [row[2] for row in table[10:14]]
    assert all(b > 1.3 * a for a, b in zip(peaks, peaks[1:])), peaks
    default, arm = table[0][3], table[8][3]
    # the stabilised literal mode and the default mode are the same visible dynamics (the heavy modes): per channel
    # within 25 % of each other
    assert np.abs(arm / default - 1.0).max() < 0.25, arm / default
    for name, cls, worst, ratios in table:
        print("%-70s %-9s max|qd| %8.1f  %s" % (name, cls, worst, "" if ratios is None else np.array2string(
            ratios, precision=2, suppress_small=True, max_line_width=200)))
