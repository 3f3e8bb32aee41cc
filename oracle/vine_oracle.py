"""ctypes binding of the CPU oracle (``oracle/vine_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from vine_robot_isaacgymenvs_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
FORM_CRBA, FORM_ABA, FORM_ABS = 0, 1, 2


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


_cache = {}
_D = C.POINTER(C.c_double)


def load(precision="f64", omp=False):
    key = (precision, omp, "" if omp else os.environ.get("VINE_ORACLE_VARIANT", ""))
    if key in _cache:
        return _cache[key]
    # VINE_ORACLE_VARIANT=_asan: the sanitizer build (oracle/Makefile `asan-test`); "_flops": the flop-counting build
    variant = "" if omp else os.environ.get("VINE_ORACLE_VARIANT", "")
    name = "libvine_oracle_%s%s%s.so" % (precision, "_omp" if omp else "", variant)
    path = os.path.join(_BUILD, name)
    if variant == "_asan":
        subprocess.check_call(["make", "-C", _HERE, "asan"], stdout=subprocess.DEVNULL)
    else:
        build()          # make: a no-op unless vine_oracle.c / include/vine.h changed (never run a stale checker)
    lib = abi.declare(C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0)))
    P = C.POINTER(abi.VineConfig)
    lib.vine_oracle_real_bytes.restype = C.c_int
    lib.vine_oracle_set_threads.argtypes = [C.c_int]
    lib.vine_oracle_set_threads.restype = C.c_int
    lib.vine_oracle_philox.argtypes = [C.POINTER(C.c_uint32)] * 3
    lib.vine_oracle_forward_dynamics.argtypes = [P, C.c_int, _D, _D, _D, _D, C.c_double, _D]
    lib.vine_oracle_forward_dynamics.restype = C.c_int
    lib.vine_oracle_simulate.argtypes = [P, C.c_int, _D, _D, _D, _D, C.c_double, C.c_int]
    lib.vine_oracle_simulate.restype = C.c_int
    lib.vine_oracle_simulate_obstacles.argtypes = [P, C.c_int, _D, _D, _D, _D, C.c_double, C.c_int, C.c_int, C.c_int, _D]
    lib.vine_oracle_simulate_obstacles.restype = C.c_double
    lib.vine_oracle_tip.argtypes = [P, _D, _D, _D]
    lib.vine_oracle_energy.argtypes = [P, _D, _D]
    lib.vine_oracle_energy.restype = C.c_double
    lib.vine_oracle_actions.argtypes = [P, C.c_double, C.c_double, _D, _D]
    lib.vine_oracle_smooth.argtypes = [P, C.c_double, C.c_double]
    lib.vine_oracle_smooth.restype = C.c_double
    lib.vine_oracle_actuation.argtypes = [P, _D, _D, C.c_double, C.c_double, C.c_double, _D, _D, _D, _D]
    lib.vine_oracle_observations.argtypes = [P, _D, _D, _D, _D, _D, C.c_double, C.c_double, _D, _D]
    lib.vine_oracle_observations.restype = C.c_int
    lib.vine_oracle_observations_ex.argtypes = [P, _D, _D, _D, _D, _D, _D, _D, C.c_double, C.c_double, _D, _D]
    lib.vine_oracle_observations_ex.restype = C.c_int
    lib.vine_oracle_reward.argtypes = ([P] + [C.c_double, C.c_int] + [C.c_double] * 6 +
                                       [C.c_int, C.c_int, C.c_double, C.c_double, _D])
    lib.vine_oracle_reward.restype = C.c_double
    lib.vine_oracle_reset_logic.argtypes = [P, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.vine_oracle_reset_logic.restype = C.c_int64
    lib.vine_oracle_shelf_contact.argtypes = [P, _D, _D, C.c_double, C.c_double, _D]
    lib.vine_oracle_shelf_contact.restype = C.c_double
    lib.vine_oracle_pipe_contact.argtypes = [P, _D, _D, C.c_double, C.c_double, C.c_double, _D]
    lib.vine_oracle_state.argtypes = [C.c_void_p]
    lib.vine_oracle_state.restype = C.c_void_p
    lib.vine_oracle_set_formulation.argtypes = [C.c_void_p, C.c_int]
    lib.vine_oracle_pull_mirror.argtypes = [C.c_void_p]
    lib.vine_oracle_set_probe.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int]
    _cache[key] = lib
    return lib


def flop_counts(num_envs=256, steps=20, randomize=True, seed=0):
    """Algorithmic flops per env step measured by the counting build of the oracle (``make -C oracle flops``): random
    actions through the full step; returns {"F_sub", "F_act", "F_post", "per_env_step"} (SURVEY 8d:
    40 F_sub + 4 F_act + F_post).  RNG draws, the contact tests and the resets are NOT counted (integer / branchy work)."""
    old = os.environ.get("VINE_ORACLE_VARIANT")
    os.environ["VINE_ORACLE_VARIANT"] = "_flops"
    try:
        lib = load("f32")
    finally:
        if old is None:
            os.environ.pop("VINE_ORACLE_VARIANT", None)
        else:
            os.environ["VINE_ORACLE_VARIANT"] = old
    lib.vine_oracle_flop_counters.argtypes = [_D, _D, C.c_int]
    cfg = default_config(lib, num_envs=num_envs)
    cfg.set_flag(abi.FLAG_VINE_RANDOMIZE, randomize)
    h = C.c_void_p()
    assert lib.vine_create(C.byref(cfg), -1, None, C.byref(h)) == 0
    n, nobs = num_envs, lib.vine_num_obs(C.byref(cfg))
    obs, rew = np.zeros((n, nobs), np.float32), np.zeros(n, np.float32)
    rst, prog, to = np.ones(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.uint8)
    rng = np.random.default_rng(seed)
    fl, calls = np.zeros(3), np.zeros(3)
    lib.vine_oracle_flop_counters(_dp(fl), _dp(calls), 1)
    for _ in range(steps):
        a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        lib.vine_step(h, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, rst.ctypes.data, prog.ctypes.data, to.ctypes.data, None)
    lib.vine_oracle_flop_counters(_dp(fl), _dp(calls), 1)
    lib.vine_destroy(h)
    f_sub, f_act, f_post = fl[0] / calls[0], fl[1] / calls[1], fl[2] / calls[2]
    return {"F_sub": f_sub, "F_act": f_act, "F_post": f_post,
            "per_env_step": cfg.control_freq_inv * cfg.substeps * f_sub + cfg.control_freq_inv * f_act + f_post,
            "substeps_per_env_step": calls[0] / calls[2], "note": "+, -, *, / and each sin/cos/sqrt counted as one; oracle's "
            "plain formulation (full mass matrix, sincos per substep); RNG, contacts and resets not counted"}


def default_config(lib=None, **overrides):
    lib = lib or load()
    cfg = abi.VineConfig()
    rc = lib.vine_config_default(C.byref(cfg))
    assert rc == 0
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def _dp(a):
    return a.ctypes.data_as(_D)


def forward_dynamics(cfg, q, qd, eff, h=0.0, form=FORM_ABS, precision="f64", cj=None):
    lib = load(precision)
    q, qd, eff = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, qd, eff))
    out = np.zeros(6)
    cjp = None if cj is None else _dp(np.ascontiguousarray(cj, dtype=np.float64))
    rc = lib.vine_oracle_forward_dynamics(C.byref(cfg), form, _dp(q), _dp(qd), _dp(eff), cjp, h, _dp(out))
    if rc:
        raise RuntimeError("forward dynamics failed rc=%d" % rc)
    return out


def simulate(cfg, q, qd, eff, h, n, form=FORM_ABS, precision="f64", cj=None):
    lib = load(precision)
    q = np.array(q, dtype=np.float64)
    qd = np.array(qd, dtype=np.float64)
    eff = np.ascontiguousarray(eff, dtype=np.float64)
    cjp = None if cj is None else _dp(np.ascontiguousarray(cj, dtype=np.float64))
    rc = lib.vine_oracle_simulate(C.byref(cfg), form, _dp(q), _dp(qd), _dp(eff), cjp, h, n)
    if rc:
        raise RuntimeError("simulate failed rc=%d" % rc)
    return q, qd


def simulate_obstacles(cfg, q, qd, eff, h, n, shelf, pipe, obstacle, form=FORM_ABS, precision="f64", cj=None):
    """One ``gym.simulate`` with the shelf / pipe contacts re-evaluated per substep (the env step's own inner loop);
    ``obstacle`` = (shelf_y, shelf_z, pipe_y, pipe_z, theta').  Returns (q, qd, reported shelf-strip contact force)."""
    lib = load(precision)
    q = np.array(q, dtype=np.float64)
    qd = np.array(qd, dtype=np.float64)
    eff = np.ascontiguousarray(eff, dtype=np.float64)
    ob = np.ascontiguousarray(obstacle, dtype=np.float64)
    cjp = None if cj is None else _dp(np.ascontiguousarray(cj, dtype=np.float64))
    contact = lib.vine_oracle_simulate_obstacles(C.byref(cfg), form, _dp(q), _dp(qd), _dp(eff), cjp, h, n, int(shelf),
                                                 int(pipe), _dp(ob))
    return q, qd, contact


def tip(cfg, q, qd, precision="f64"):
    lib = load(precision)
    q, qd = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, qd))
    out = np.zeros(4)
    lib.vine_oracle_tip(C.byref(cfg), _dp(q), _dp(qd), _dp(out))
    return out


def energy(cfg, q, qd):
    lib = load("f64")
    q, qd = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, qd))
    return lib.vine_oracle_energy(C.byref(cfg), _dp(q), _dp(qd))


class OracleEnv:
    """The oracle behind the same C ABI as the product (device_id = -1), numpy buffers."""

    def __init__(self, cfg, precision="f64", omp=False):
        self.lib = load(precision, omp)
        self.cfg = cfg
        self.n = cfg.num_envs
        self.num_obs = self.lib.vine_num_obs(C.byref(cfg))
        if self.num_obs < 0:
            raise NotImplementedError(self.lib.vine_last_error().decode())
        h = C.c_void_p()
        rc = self.lib.vine_create(C.byref(cfg), -1, None, C.byref(h))
        if rc:
            raise ValueError(self.lib.vine_last_error().decode())
        self.h = h
        self.real = np.float64 if self.lib.vine_oracle_real_bytes() == 8 else np.float32
        ptr = self.lib.vine_oracle_state(self.h)
        nbytes = abi.VF_COUNT * self.n * np.dtype(self.real).itemsize
        buf = (C.c_char * nbytes).from_address(ptr)
        self.state = np.frombuffer(buf, dtype=self.real).reshape(abi.VF_COUNT, self.n)
        self.obs = np.zeros((self.n, self.num_obs), np.float32)
        self.rew = np.zeros(self.n, np.float32)
        self.reset_buf = np.ones(self.n, np.int64)     # vec_task.py:275
        self.progress = np.zeros(self.n, np.int64)
        self.timeouts = np.zeros(self.n, np.uint8)
        self.reward_matrix = None
        self._reset_values = None

    def bind_reward_matrix(self):
        self.reward_matrix = np.zeros((self.n, abi.NUM_REWARDS), np.float32)
        self.lib.vine_bind_reward_matrix(self.h, self.reward_matrix.ctypes.data)
        return self.reward_matrix

    def bind_reset_values(self, values):
        if values is None:
            self._reset_values = None
            self.lib.vine_bind_reset_values(self.h, None)
        else:
            self._reset_values = np.ascontiguousarray(values, np.float32).reshape(self.n, 10)
            self.lib.vine_bind_reset_values(self.h, self._reset_values.ctypes.data)

    def set_formulation(self, form):
        self.lib.vine_oracle_set_formulation(self.h, form)

    def set_probe(self, armature=0.0, vmax_link=0.0, vmax_joint=0.0, effort_first_substep_only=False):
        """Oracle-only physics probe switches (tests/test_oracle_physics.py: sweep of unverifiable PhysX defaults)."""
        self.lib.vine_oracle_set_probe(self.h, armature, vmax_link, vmax_joint, int(effort_first_substep_only))

    def stats(self, index_to_view=0):
        """vine_stats: the dashboard vector (abi.VS_*) of the state the last step left behind."""
        out = np.zeros(abi.NUM_STATS, np.float32)
        rc = self.lib.vine_stats(self.h, self.rew.ctypes.data, self.progress.ctypes.data, int(index_to_view),
                                 out.ctypes.data, None)
        if rc:
            raise RuntimeError(self.lib.vine_last_error().decode())
        return out

    @property
    def step_count(self):
        return self.lib.vine_get_step_count(self.h)

    @step_count.setter
    def step_count(self, v):
        self.lib.vine_set_step_count(self.h, int(v))

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32).reshape(self.n, 2)
        rc = self.lib.vine_step(self.h, a.ctypes.data, self.obs.ctypes.data, self.rew.ctypes.data,
                                self.reset_buf.ctypes.data, self.progress.ctypes.data, self.timeouts.ctypes.data, None)
        if rc:
            raise RuntimeError(self.lib.vine_last_error().decode())
        return self.obs, self.rew, self.reset_buf, self.timeouts

    def reset_idx(self, env_ids):
        ids = np.ascontiguousarray(env_ids, np.int64)
        rc = self.lib.vine_reset_idx(self.h, ids.ctypes.data, len(ids), self.rew.ctypes.data,
                                     self.reset_buf.ctypes.data, self.progress.ctypes.data, None)
        if rc:
            raise RuntimeError(self.lib.vine_last_error().decode())

    def close(self):
        if self.h:
            self.state = None
            self.lib.vine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
