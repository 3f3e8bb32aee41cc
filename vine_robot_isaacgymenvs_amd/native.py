"""Build and load ``libvine_hip.so`` — the only compute backend of this package.

There is deliberately no CPU fallback: if the HIP extension is missing or no MI355X is
visible, loading/creating fails loudly.
"""
import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "vine_hip.hip")
SRC_PPO = os.path.join(_HERE, "csrc", "ppo_kernels.hip")
LIB = os.path.join(_HERE, "libvine_hip.so")
ARCH = "gfx950"

_lib = None


def build(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 (works without a GPU); output stays in-tree."""
    inc = os.path.join(os.path.dirname(_HERE), "include")
    deps = [SRC, SRC_PPO, os.path.join(inc, "vine.h"), os.path.join(inc, "vine_ppo.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast", "-fno-slp-vectorize",
           "-Wall", "-Wno-unused-function", "-o", LIB, SRC, SRC_PPO]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    extra = os.environ.get("VINE_HIPCC_FLAGS")       # experiments only (e.g. "-fslp-vectorize")
    if extra:
        cmd[-4:-4] = extra.split()      # after the default flags (before "-o"), so that they win
    subprocess.check_call(cmd)
    return LIB


def load():
    """Return the ctypes handle of libvine_hip.so with prototypes attached."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64.so.7; it must be the HIP runtime of the process, so it is
        # loaded first (two runtimes in one process do not both see the device).
        import torch  # noqa: F401
        if not os.path.exists(LIB):
            raise RuntimeError(
                "libvine_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or vine_robot_isaacgymenvs_amd.native.build(); this package has no CPU fallback." % LIB)
        _lib = abi.declare_ppo(abi.declare(C.CDLL(LIB)))
    return _lib


def check(rc, lib=None):
    if rc == abi.OK:
        return
    lib = lib or load()
    msg = lib.vine_last_error().decode(errors="replace")
    if rc == abi.ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == abi.ERR_INVALID_ARG:
        raise ValueError(msg)
    raise RuntimeError("libvine_hip error %d: %s" % (rc, msg))
