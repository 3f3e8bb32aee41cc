"""Build and load ``libvine_hip.so`` — the only compute backend of this package.

There is deliberately no CPU fallback: if the HIP extension is missing or no MI355X is
visible, loading/creating fails loudly.
"""
import ctypes as C
import hashlib
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "vine_hip.hip")
SRC_PPO = os.path.join(_HERE, "csrc", "ppo_kernels.hip")
LIB = os.path.join(_HERE, "libvine_hip.so")
ARCH = "gfx950"

FINGERPRINT = LIB + ".fingerprint"
_INC = os.path.join(os.path.dirname(_HERE), "include")
DEPS = [SRC, SRC_PPO, os.path.join(_INC, "vine.h"), os.path.join(_INC, "vine_ppo.h")]

_lib = None


def source_fingerprint():
    """sha256 over the kernel sources and the two ABI headers: what a built library must correspond to."""
    h = hashlib.sha256()
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(os.environ.get("VINE_HIPCC_FLAGS", "").encode())
    return h.hexdigest()


def is_fresh():
    try:
        return os.path.exists(LIB) and open(FINGERPRINT).read().strip() == source_fingerprint()
    except OSError:
        return False


def _flags():
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize", "-Wall",
             "-Wno-unused-function"]
    extra = os.environ.get("VINE_HIPCC_FLAGS")       # experiments only (e.g. "-fslp-vectorize"); after the defaults, so they win
    return flags + (extra.split() if extra else [])


def _object_fingerprint(src):
    h = hashlib.sha256()
    for d in (src, os.path.join(_INC, "vine.h"), os.path.join(_INC, "vine_ppo.h")):
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 (works without a GPU); output stays in-tree.  The two translation units are
    compiled side by side into ``build/obj`` (each object is reused while its source, the headers and the flags are
    unchanged) and linked into the one library.  A sidecar fingerprint of the sources is written next to the library:
    ``load()`` refuses to call into a library built from other sources (a stale binary behind a changed C signature is a
    wild pointer on the GPU)."""
    if not force and is_fresh():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(os.path.dirname(_HERE), "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    if os.path.exists(FINGERPRINT):
        os.remove(FINGERPRINT)
    jobs, objs = [], []
    for src in (SRC, SRC_PPO):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        fp = _object_fingerprint(src)
        try:
            fresh = not force and os.path.exists(obj) and open(obj + ".fingerprint").read().strip() == fp
        except OSError:
            fresh = False
        if fresh:
            continue
        cmd = [hipcc] + _flags() + ["-c", "-o", obj, src]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if os.path.exists(obj + ".fingerprint"):
            os.remove(obj + ".fingerprint")
        jobs.append((subprocess.Popen(cmd), cmd, obj, fp))
    for proc, cmd, obj, fp in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        with open(obj + ".fingerprint", "w") as f:
            f.write(fp + "\n")
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    with open(FINGERPRINT, "w") as f:
        f.write(source_fingerprint() + "\n")
    return LIB


def load():
    """Return the ctypes handle of libvine_hip.so with prototypes attached."""
    global _lib
    if _lib is None and os.environ.get("VINE_HIP_LIB"):
        # experiments only (A/B builds of the kernels with other compile flags): load exactly this file
        import torch  # noqa: F401
        _lib = abi.declare_ppo(abi.declare(C.CDLL(os.environ["VINE_HIP_LIB"])))
        return _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64.so.7; it must be the HIP runtime of the process, so it is
        # loaded first (two runtimes in one process do not both see the device).
        import torch  # noqa: F401
        if not os.path.exists(LIB):
            raise RuntimeError(
                "libvine_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or vine_robot_isaacgymenvs_amd.native.build(); this package has no CPU fallback." % LIB)
        if not is_fresh():
            # sources changed since the build (or the sidecar is missing): rebuild in place when a compiler is
            # here, otherwise stop -- never run a binary whose ABI may not match abi.py
            if os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
                import fcntl
                with open(LIB + ".lock", "w") as lock:      # ranks of one job: one of them rebuilds, the others wait
                    fcntl.flock(lock, fcntl.LOCK_EX)
                    if not is_fresh():
                        print("libvine_hip.so does not match the sources: rebuilding (hipcc, ~2 min)", flush=True)
                        build()
            else:
                raise RuntimeError("libvine_hip.so was built from different sources and no hipcc is available")
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
