#!/usr/bin/env python3
"""Hazard check for the hand-placed v_fmac_f32_dpp instructions of vine_step_quad_kernel (csrc/vine_hip.hip).

gfx9-family rule: a DPP instruction that READS a VGPR needs two wait states after the VALU instruction that WROTE it
(other instructions count one each, `s_nop N` counts N + 1).  The compiler inserts the nops for its own DPP instructions;
the ones inside asm statements it cannot see.  This script compiles the file to assembly and walks every function:
for each `v_*_dpp`, the DPP source operand (src0) must not have been written by a VALU instruction within the last two
wait states.  Exit status 1 and a listing when a violation is found.

    python scripts/check_dpp_hazards.py            (about two minutes: one device compile)
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "vine_robot_isaacgymenvs_amd", "csrc", "vine_hip.hip")


def regs(tok):
    """VGPR numbers named by an operand token: v12, v[4:5], with optional modifiers."""
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def main():
    asm = sys.argv[1] if len(sys.argv) > 1 else None
    if asm is None:
        asm = os.path.join(tempfile.gettempdir(), "vine_hip_dpp_check.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-ffp-contract=fast", "-fno-slp-vectorize", "-Wno-unused-function", "--cuda-device-only", "-S",
                               "-o", asm, SRC], stderr=subprocess.DEVNULL)
    bad, n_dpp, func = [], 0, "?"
    recent = []            # (wait states provided, set of VGPRs written by a VALU instruction)
    for ln, line in enumerate(open(asm), 1):
        t = line.strip()
        if re.match(r"^_Z\w+:", t):
            func, recent = t.split(":")[0], []
            continue
        if re.match(r"^\.LBB", t):
            recent = []        # a label: predecessors unknown; the compiler's own scheduling regions end here too
            continue
        m = re.match(r"^([a-z]\w*)\s*(.*?)(;.*)?$", t)
        if not m or not re.match(r"^[vsdgb]_|^buffer_|^global_|^flat_|^scratch_", m.group(1)):
            continue
        op, args = m.group(1), m.group(2)
        toks = [a.strip() for a in args.split(",")] if args else []
        if op.endswith("_dpp"):
            n_dpp += 1
            src0 = regs(toks[1]) if len(toks) > 1 else set()
            dist = 0
            for ws, written in reversed(recent):
                if dist >= 2:
                    break
                if written & src0:
                    bad.append((func, ln, t, dist))
                    break
                dist += ws
        if op == "s_nop":
            recent.append((int(toks[0], 0) + 1, set()))
        elif op.startswith("v_") and not op.startswith("v_cmp") and toks:
            recent.append((1, regs(toks[0])))
        else:
            recent.append((1, set()))
        recent = recent[-6:]
    print("%d DPP instructions checked, %d unprotected reads" % (n_dpp, len(bad)))
    for f, ln, t, d in bad[:40]:
        print("  %s line %d (only %d wait state(s) after the write): %s" % (f[:60], ln, d, t))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
