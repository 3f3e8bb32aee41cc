"""Time the two env-step kernels (one lane per env / four lanes per env) at several env counts.
usage: python scripts/ubench/step_kernel_bench.py [num_envs ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from tests.helpers import base_cfg  # noqa: E402
from tests.hip_env import HipEnv  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [4096, 16384, 32768, 65536, 262144]:
    row = []
    for kern in ("lane", "quad"):
        cls = type("H", (HipEnv,), {"kernel": kern})
        cfg = base_cfg(n, 0, True)
        cfg.set_flag(1 << 15, False)
        env = cls(cfg)
        env.set_introspection(False)
        g = torch.Generator(device=env.dev).manual_seed(0)
        acts = [torch.rand((n, 2), device=env.dev, generator=g) * 2 - 1 for _ in range(8)]
        for i in range(30):
            env.step_t(acts[i % 8], sync=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            env.step_t(acts[i % 8], sync=False)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 200 * 1e3)
        env.close()
    print("num_envs %7d   one lane per env %7.1f us   four lanes per env %7.1f us   ratio %.2f" % (n, row[0], row[1], row[0] / row[1]), flush=True)
