"""Long random-action soak of the env kernel at full size (stability of contacts / resets at scale)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from vine_robot_isaacgymenvs_amd import abi, load_task_config  # noqa: E402
from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map  # noqa: E402

for name, ov in (("free", ["task.env.CREATE_PIPE=False"]), ("pipe", []),
                 ("shelf+pipe", ["task.env.CREATE_SHELF=True", "task.env.USE_NONZERO_CONTACT_FORCE_RESET=False"]),
                 ("shelf reset-on-contact", ["task.env.CREATE_PIPE=False", "task.env.CREATE_SHELF=True",
                                             "task.env.USE_NONZERO_CONTACT_FORCE_RESET=True"])):
    cfg = load_task_config("Vine5LinkMovingBase", overrides=["num_envs=16384"] + ov)
    env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg, rl_device="cuda:0", sim_device="cuda:0",
                                                  graphics_device_id=0, headless=True)
    g = torch.Generator(device="cuda:0").manual_seed(0)
    a = torch.zeros(16384, 2, device="cuda:0")
    mx_qd = torch.zeros((), device="cuda:0")
    done = torch.zeros((), device="cuda:0")
    contact = torch.zeros((), device="cuda:0")
    t0 = time.time()
    T = 3000
    for t in range(T):
        if t % 6 == 0:      # mix of bang-bang and random actions
            a = torch.where(torch.rand(16384, 1, device="cuda:0", generator=g) < 0.5,
                            torch.sign(torch.rand(16384, 2, device="cuda:0", generator=g) - 0.5),
                            torch.rand(16384, 2, device="cuda:0", generator=g) * 2 - 1)
        obs, rew, d, info = env.step(a)
        mx_qd = torch.maximum(mx_qd, env.state[abi.VF_QD0 + 1:abi.VF_QD0 + 6].abs().max())
        done += d.sum()
        contact += (env.state[abi.VF_CONTACT_MEAN] > 0).sum()
    torch.cuda.synchronize()
    dt = time.time() - t0
    st = env.state
    print("%-24s finite=%s max|qd|=%.1f max|q|=%.2f episodes=%d contact-steps=%d  %.1f M env-steps/s" % (
        name, bool(torch.isfinite(st).all()), float(mx_qd), float(st[abi.VF_Q0 + 1:abi.VF_Q0 + 6].abs().max()),
        int(done), int(contact), 16384 * T / dt / 1e6))
    env.close()
