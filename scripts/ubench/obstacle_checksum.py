"""Checksum of 400 env steps (16384 envs, fixed action stream) with the shelf, shelf + pipe and pipe configurations: equal
checksums from two builds of the library (VINE_HIP_LIB) = bit-identical trajectories; also the time per step incl. host."""
import sys, torch
sys.path.insert(0, ".")
from vine_robot_isaacgymenvs_amd import load_task_config
from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
for ov in (["task.env.CREATE_PIPE=False", "task.env.CREATE_SHELF=True"], ["task.env.CREATE_SHELF=True"], []):
    cfg = load_task_config("Vine5LinkMovingBase", overrides=["num_envs=16384"] + ov)
    env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg, rl_device="cuda:0", sim_device="cuda:0", graphics_device_id=0, headless=True)
    g = torch.Generator(device="cuda:0").manual_seed(0)
    acc = torch.zeros((), device="cuda:0", dtype=torch.float64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for t in range(400):
        a = torch.sign(torch.rand(16384, 2, device="cuda:0", generator=g) - 0.5) if t % 3 else torch.rand(16384, 2, device="cuda:0", generator=g) * 2 - 1
        if t == 100: e0.record()
        obs, rew, d, info = env.step(a)
        acc += obs["obs"].double().sum() * (t + 1) + rew.double().sum()
    e1.record(); torch.cuda.synchronize()
    print(ov, "checksum %.10e" % float(acc), "us/step incl. host %.1f" % (e0.elapsed_time(e1) / 300 * 1e3))
    env.close()
