"""vine_rollout_post / vine_policy_head timings against the fraction of finished envs (HIP events, 200 launches)."""
import torch
from vine_robot_isaacgymenvs_amd.learning import fused

dev = "cuda"
lib = fused._lib()
N, H, A = 16384, 256, 2
st = torch.cuda.current_stream().cuda_stream
for frac in (0.0, 0.01, 0.3, 1.0):
    rew = torch.randn(N, device=dev)
    reset = (torch.rand(N, device=dev) < frac).long()
    tmo = torch.zeros(N, device=dev, dtype=torch.uint8)
    values = torch.randn(N, device=dev)
    shaped, dones = torch.empty(N, device=dev), torch.empty(N, device=dev, dtype=torch.uint8)
    cr, cl = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    h, c = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
    meter = torch.zeros(8, device=dev)
    counter = torch.zeros(1, device=dev, dtype=torch.long)
    xh = torch.zeros(N, 352, device=dev, dtype=torch.bfloat16)
    scratch = torch.empty(3072, device=dev)

    def f():
        fused._check(lib.vine_rollout_post(N, H, rew.data_ptr(), reset.data_ptr(), tmo.data_ptr(), values.data_ptr(), 0.0, 1.0,
                                           0.99, shaped.data_ptr(), dones.data_ptr(), cr.data_ptr(), cl.data_ptr(),
                                           h.data_ptr(), c.data_ptr(), meter.data_ptr(), 100.0, counter.data_ptr(),
                                           xh.data_ptr() + 2 * 96, 352, 1, scratch.data_ptr(), st), "post")
    for _ in range(10):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200):
        f()
    e.record()
    torch.cuda.synchronize()
    print("rollout_post (+ finalize) done fraction %.2f: %.1f us per call" % (frac, s.elapsed_time(e) * 5))
