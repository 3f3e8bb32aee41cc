#!/usr/bin/env python3
"""Can the recurrent (h) part of the rollout's LSTM gate product -- it depends on the PREVIOUS step's h only -- run on a second
stream under the env step and the MLP of the same step?  Times, at 16384 rows / envs, inside captured graphs: [MLP + env
step] alone, [split-product LSTM step] alone (stands in for the h part: 8 of its 11 k-steps), both back to back on one
stream, both concurrently on two streams."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vine_robot_isaacgymenvs_amd import load_config  # noqa: E402
from vine_robot_isaacgymenvs_amd.learning import fused  # noqa: E402
from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map  # noqa: E402

lib = fused._lib()
dev = torch.device("cuda:0")
N, H, K, F = 16384, 256, 352, 28
torch.manual_seed(0)
xh = [torch.randn(N, K, device=dev) for _ in range(2)]
wcat = torch.randn(4 * H, K, device=dev) / np.sqrt(K)
bias = torch.randn(4 * H, device=dev) * 0.1
c = torch.randn(N, H, device=dev)
h = torch.empty(N, H, device=dev)
wt = torch.empty(4 * H * K, device=dev)
raw = torch.randn(N, F, device=dev)
mean, var = torch.zeros(F, device=dev, dtype=torch.float64), torch.ones(F, device=dev, dtype=torch.float64)
Ws = [torch.randn(o, i, device=dev) / np.sqrt(i) for o, i in ((256, F), (128, 256), (64, 128))]
bs = [torch.randn(o, device=dev) * 0.1 for o in (256, 128, 64)]
w1p = torch.zeros(256, 32, device=dev)
w1p[:, :F] = Ws[0]
ws = torch.empty(3 * 4 * H * K, device=dev, dtype=torch.bfloat16)
assert lib.vine_lstm_tile_weights_split(H, K, wcat.data_ptr(), K, ws.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
cfg = load_config(overrides=["num_envs=%d" % N, "task.env.CREATE_PIPE=False"])
env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0", graphics_device_id=0, headless=True)
act = torch.rand(N, 2, device=dev) * 2 - 1
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def mlp(i, stream):
    a = xh[i & 1]
    st = stream.cuda_stream
    assert lib.vine_mlp3_elu_f32(N, a.data_ptr(), K, raw.data_ptr(), F, mean.data_ptr(), var.data_ptr(), 1e-5, 5.0, w1p.data_ptr(), 32,
                                 bs[0].data_ptr(), 256, Ws[1].data_ptr(), 256, bs[1].data_ptr(), 128, Ws[2].data_ptr(), 128,
                                 bs[2].data_ptr(), 64, 1.0, st) == 0


xh2 = [torch.randn(N, K, device=dev) for _ in range(2)]


def lstm(i, stream):
    a, b = xh2[i & 1], xh2[(i & 1) ^ 1]
    assert lib.vine_lstm_step_f32_split(N, H, K, a.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c.data_ptr(), h.data_ptr(), H,
                                        c.data_ptr(), b.data_ptr() + 4 * 96, K, 9, stream.cuda_stream) == 0


def envstep(stream):
    with torch.cuda.stream(stream):
        env._native_step(act, env.obs_buf)


def timeit(fn, iters=40):
    """`fn(i)` issued `iters` times inside ONE captured graph (the host cannot launch these kernels fast enough for the
    streams to overlap otherwise); the graph is replayed 5 times."""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        for i in range(iters):
            fn(i)
        cur.wait_stream(s1); cur.wait_stream(s2)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


t_a = timeit(lambda i: (envstep(s1), mlp(i, s1)))
t_b = timeit(lambda i: lstm(i, s1))
t_seq = timeit(lambda i: (envstep(s1), mlp(i, s1), lstm(i, s1)))
t_par = timeit(lambda i: (envstep(s1), mlp(i, s1), lstm(i, s2)))
print("16384 rows: env step + MLP %.1f us | split LSTM step %.1f us | one stream %.1f us | two streams %.1f us" % (t_a, t_b, t_seq, t_par))
env.close()
