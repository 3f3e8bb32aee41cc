"""``bench.py --mode ppo``: time whole PPO iterations (rollout with policy inference + update with the RCCL
gradient all-reduce) on the env already built by bench.py."""
import time

import torch


def ppo_kernel_rooflines(device, B=8192, T=4, H=256, width=92, wpad=96, sets=4):
    """Stand-alone HIP-event timings of the two largest hand-written kernels of the mixed-precision update at its
    shapes (the LSTM over the 4-step sequences of a 32768-sample minibatch, ONE persistent launch per direction),
    priced against HBM with their algorithmic bytes.  Every launch runs on its own operand / output set, `sets` of them
    in rotation (> 256 MB in total: more than the Infinity Cache holds), so the figures are cold-cache ones; the
    in-situ durations of the same kernels are in profiles/r03/ppo_iteration_graphed_kernel_stats.csv.  ("bf16" below =
    the library's 16-bit operand format: float16 in the default build.)
      lstm_seq_fwd_kernel ("h once", the update's form since round 3): reads x bf16 [B*T, wpad] + h0, c0 fp32 [B, H] +
          weights (bf16, 4H x (wpad + H)); writes the ONE copy of the hidden states bf16 [B, T+1, H] (slot 0 = h0),
          c bf16 [T-1, B, H] + c_T fp32, gate activations bf16 [T, B, 4H]
      lstm_seq_bwd_kernel: reads dh bf16 [B*T, H], gates bf16 [T, B, 4H], c (c0, c_T fp32; the rest bf16), w_hh bf16;
          writes dG bf16 [B*T, 4H]"""
    from . import fused
    lib = fused._lib()
    st = torch.cuda.current_stream(device).cuda_stream
    bf = fused.lp_dtype()          # the library's 16-bit operand format (float16 in the default build)
    w_ih = (torch.randn(4 * H, width, device=device) / 10).to(bf)
    w_hh = (torch.randn(4 * H, H, device=device) / 16).to(bf)
    wtile = torch.empty(4 * H * (wpad + H), device=device, dtype=bf)
    whh_tiled = torch.empty(4 * H * H, device=device, dtype=bf)
    prep = fused.CopyBatch()
    prep.add_lstm_tiles(w_ih, w_hh, wpad, wtile, whh_tiled)
    prep.flush(wtile)
    bias = torch.zeros(4 * H, device=device)
    data = []
    for _ in range(sets):
        x = torch.zeros(B * T, wpad, device=device, dtype=bf)
        x[:, :width] = (torch.randn(B * T, width, device=device) * 0.7).to(bf)
        c0 = torch.randn(B, H, device=device) * 0.5
        h0 = torch.randn(B, H, device=device) * 0.5
        dones = (torch.rand(B * T, device=device) < 0.2).to(torch.uint8)
        data.append(dict(x=x, c0=c0, h0=h0, dones=dones, out=torch.empty(B * (T + 1), H, device=device, dtype=bf),
                         c_all=torch.empty(T + 1, B, H, device=device, dtype=bf), c_last=torch.empty(B, H, device=device),
                         gates=torch.empty(T, B, 4 * H, device=device, dtype=bf),
                         g_out=(torch.randn(B * T, H, device=device) * 0.1).to(bf),
                         dG=torch.empty(B * T, 4 * H, device=device, dtype=bf)))
    part = torch.empty(B // 32, 4 * H, device=device)

    def fwd(i):
        d = data[i % sets]
        assert lib.vine_lstm_seq_forward_mfma(B, T, H, wpad, d["x"].data_ptr(), wpad, None, T * H,
                                              wtile.data_ptr(), bias.data_ptr(), d["c0"].data_ptr(), d["dones"].data_ptr(),
                                              d["out"].data_ptr(), d["c_all"].data_ptr(), d["gates"].data_ptr(), 1 | 2,
                                              d["c_last"].data_ptr(), d["h0"].data_ptr(), st) == 0

    def bwd(i):
        d = data[i % sets]
        assert lib.vine_lstm_seq_backward_mfma(B, T, H, d["g_out"].data_ptr(), whh_tiled.data_ptr(), d["gates"].data_ptr(),
                                               d["c_all"].data_ptr(), d["c0"].data_ptr(), d["dones"].data_ptr(),
                                               d["dG"].data_ptr(), part.data_ptr(), 1, d["c_last"].data_ptr(), 1, st) == 0

    # (production configuration: saved cell states c_1 .. c_{T-1} and the hidden-state gradient in 16 bits, c_T fp32)
    fwd_bytes = (B * T * wpad * 2 + 2 * B * H * 4 + wtile.numel() * 2 + B * T
                 + B * (T + 1) * H * 2 + ((T - 1) * B * H * 2 + B * H * 4) + T * B * 4 * H * 2)
    bwd_bytes = (B * T * H * 2 + T * B * 4 * H * 2 + (2 * B * H * 4 + (T - 1) * B * H * 2) + whh_tiled.numel() * 2 + B * T
                 + B * T * 4 * H * 2 + part.numel() * 4)
    # what the memory system of THIS box sustains for a plain streaming kernel (y = x + 1 over 1 GiB, read + write):
    # the practical ceiling for mixed read/write traffic, quoted beside the 8 TB/s spec figure
    xs = torch.empty(1 << 28, device=device)
    ys = torch.empty_like(xs)
    for _ in range(3):
        torch.add(xs, 1.0, out=ys)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.add(xs, 1.0, out=ys)
    e1.record()
    torch.cuda.synchronize(device)
    stream_gbs = 10 * 2 * 4 * xs.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del xs, ys
    res = []
    for name, f, nbytes in (("lstm_seq_fwd_kernel", fwd, fwd_bytes), ("lstm_seq_bwd_kernel", bwd, bwd_bytes)):
        for i in range(2 * sets):
            f(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10 * sets
        for i in range(n):
            f(i)
        e1.record()
        torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / n
        res.append({"kernel": name, "us": us, "algorithmic_bytes_per_launch": nbytes, "achieved_GBs": nbytes / us / 1e3,
                    "hbm_frac": nbytes / us / 1e3 / 8000.0, "launches_per_ppo_iteration": 32, "cache_state": "cold (rotating sets)",
                    "streaming_kernel_GBs": stream_gbs, "frac_of_streaming_kernel": nbytes / us / 1e3 / stream_gbs})
    # the rollout's LSTM step (fp32 operands, exact products from bf16 pieces): MFMA-bound.  Priced twice: the bf16 matrix
    # work it executes (9 piece pairs) against the dense bf16 peak, and the fp32 product it delivers against the fp32
    # matrix peak (what the native fp32 instruction could reach at most)
    try:
        N, K = 16384, 352
        xh = [torch.randn(N, K, device=device) for _ in range(2)]
        wcat = torch.randn(4 * H, K, device=device) / K ** 0.5
        ws = torch.empty(3 * 4 * H * K, device=device, dtype=torch.bfloat16)
        c = torch.randn(N, H, device=device)
        hbuf = torch.empty(N, H, device=device)
        assert lib.vine_lstm_tile_weights_split(H, K, wcat.data_ptr(), K, ws.data_ptr(), st) == 0

        def step(i):
            a, b = xh[i & 1], xh[(i & 1) ^ 1]
            assert lib.vine_lstm_step_f32_split(N, H, K, a.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), c.data_ptr(),
                                                hbuf.data_ptr(), H, c.data_ptr(), b.data_ptr() + 4 * 96, K, 9, st) == 0
        for i in range(10):
            step(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            step(i)
        e1.record()
        torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / 50
        flop32 = 2.0 * N * K * 4 * H
        res.append({"kernel": "lstm_step_split_kernel", "bound": "mfma", "us": us, "rows": N,
                    "algorithmic_fp32_flops_per_launch": flop32, "executed_bf16_flops_per_launch": 9 * flop32,
                    "achieved_TFLOPs": 9 * flop32 / us / 1e6, "peak_TFLOPs": 2500.0, "frac": 9 * flop32 / us / 1e6 / 2500.0,
                    "delivered_fp32_TFLOPs": flop32 / us / 1e6, "fp32_matrix_peak_TFLOPs": 157.3,
                    "launches_per_ppo_iteration": 17, "cache_state": "back to back"})
    except AssertionError:
        pass
    return res


def ppo_path_rooflines(agent):
    """Rows for the kernels that are ON the timed path, keyed by the kernel names of the graph-replayed trace (VERDICT r4 item
    3).  bench.py cannot trace or count by itself: the in-situ duration of a kernel (``in_situ_us``: rocprofv3 kernel trace of
    `bench.py --mode ppo` with the iteration replayed from hipGraphs) and its HBM-side traffic (``traffic_bytes`` = (2 x
    FETCH_SIZE + WRITE_SIZE) x 1024, separate counter passes; the x2 is MI355X_MICROARCH.md's gfx950 correction) come from the
    newest committed ``profiles/rNN/ppo_traffic_*_pmc_summary.json`` (scripts/profile_ppo_traffic.sh), named in ``source``.
    ``algorithmic_bytes`` are computed here from the shapes of the agent that just ran (formulas below);
    ``hbm_frac`` = algorithmic bytes / in-situ duration / 8 TB/s; ``launches_per_iteration`` x ``in_situ_us`` is the kernel's
    share of the iteration.  Everything reproduces by hand from the named file."""
    import glob
    import json
    import os
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    # the newest committed profile taken at this env count (file names carry it: ppo_traffic_<tag>_<N>envs_pmc_summary.json)
    files = sorted(glob.glob(os.path.join(repo, "profiles", "r*", "ppo_traffic_*_%denvs_pmc_summary.json" % agent.num_actors)),
                   reverse=True)
    if not files:
        return None
    prof = json.load(open(files[0]))
    net = agent.model.a2c_network
    n = agent.minibatch_size                      # samples per optimiser step
    T = agent.seq_len
    B = n // T                                    # sequences per optimiser step
    H = net.rnn_units
    F = agent.obs_shape[0]
    U1, U2, U3 = net.units
    wpad = (U3 + F + 15) // 16 * 16               # LSTM input block [MLP out | obs | pad]
    A = agent.actions_num
    steps = agent.mini_epochs_num * agent.num_minibatches
    N = agent.num_actors
    lp = 2                                        # bytes of the 16-bit operand format
    # LSTM forward ("h once"): x, h0 / c0 (fp32), weights, done flags in; h [B, T+1, H], c_1..c_{T-1} (16 bit) + c_T (fp32), gates out
    fwd = B * T * wpad * lp + 2 * B * H * 4 + 4 * H * (wpad + H) * lp + B * T + B * (T + 1) * H * lp + ((T - 1) * B * H * lp + B * H * 4) + T * B * 4 * H * lp
    # LayerNorm + heads + loss + backward: h rows and the per-sample loss inputs in; dh, heads, mu / sigma refresh out
    loss = n * H * lp + n * (3 * A + 4) * 4 + n * H * lp + n * (A + 1) * 4 + 2 * n * A * 4
    # LSTM backward: dh, gates, cell states, w_hh, done flags in; dG out
    bwd = n * H * lp + T * B * 4 * H * lp + (2 * B * H * 4 + (T - 1) * B * H * lp) + 4 * H * H * lp + B * T + n * 4 * H * lp
    # MLP backward: dG (second pass) and the three stored activations in; the three pre-activation gradients out
    mlpb = n * 4 * H * lp + 2 * n * (U1 + U2 + U3) * lp + (U3 * 4 * H + U2 * U3 + U1 * U2) * lp
    # LSTM weight gradients: dG and [x | h] in, 32 row slices of the [4H, wpad + H] fp32 product out
    wcat = n * 4 * H * lp + n * wpad * lp + B * T * H * lp + 32 * 4 * H * (wpad + H) * 4
    # MLP forward: observations (fp32) in, three activations out (+ the LSTM operand's observation block)
    mlpf = n * F * 4 + n * (U1 + U2 + U3 + 32) * lp
    rows = [
        ("trunk_phases_kernel", fwd + loss + bwd + mlpb, steps,
         {"phases": {"lstm_forward": fwd, "layernorm_heads_loss": loss, "lstm_backward": bwd, "mlp_backward": mlpb}}),
        ("wgrad_cat_wide_kernel", wcat, steps, {}),
        ("mlp3_elu_mfma_kernel", mlpf, steps, {}),
    ]
    # the step kernel of the rollout: since round 5 the instantiation with the policy head and the rollout bookkeeping inside
    # (last template argument true: + the LSTM output rows in, mu / sigma / value / action / neglogp / shaped reward / done out)
    roll = getattr(agent, "rollout_step_launches", 5) == 3
    env_bytes = (320 if F == 28 else 280) * N + (N * H * 4 + N * (3 * A + 3) * 4 + N if roll else 0)
    rows.append(("vine_step_quad_kernel", env_bytes, agent.horizon_length,
                 {"bound": "valu", "policy_head_and_bookkeeping_inside": roll}))
    out = []
    for name, algo, launches, more in rows:
        cands = [k for k in prof if k.startswith(name)]
        if name == "vine_step_quad_kernel":
            cands = [k for k in cands if k.rstrip(">").endswith(", true" if roll else ", false")] or ([] if roll else cands)
        key = cands[0] if cands else None
        if key is None:
            continue
        r = prof[key]
        us = r["avg_ns"] / 1e3
        row = {"kernel": key, "in_situ_us": us, "launches_per_iteration": launches, "algorithmic_bytes": int(algo),
               "achieved_GBs": algo / us / 1e3, "peak_GBs": 8000.0, "hbm_frac": algo / us / 1e3 / 8000.0,
               "traffic_bytes": r.get("traffic_bytes"), "source": os.path.relpath(files[0], repo)}
        if r.get("traffic_bytes"):
            row["traffic_over_algorithmic"] = r["traffic_bytes"] / algo
            row["traffic_hbm_frac"] = r["traffic_bytes"] / us / 1e3 / 8000.0
        row.update(more)
        out.append(row)
    # MFMA-bound rows from the same trace: the rollout's LSTM step (fp32 products from bf16 pieces)
    terms = (getattr(agent, "_fast", None) or {}).get("f32_split", 0)
    key = next((k for k in prof if k.startswith("lstm_step_split_kernel") or k.startswith("lstm_step_nsplit_kernel")), None)
    if key is not None and terms:
        import re
        m = re.search(r"<\d+, (?:\d+, )?(\d+)(?:, (?:true|false))?>", key)      # <KS, RT, NT> or <KS, NT, DUAL>: piece pairs
        kt = int(m.group(1)) if m else terms
        us = prof[key]["avg_ns"] / 1e3
        flop32 = 2.0 * N * (wpad + H) * 4 * H
        out.append({"kernel": key, "bound": "mfma", "in_situ_us": us, "launches_per_iteration": agent.horizon_length + 1,
                    "algorithmic_fp32_flops": flop32, "executed_bf16_flops": kt * flop32,
                    "achieved_TFLOPs": kt * flop32 / us / 1e6, "peak_TFLOPs": 2500.0, "frac": kt * flop32 / us / 1e6 / 2500.0,
                    "delivered_fp32_TFLOPs": flop32 / us / 1e6, "traffic_bytes": prof[key].get("traffic_bytes"),
                    "source": os.path.relpath(files[0], repo)})
    return out


def ppo_iteration_rate(env, cfg, steps=20, warmup=3, amp=None, use_graphs=True):
    """Whole PPO iterations (rollout with policy inference + GAE + every optimiser step) on ``env`` with the packaged
    train config: the same quantity as the bench line's ``value`` for another task configuration (bench.py `configs`)."""
    import copy
    from .a2c_continuous import A2CAgent
    params = copy.deepcopy(cfg["train"]["params"])
    conf = params["config"]
    conf.update(device=str(env.device), device_pinned=True, multi_gpu=False, write_files=False, print_stats=False,
                use_graphs=use_graphs)
    if amp == "off":
        conf["mixed_precision"] = False
    elif amp:
        conf["mixed_precision"], conf["mixed_precision_dtype"] = True, amp
    agent = A2CAgent("bench_cfg", params, vec_env=env)
    agent.init_tensors()
    agent.obs = agent.env_reset()["obs"].to(agent.device)
    for _ in range(warmup):
        agent.train_epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    play = upd = 0.0
    for _ in range(steps):
        p, u, _s = agent.train_epoch()
        play += p
        upd += u
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = agent.horizon_length * agent.num_actors
    k = dt / max(play + upd, 1e-12)
    return {"env_steps_per_sec": frames * steps / dt, "ms_per_iteration": dt / steps * 1e3, "rollout_ms": play * k / steps * 1e3,
            "update_ms": upd * k / steps * 1e3, "minibatch": agent.minibatch_size,
            "optimizer_steps_per_iter": agent.mini_epochs_num * agent.num_minibatches,
            "hipgraphs_active": dict(agent.graph_status), "iterations_timed": steps}


def run_ppo_bench(env, cfg, args, steps, warmup, barrier, world, rank):
    from .a2c_continuous import A2CAgent

    params = cfg["train"]["params"]
    conf = params["config"]
    conf["device"] = str(env.device)
    conf["device_pinned"] = True          # bench.py already chose the device of this rank
    conf["multi_gpu"] = world > 1 or bool(getattr(args, "force_multi_gpu_path", False))
    conf["write_files"] = False
    conf["print_stats"] = False
    conf["use_graphs"] = not args.no_graph
    if args.amp == "off":
        conf["mixed_precision"] = False
    elif args.amp:
        conf["mixed_precision"], conf["mixed_precision_dtype"] = True, args.amp
    agent = A2CAgent("bench", params, vec_env=env)
    agent.init_tensors()
    agent.obs = agent.env_reset()["obs"].to(agent.device)
    agent.broadcast_parameters()

    # HIP events around every env-step launch of the timed region, on the launching stream
    events = []
    orig = env._native_step

    def timed_native_step(a, o):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(a, o)
        e1.record()
        events.append((e0, e1))

    for _ in range(warmup):
        agent.train_epoch()
    if not conf["use_graphs"]:
        env._native_step = timed_native_step
    barrier()
    play = upd = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        p, u, _stats = agent.train_epoch()
        play += p
        upd += u
    barrier()
    elapsed = time.perf_counter() - t0
    env._native_step = orig
    # (train_epoch waits for the PREVIOUS iteration since round 4: the last iteration of the timed region ends inside the
    # barrier above, so the per-call times sum to a little less than `elapsed` -- the split is rescaled to it)
    if play + upd > 0:
        k = elapsed / (play + upd)
        play, upd = play * k, upd * k

    # env step alone (resident random actions), same process, right after the timed region
    g = torch.Generator(device=agent.device).manual_seed(1)
    pool = [torch.rand((env.num_envs, 2), device=agent.device, generator=g) * 2 - 1 for _ in range(16)]
    pairs = []
    n_env_only = 500
    for i in range(50):
        orig(pool[i % 16], env.obs_buf)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(n_env_only):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(pool[i % 16], env.obs_buf)
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    env_only_s = time.perf_counter() - t1
    env_only_kernel_ms = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
    # the same kernel as the DEVICE runs it inside the rollout graph: 64 launches captured into one hipGraph and replayed,
    # timed by ONE pair of events around the replays -- event pairs around single eager launches are host-bound (25.7 us for
    # a kernel whose trace duration is 23.1 us); the graph form leaves only the ~1 us between two nodes
    graph_kernel_ms = None
    if conf["use_graphs"] and getattr(env, "graph_capturable", True):
        try:
            gcap = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(gcap, capture_error_mode="thread_local"):
                for i in range(64):
                    orig(pool[i % 16], env.obs_buf)
            gcap.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                gcap.replay()
            e1.record()
            torch.cuda.synchronize()
            graph_kernel_ms = e0.elapsed_time(e1) / (8 * 64)
            del gcap
        except RuntimeError:
            torch.cuda.synchronize()
    # kernel duration inside the timed region when it was launched eagerly; otherwise (launches live inside the
    # replayed hipGraph, where no event can be recorded) the same kernel replayed back to back from a graph just above
    kernel_ms = (sum(a.elapsed_time(b) for a, b in events) / len(events) if events
                 else (graph_kernel_ms if graph_kernel_ms is not None else env_only_kernel_ms))
    in_sync = None
    if world > 1:       # replicas must hold identical weights and learning rate after all-reduced updates
        import torch.distributed as dist
        chk = torch.stack([torch.cat([p.detach().flatten() for p in agent.model.parameters()]).double().sum(),
                           agent.lr.double()])
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        in_sync = all(bool(torch.equal(g, gathered[0])) for g in gathered)
    frames = agent.horizon_length * agent.num_actors
    def describe(a):
        from . import fused
        lp = {torch.float16: "fp16", torch.bfloat16: "bf16"}.get(a.amp_dtype, "?")
        if a.fused_mixed:
            upd = ("update: %s GEMM operands and backward-only saved activations (gates, saved cell states, dG, dh), f32 "
                   "accumulate / recurrent state / heads / loss / optimiser%s (mixed_precision: True = the reference YAML's "
                   "value, PY:53)" % (lp, ", device-side GradScaler loss scaling" if a.optimizer.amp_state is not None else ""))
        elif a.mixed_precision:
            upd = "update: torch autocast %s" % lp
        else:
            upd = "update: f32"
        terms = (getattr(a, "_fast", None) or {}).get("f32_split", 0)
        if getattr(a, "rollout_lp16", False):
            roll = "rollout inference: %s GEMM operands" % lp
        elif terms:
            mlp_split = (getattr(a, "_fast", None) or {}).get("mlp_wt_split") is not None
            roll = ("rollout inference: f32 (the reference's): fp32 operands, fp32 accumulation; %s products formed "
                    "from the exact three-way bf16 split of both fp32 operands on the bf16 matrix cores, %s"
                    % ("MLP and LSTM-gate" if mlp_split else "MLP on the fp32 matrix cores, LSTM-gate",
                       "all 9 piece pairs = every bit of every fp32 product" if terms == 9 else
                       "6 of 9 piece pairs (the three below 2^-24 of a product left out)")
                    + (", two fp32 accumulators per tile (hi x hi pair apart from the smaller pairs)" if fused.ROLLOUT_F32_DUAL
                       else "")
                    + ": error against float64 <= %s the native fp32 matrix-core instruction's on every input measured, max and "
                      "rms (profiles/r05/split_terms_error.txt)" % ("0.6x" if fused.ROLLOUT_F32_DUAL else "1.2x (max), 0.9x (rms)"))
        else:
            roll = "rollout inference: f32 (the reference's; fp32 matrix-core kernels)"
        return upd + "; " + roll

    precision = describe(agent)

    def rerun(**over):
        conf2 = dict(conf)
        conf2.update(over)
        params2 = dict(params)
        params2["config"] = conf2
        agent2 = A2CAgent("bench2", params2, vec_env=env)
        agent2.init_tensors()
        agent2.obs = agent2.env_reset()["obs"].to(agent2.device)
        for _ in range(warmup):
            agent2.train_epoch()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        p2 = u2 = 0.0
        for _ in range(steps):
            p, u, _s = agent2.train_epoch()
            p2 += p
            u2 += u
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t2
        k2 = e2 / max(p2 + u2, 1e-12)
        return {"update_precision": describe(agent2), "value": frames * steps / e2, "unit": "env-steps/s",
                "ppo_iters_per_sec": steps / e2, "rollout_ms": p2 * k2 / steps * 1e3, "update_ms": u2 * k2 / steps * 1e3}

    other = lp16_rollout = native_f32_rollout = split_other_rollout = None
    if world == 1 and not getattr(args, "no_secondary", False) and not agent.mixed_precision:
        # the same PPO iteration in the OTHER update precision (fp32 <-> 16-bit operands), and -- an extra, NARROWER than the
        # reference's fp32 rollout -- with 16-bit GEMM operands in the rollout inference as well (the round-2 configuration);
        # same env, same process: reported beside the headline so that all of them are always visible
        other = rerun(mixed_precision=not agent.fused_mixed)
        if agent.fused_mixed and not agent.rollout_lp16:
            lp16_rollout = rerun(rollout_precision="lp16")
        if agent.fused_mixed and not agent.rollout_lp16 and (agent._fast or {}).get("f32_split", 0):
            # the rollout's LSTM step on the native fp32 matrix-core instruction, and the 6-pair form of the split
            native_f32_rollout = rerun(rollout_f32_terms=0)
            other_terms = 9 if (agent._fast or {}).get("f32_split", 0) == 6 else 6
            split_other_rollout = (other_terms, rerun(rollout_f32_terms=other_terms))
    ppo_kernels = None
    if world == 1 and agent.fused_mixed and not getattr(args, "no_secondary", False):
        try:
            ppo_kernels = ppo_path_rooflines(agent)
        except (AssertionError, RuntimeError, KeyError, ValueError) as err:
            ppo_kernels = "unavailable: %s" % str(err)[:100]
    extra = {
        "update_precision": precision,
        "ppo_kernel_rooflines": ppo_kernels,
        "other_precision": other,
        "extra_lp16_rollout": lp16_rollout,
        "extra_native_f32_mfma_rollout": native_f32_rollout,
        # the other form of the split products (9 = all piece pairs when the default is 6, and the other way round)
        ("extra_split%d_rollout" % split_other_rollout[0]) if split_other_rollout else "extra_split_other_rollout":
            split_other_rollout[1] if split_other_rollout else None,
        "replicas_in_sync": in_sync,
        # several ranks: where the gradient all-reduce ran (inside the mini-epoch graph, or eagerly between per-step graphs)
        # and what the collective probe said -- top level, next to replicas_in_sync (also under ppo.collective_in_graph)
        "collective_in_graph": getattr(agent, "collective_capture", None) if (world > 1 or conf["multi_gpu"]) else None,
        "env_only": {"env_steps_per_sec": env.num_envs * world * n_env_only / env_only_s, "kernel_ms": env_only_kernel_ms,
                     "graph_replayed_kernel_ms": graph_kernel_ms,
                     "steps": n_env_only, "note": "VecTask.step alone on resident random actions, per-rank x ranks"},
        "ppo_iters_per_sec": steps / elapsed,
        "rollout_env_steps_per_sec": frames * world * steps / max(play, 1e-9),
        "rollout_ms": play / steps * 1e3, "update_ms": upd / steps * 1e3,
        "ppo": {"horizon": agent.horizon_length, "minibatch": agent.minibatch_size, "mini_epochs": agent.mini_epochs_num,
                "optimizer_steps_per_iter": agent.mini_epochs_num * agent.num_minibatches,
                "params": agent.num_params, "mixed_precision": bool(agent.fused_mixed or agent.mixed_precision),
                # the OUTCOME, not the request: a refused capture falls back to eager launches
                "hipgraphs_requested": bool(conf["use_graphs"]), "hipgraphs_active": dict(agent.graph_status),
                "hipgraphs": agent.graph_status["rollout"] == "graph" and agent.graph_status["update"].startswith("graph"),
                "update_hipgraphs": len(getattr(agent, "_upd_graphs", {})),
                # several ranks: did the RCCL all-reduce go INTO the mini-epoch graph (probe outcome), or between per-step graphs
                "collective_in_graph": getattr(agent, "collective_capture", None)},
    }
    return elapsed, kernel_ms, frames, extra
