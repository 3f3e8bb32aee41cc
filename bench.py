#!/usr/bin/env python3
"""Benchmark of the Vine5LinkMovingBase hot path on MI355X (BASELINE.json: env-steps/sec at 16384 envs per GPU,
PPO iters/sec).

    python bench.py --gpus 1 --steps K --warmup W [--mode ppo|env] [--num-envs 16384]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One rank per GPU; every rank owns `num_envs` independent envs (weak scaling, BASELINE configs 3/4).
mode=ppo (default once the learner is available): a "step" is one full PPO iteration = horizon x [policy
  inference + env step] + GAE + mini-epoch updates with the RCCL gradient all-reduce; value = env-steps/sec of the
  whole job (N x horizon x ranks / iteration time), ppo_iters_per_sec reported beside it.
mode=env: a "step" is one VecTask.step over resident random actions (the hand-written HIP kernel alone).
Rank 0 prints ONE JSON line.  The `roofline` object prices the env-step kernel (`roofline.kernel`: four lanes per env up
to 16384 envs, one lane per env beyond) with HIP events recorded on the launching stream inside the timed region; `cpu_baseline` times the CPU oracle
(oracle/, OpenMP over envs) on a bounded sample of the same workload on this host's cores.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

ALGO_BYTES_PER_ENV_STEP = {28: 320, 18: 280}   # reads 112 B + writes 205 B (obs 28) -> 320 B
HBM_PEAK_GBS = 8000.0                           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# MI355X_MICROARCH.md cycle table: `v_fma_f32` (wave64) occupies a SIMD-32 for 2 cycles (4 only for a lone wave's own stream):
# 256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1.2288e12 wave-instructions/s; = 64 FLOP/clk/SIMD = 157.3 TFLOP/s of fp32 FMA.
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2.0
FP32_VECTOR_PEAK_TFLOPS = 157.3
ALGO_FLOPS_PER_ENV_STEP = 27149.0               # oracle counting build: 40 F_sub (663) + 4 F_act (61) + F_post (385); DESIGN.md section 7


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None)
    p.add_argument("--warmup", type=int, default=None)
    p.add_argument("--mode", choices=["env", "ppo"], default=None)
    p.add_argument("--num-envs", type=int, default=16384,
                   help="envs per GPU with --scaling weak (BASELINE config 3: 16384); TOTAL envs with --scaling strong")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="weak: every rank owns --num-envs envs and its own 32768-sample minibatches (BASELINE configs "
                        "3/4); strong: --num-envs envs and every 32768-sample minibatch are split over the ranks "
                        "(rank r owns envs [r n/W, (r+1) n/W) of the SAME batch: same seed, VineConfig.env_id_offset)")
    p.add_argument("--randomize", type=int, default=1, help="task.vine_randomize (task YAML default: True)")
    p.add_argument("--obs-type", default="POS_AND_FD_VEL_AND_OBJ_INFO")
    p.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true", help="ppo mode: skip the run in the other update precision")
    p.add_argument("--no-saturated", action="store_true", help="skip the 2^20-env run of the step kernel")
    p.add_argument("--no-other-configs", action="store_true",
                   help="skip the env-step timings of the other BASELINE configs (4096 free-space, shelf, default pipe)")
    p.add_argument("--override", action="append", default=[],
                   help="extra Hydra-style override (repeatable), applied last, e.g. --override task.env.CREATE_SHELF=True: "
                        "profiling of the other task configurations; the default line never passes any")
    p.add_argument("--minibatch-size", type=int, default=None, help="ppo mode: override train.params.config.minibatch_size")
    p.add_argument("--force-multi-gpu-path", action="store_true",
                   help="ppo mode on ONE GPU: run the multi-rank code path (multi_gpu=True, world size 1: two graph replays "
                        "per optimiser step with the eager RCCL all-reduce between them) -- the per-rank cost of a shard, "
                        "used by scripts/scaling_projection.py; never part of the default line")
    p.add_argument("--no-graph", action="store_true", help="ppo mode: eager rollout instead of hipGraph replay")
    p.add_argument("--amp", choices=["fp16", "bf16", "off"], default=None,
                   help="ppo mode: update precision: off = fp32; fp16 (the packaged default, mixed_precision: True as in the "
                        "reference YAML) = hand-written mixed precision with device-side loss scaling when the library is "
                        "built for fp16 operands (default build), torch autocast otherwise; bf16 likewise for a "
                        "-DVINE_LP_BF16 build")
    return p.parse_args()


def make_env(args, rank, device_index, world=1, extra_overrides=()):
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    strong = getattr(args, "scaling", "weak") == "strong" and world > 1
    n_rank = args.num_envs // world if strong else args.num_envs
    ov = ["task=Vine5LinkMovingBase", "num_envs=%d" % n_rank, "vine_randomize=%s" % bool(args.randomize),
          "OBSERVATION_TYPE=%s" % args.obs_type, "headless=True",
          "task.env.CREATE_PIPE=False",      # default task YAML except CREATE_PIPE / CAPTURE_VIDEO
          "sim_device=cuda:%d" % device_index,
          "rl_device=cuda:%d" % device_index, "multi_gpu=%s" % (args.gpus > 1)]
    cfg = load_config(overrides=ov + list(extra_overrides) + list(getattr(args, "override", [])))
    if getattr(args, "minibatch_size", None):
        cfg["train"]["params"]["config"]["minibatch_size"] = int(args.minibatch_size)
    if strong:
        # one batch of --num-envs envs cut into W shards: same seed everywhere, the RNG keyed by the global env id, and
        # the global minibatch (PY:80: 32768 samples) split evenly so that the optimiser takes the same number of steps
        if args.num_envs % world or cfg["train"]["params"]["config"]["minibatch_size"] % world:
            raise SystemExit("--scaling strong: num_envs and minibatch_size must be divisible by the number of ranks")
        cfg["task"]["seed"] = 42
        cfg["task"]["env"]["envIdOffset"] = rank * n_rank
        cfg["train"]["params"]["config"]["minibatch_size"] //= world
    else:
        cfg["task"]["seed"] = 42 + 2 * rank        # train.py:78 + utils.py:50: the rank is added twice
    dev = "cuda:%d" % device_index
    env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device=dev, sim_device=dev,
                                                  graphics_device_id=device_index, headless=True,
                                                  virtual_screen_capture=False, force_render=False)
    return env, cfg


def kernel_instance(env):
    """Name of the step-kernel instantiation `env` launches, as rocprofv3 prints it: template arguments = observation
    layout (0: 28 columns, 1: 18), vine_randomize, obstacles (bit 0 shelf, bit 1 pipe)."""
    from vine_robot_isaacgymenvs_amd import abi
    c = env._vcfg
    obst = (1 if c.has_flag(abi.FLAG_CREATE_SHELF) else 0) | (2 if c.has_flag(abi.FLAG_CREATE_PIPE) else 0)
    return "%s<%d, %s, %d>" % (env.step_kernel_name, c.obs_type, "true" if c.has_flag(abi.FLAG_VINE_RANDOMIZE) else "false", obst)


def _pmc_summary(kernel):
    """Latest committed PMC summary (profiles/rNN/env_step_*_pmc_summary.json, written by scripts/pmc_summary.py) whose
    kernel-trace row is the instantiation `kernel` (kernel_instance(): e.g. "vine_step_quad_kernel<0, true, 0>"; the
    round-1/2 summaries predate the obstacle template argument of the four-lane kernel: "<0, true>" matches there)."""
    import glob
    old = kernel.rsplit(",", 1)[0] + ">" if kernel.endswith(", 0>") and "quad" in kernel else None
    # newest round first, and within a round the newest summary of that instantiation (file modification order is not
    # preserved by git: the version tag in the file name decides -- later tags sort later)
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "env_step_*pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
        except ValueError:
            continue
        name = d.get("kernel_stats", {}).get("name", "")
        # (since round 5 the four-lane kernel has a fourth template argument -- the policy head / bookkeeping inside -- and the
        # plain step is its `false` instantiation)
        if kernel in name or (kernel[:-1] + ", false>") in name or (old and old in name):
            return f, d
    return None, None


def pmc_traffic(kernel):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/: separate FETCH_SIZE and WRITE_SIZE passes, KiB units), with the gfx950 correction of
    MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies 128-B read requests at 64 B, so reads are doubled; WRITE_SIZE
    is exact.  Both factors were re-measured for this kernel's own access widths (4-B-per-lane SoA loads/stores,
    float2/float4 row stores) on known byte counts: scripts/ubench/pmc_calib.hip, profiles/r01/pmc_calibration.txt.
    bench.py cannot collect PMC itself: the figure belongs to the committed profile named in `traffic_source`."""
    f, d = _pmc_summary(kernel)
    try:
        return (2.0 * d["FETCH_SIZE"]["mean_per_launch"] + d["WRITE_SIZE"]["mean_per_launch"]) * 1024.0, os.path.relpath(f, REPO)
    except (KeyError, TypeError):
        return None, None


def trace_kernel_us(kernel):
    """Average duration of this instantiation in the committed kernel trace of its PMC summary (the figure the live
    `kernel_ms` has to agree with)."""
    _f, d = _pmc_summary(kernel)
    try:
        return d["kernel_stats"]["avg_ns"] / 1e3
    except (KeyError, TypeError):
        return None


def pmc_valu_per_wave(kernel):
    """VALU instructions one wave issues per env step (SQ_INSTS_VALU / SQ_WAVES of the committed PMC pass)."""
    _f, d = _pmc_summary(kernel)
    try:
        return d["SQ_INSTS_VALU"]["mean_per_launch"] / d["SQ_WAVES"]["mean_per_launch"]
    except (KeyError, TypeError, ZeroDivisionError):
        return None


def saturated_env_rate(args, device_index, n_sat=1 << 20, steps=40):
    """The same kernel with enough envs to give every SIMD several waves (16384 envs are 256 waves for 1024 SIMDs):
    the throughput the kernel itself sustains, priced against HBM and against the fp32 VALU issue rate
    (VALU_ISSUE_PEAK: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction)."""
    import copy
    import torch
    a = copy.copy(args)
    a.num_envs = n_sat
    env, _ = make_env(a, 0, device_index)
    act = torch.rand((n_sat, 2), device=env.device) * 2 - 1
    for _ in range(5):
        env._native_step(act, env.obs_buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        env._native_step(act, env.obs_buf)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    rate = n_sat / (ms * 1e-3)
    gbs = rate * ALGO_BYTES_PER_ENV_STEP[env.num_obs] / 1e9
    out = {"num_envs": n_sat, "kernel_ms": ms, "env_steps_per_sec": rate, "achieved_GBs": gbs,
           "hbm_frac": gbs / HBM_PEAK_GBS}
    out["kernel"] = env.step_kernel_name
    per_wave = pmc_valu_per_wave(kernel_instance(env))
    if per_wave:
        peak = VALU_ISSUE_PEAK
        out.update({"valu_insts_per_wave_step": per_wave, "valu_wave_insts_per_sec": rate / 64.0 * per_wave,
                    "valu_issue_peak": peak, "valu_issue_frac": rate / 64.0 * per_wave / peak})
    env.close()
    return out


def compute_roofline(kernel_name, n_envs, kernel_ms, instance=None):
    """What actually binds the env-step kernel (~85 flop per algorithmic byte, state resident in L2): the
    fp32 VALU.  Two fractions at the metric's own env count: VALU ISSUE (instructions per wave from the committed PMC
    pass of this kernel x waves per launch / kernel time, against VALU_ISSUE_PEAK) and fp32 FLOPs (the oracle's
    instrumented algorithmic flop count per env step, against the 157.3 TFLOP/s vector peak)."""
    lanes_per_env = 4 if "quad" in kernel_name else 1
    waves = (n_envs * lanes_per_env + 63) // 64
    per_wave = pmc_valu_per_wave(instance or (kernel_name + "<"))
    out = {"waves_per_launch": waves, "valu_issue_peak_wave_insts_per_sec": VALU_ISSUE_PEAK,
           "algorithmic_flops_per_env_step": ALGO_FLOPS_PER_ENV_STEP,
           "achieved_TFLOPs": ALGO_FLOPS_PER_ENV_STEP * n_envs / (kernel_ms * 1e-3) / 1e12,
           "peak_TFLOPs": FP32_VECTOR_PEAK_TFLOPS}
    out["flop_frac"] = out["achieved_TFLOPs"] / FP32_VECTOR_PEAK_TFLOPS
    if per_wave:
        rate = per_wave * waves / (kernel_ms * 1e-3)
        out.update({"valu_insts_per_wave_step": per_wave, "valu_wave_insts_per_sec": rate,
                    "valu_issue_frac": rate / VALU_ISSUE_PEAK})
    return out


# The other single-GPU BASELINE.json configurations and the reference's default obstacle, env step kernel only (HIP events
# around the C-ABI launch, resident random actions): configs[1] = 4096 envs with the README.md:63 free-space overrides,
# configs[4]'s per-GPU share = vine_randomize + CREATE_SHELF + ACTION_DELAY=1 at 16384 envs, and the task YAML's own default
# (CREATE_PIPE: True, TY:35).
OTHER_CONFIGS = [
    ("configs[1]: 4096 envs free-space reaching (README.md:63 overrides), fp32", 4096,
     ["task.env.CREATE_SHELF=False", "task.env.CREATE_PIPE=False", "OBSERVATION_TYPE=TIP_AND_CART_AND_OBJ_INFO",
      "task.env.maxEpisodeLength=100", "task.env.SUCCESS_DIST=0.04", "task.env.MIN_TARGET_Y=-0.4", "task.env.MAX_TARGET_Y=0.4",
      "task.env.MIN_TARGET_Z=0.55", "task.env.MAX_TARGET_Z=0.7", "RAIL_SOFT_LIMIT=0.25", "RAIL_P_GAIN=30", "RAIL_ACCELERATION=6"]),
    ("configs[4] per-GPU share: 16384 envs, vine_randomize + CREATE_SHELF + ACTION_DELAY=1", 16384,
     ["vine_randomize=True", "task.env.CREATE_SHELF=True", "task.env.CREATE_PIPE=False", "task.env.ACTION_DELAY=1"]),
    ("task YAML default obstacle: 16384 envs, CREATE_PIPE=True (TY:35)", 16384, ["task.env.CREATE_PIPE=True"]),
]


def other_config_rates(args, device_index, steps=200, mode="ppo"):
    """Per configuration: the env-step kernel alone (HIP events, as the headline's `env_only`) and -- mode ppo -- the whole
    PPO iteration on it (`ppo`: same quantity as the headline's `value`)."""
    import copy
    import torch
    rows = []
    for name, n, ov in OTHER_CONFIGS:
        a = copy.copy(args)
        a.num_envs = n
        if any(o.startswith("OBSERVATION_TYPE=") for o in ov):
            a.obs_type = [o for o in ov if o.startswith("OBSERVATION_TYPE=")][0].split("=")[1]
        env, cfg_o = make_env(a, 0, device_index, extra_overrides=ov)
        g = torch.Generator(device=env.device).manual_seed(7)
        pool = [torch.rand((n, 2), device=env.device, generator=g) * 2 - 1 for _ in range(16)]
        for i in range(60):                  # past the all-env reset of the first step; episodes de-synchronise
            env._native_step(pool[i % 16], env.obs_buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            env._native_step(pool[i % 16], env.obs_buf)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        rows.append({"config": name, "num_envs": n, "kernel": kernel_instance(env), "kernel_us": ms * 1e3,
                     "env_steps_per_sec": n / (ms * 1e-3),
                     "hbm_frac": ALGO_BYTES_PER_ENV_STEP[env.num_obs] * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        if mode == "ppo":
            try:
                from vine_robot_isaacgymenvs_amd.learning.bench_support import ppo_iteration_rate
                rows[-1]["ppo"] = ppo_iteration_rate(env, cfg_o, amp=args.amp, use_graphs=not args.no_graph)
            except Exception as err:      # a configuration the agent refuses must not cost the line
                rows[-1]["ppo"] = "unavailable: %s" % str(err)[:160]
        env.close()
    return rows


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, seconds, mode, cfg):
    """CPU restatement of the reference path on this host's usable cores, on a bounded sample of the same workload.
    The reference's own sim_device=cpu path needs Isaac Gym/PhysX and cannot run anywhere here (kind = "port"):
      * env step: oracle/ (float32, OpenMP over envs) through the same step API;
      * mode=ppo: ONE full PPO iteration (same horizon / minibatch / mini-epoch configuration) with that env and the
        stock PyTorch-CPU agent -- the same unit as `value` (whole-job env-steps/s)."""
    import copy

    import numpy as np
    import torch
    from oracle import vine_oracle as vo
    from vine_robot_isaacgymenvs_amd import abi
    lib = vo.load("f32", omp=True)
    cores = lib.vine_oracle_set_threads(usable_cores())
    ocfg = vo.default_config(lib, num_envs=args.num_envs)
    lib.vine_config_set_obs_type(ocfg, abi.OBS_TYPE_BY_NAME[args.obs_type], 1)
    ocfg.set_flag(abi.FLAG_VINE_RANDOMIZE, bool(args.randomize))
    env = vo.OracleEnv(ocfg, "f32", omp=True)
    rng = np.random.default_rng(42)
    acts = rng.uniform(-1, 1, (4, args.num_envs, 2)).astype(np.float32)
    env.step(acts[0])
    env_seconds = seconds if mode == "env" else min(3.0, seconds)
    t0 = time.perf_counter()
    n = 0
    while True:
        env.step(acts[n % 4])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= env_seconds or n >= 1000:
            break
    env_only = args.num_envs * n / dt
    # (a) of SURVEY 8(d): the same loop on ONE thread (a 2 s sample of a smaller batch: the per-env cost is what counts)
    lib.vine_oracle_set_threads(1)
    n1 = min(args.num_envs, 2048)
    ocfg1 = vo.default_config(lib, num_envs=n1)
    lib.vine_config_set_obs_type(ocfg1, abi.OBS_TYPE_BY_NAME[args.obs_type], 1)
    ocfg1.set_flag(abi.FLAG_VINE_RANDOMIZE, bool(args.randomize))
    env1 = vo.OracleEnv(ocfg1, "f32", omp=True)
    env1.step(acts[0][:n1])
    t0 = time.perf_counter()
    k1 = 0
    while time.perf_counter() - t0 < 2.0 and k1 < 1000:
        env1.step(acts[k1 % 4][:n1])
        k1 += 1
    single = n1 * k1 / (time.perf_counter() - t0)
    env1.close()
    lib.vine_oracle_set_threads(int(cores))
    try:
        flops = {k: (float(v) if not isinstance(v, str) else v) for k, v in vo.flop_counts(256, 10, bool(args.randomize)).items()}
    except Exception as err:      # the counting build is optional
        flops = "unavailable: %s" % str(err)[:80]
    out = {"value": env_only, "unit": "env-steps/s", "cores": int(cores), "kind": "port",
           "sample": "%d VecTask.step calls x %d envs (env step only, random actions), oracle/ float32 + OpenMP, %.1f s"
                     % (n, args.num_envs, dt)}
    if mode == "ppo":
        from oracle.oracle_vec_task import OracleVecTask
        from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
        torch.set_num_threads(int(cores))
        ccfg = copy.deepcopy(cfg)
        venv = OracleVecTask(ccfg["task"], precision="f32", omp=True)
        params = ccfg["train"]["params"]
        params["config"].update(device="cpu", multi_gpu=False, write_files=False, print_stats=False, use_graphs=False)
        agent = A2CAgent("cpu_baseline", params, vec_env=venv)
        agent.init_tensors()
        agent.obs = agent.env_reset()["obs"]
        t0 = time.perf_counter()
        play, upd, _ = agent.train_epoch()
        dt = time.perf_counter() - t0
        frames = agent.horizon_length * agent.num_actors
        out = {"value": frames / dt, "unit": "env-steps/s", "cores": int(cores), "kind": "port",
               "sample": "1 PPO iteration (%d envs x %d steps rollout with policy inference + %d optimiser steps), oracle/ "
                         "float32 + OpenMP env, PyTorch-CPU agent, %.1f s (rollout %.1f s, update %.1f s)"
                         % (args.num_envs, agent.horizon_length, agent.mini_epochs_num * agent.num_minibatches, dt, play, upd),
               "ppo_iters_per_sec": 1.0 / dt, "env_only_env_steps_per_sec": env_only}
        # config C1 (BASELINE.json configs[0], the reference's own CPU-runnable case): 64 envs, 1 PPO iteration,
        # minibatch 1024 (the default 32768 does not divide 64 x 16), same oracle env + PyTorch-CPU agent
        try:
            c1 = copy.deepcopy(cfg)
            c1["task"]["env"]["numEnvs"] = 64
            c1["task"]["env"].pop("envIdOffset", None)
            p1 = c1["train"]["params"]
            p1["config"].update(device="cpu", multi_gpu=False, write_files=False, print_stats=False, use_graphs=False,
                                num_actors=64, minibatch_size=1024)
            a1 = A2CAgent("cpu_c1", p1, vec_env=OracleVecTask(c1["task"], precision="f32", omp=True))
            a1.init_tensors()
            a1.obs = a1.env_reset()["obs"]
            a1.train_epoch()                                   # warm-up (allocations)
            t0 = time.perf_counter()
            a1.train_epoch()
            d1 = time.perf_counter() - t0
            out["config_c1"] = {"num_envs": 64, "minibatch": 1024, "env_steps_per_sec": 64 * a1.horizon_length / d1,
                                "ppo_iters_per_sec": 1.0 / d1}
        except Exception as err:
            out["config_c1"] = "unavailable: %s" % str(err)[:120]
    out["single_thread_env_only_env_steps_per_sec"] = single
    out["algorithmic_flops"] = flops
    return out


def main():
    args = parse_args()
    # stdout carries exactly ONE line, the JSON result: anything the libraries print on the way (e.g. the action /
    # observation spaces RLGPUEnv.get_env_info echoes, as the reference's does) goes to stderr
    real_stdout = sys.stdout
    sys.stdout = sys.stderr
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # Rehearsal hook (one-GPU box): VINE_BENCH_SHARE_DEVICE=1 maps every rank onto cuda:0 and uses gloo, so the
    # multi-process code path can be exercised without N GPUs.  Never set by the driver.
    share = os.environ.get("VINE_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.force_multi_gpu_path and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK=str(local_rank))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    try:
        from vine_robot_isaacgymenvs_amd.learning import a2c_continuous  # noqa: F401
        have_ppo = True
    except ImportError:
        have_ppo = False
    mode = args.mode or ("ppo" if have_ppo else "env")
    steps = args.steps if args.steps is not None else (10 if mode == "ppo" else 2000)
    warmup = args.warmup if args.warmup is not None else (3 if mode == "ppo" else 100)

    env, cfg = make_env(args, rank, local_rank, world)
    strong = args.scaling == "strong" and world > 1
    n = env.num_envs
    extra = {}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if mode == "env":
        g = torch.Generator(device=dev).manual_seed(42 + rank)
        pool = [torch.rand((n, 2), device=dev, generator=g) * 2 - 1 for _ in range(64)]
        for i in range(warmup):
            env.step(pool[i % 64])
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i][0].record()
            env._native_step(pool[i % 64], env.obs_buf)     # the C-ABI launch (VecTask.step minus dict marshalling)
            ev[i][1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
        units_per_step = n
    else:
        from vine_robot_isaacgymenvs_amd.learning.bench_support import run_ppo_bench
        elapsed, kernel_ms, units_per_step, extra = run_ppo_bench(env, cfg, args, steps, warmup, barrier, world, rank)

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        value = units_per_step * world * steps / elapsed
        algo_bytes = ALGO_BYTES_PER_ENV_STEP[env.num_obs] * n
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(kernel_instance(env))
        out = {
            "metric": ("env-steps/sec Vine5LinkMovingBase %d envs over %d GPUs" % (n * world, world)) if strong
                      else "env-steps/sec Vine5LinkMovingBase %d envs per GPU" % n,
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Vine5LinkMovingBase num_envs=%d per GPU, obs=%d, vine_randomize=%s, mode=%s "
                                   "(BASELINE.json configs[2]; x8 ranks = configs[3])"
                                   % (n, env.num_obs, bool(args.randomize), mode),
                       "mode": mode, "num_envs_per_gpu": n, "parallelism": "env-sharded dp%d" % world},
            # achieved / peak / frac: the mandated HBM pricing of the algorithmic bytes.  `bound` names what really limits the
            # kernel (fp32 VALU issue: ~85 flop per algorithmic byte, working set resident in L2) and `compute` prices it.
            "roofline": {"bound": "valu", "kernel": env.step_kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": kernel_ms,
                         "kernel_ms_method": ("HIP events around every eager launch of the timed region" if (mode == "env" or args.no_graph)
                                              else "64 launches replayed from a hipGraph, one event pair around 8 replays"),
                         "trace_kernel_us": trace_kernel_us(kernel_instance(env)),
                         "compute": compute_roofline(env.step_kernel_name, n, kernel_ms, kernel_instance(env))},
        }
        out.update(extra)
        if mode == "ppo":
            # env step kernel f32; rollout inference f32 (the reference's); the update takes the reference's
            # `mixed_precision: True` dtype (fp16 GEMM operands, f32 accumulation / state / loss / optimiser, GradScaler loss
            # scaling).  `other_precision` carries the same iteration with an all-f32 update, `extra_lp16_rollout` the
            # narrower 16-bit-rollout variant of round 2, `extra_native_f32_mfma_rollout` the rollout's LSTM step on the
            # native fp32 matrix-core instruction instead of exact bf16-piece products, `extra_split9_rollout` the 9-pair form of the (6-pair, two-accumulator) default.
            out["dtype"] = "f32 (env step, rollout inference); " + str(extra.get("update_precision", ""))
        if world == 1 and not args.no_saturated:
            out["roofline"]["saturated"] = saturated_env_rate(args, local_rank)
        if world == 1 and not args.no_other_configs:
            out["configs"] = other_config_rates(args, local_rank, mode=mode)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_baseline_seconds, mode, cfg)
        print(json.dumps(out), file=real_stdout, flush=True)
    env.close()
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
