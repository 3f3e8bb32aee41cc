#!/usr/bin/env python3
"""Strong / weak scaling PROJECTION from one GPU (RCCL itself is unmeasured: the builder has one
MI355X).  Times `bench.py --mode ppo` at the per-rank shard sizes of W = 2 / 4 / 8 strong scaling (16384 envs and every
32768-sample minibatch split over the ranks) and of weak scaling, all on the multi-rank code path (multi_gpu=True with a
1-rank RCCL group; round 4: one graph replay per mini-epoch with the all-reduce of every optimiser step captured inside it when
the capture probe passes -- `update_graphs` in the output says which form ran; round 3: one graph per step, collective between
them), and adds
32 x an ASSUMED all-reduce time per iteration (1.63 MB of gradients + KL + overflow flag over xGMI; latency-bound).

    python scripts/scaling_projection.py [--steps 10] > profiles/r04/scaling_projection.json
"""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# assumed one-shot all-reduce time of the 1.63 MB block per optimiser step (us): ~10-15 us of launch + synchronisation
# latency plus 2 x (W-1)/W x 1.63 MB over (W-1) xGMI links at ~50 GB/s effective each.  NOT measured.
ALLREDUCE_US = {1: 0.0, 2: 20.0, 4: 25.0, 8: 30.0}


def run(num_envs, minibatch, forced, steps):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--mode", "ppo", "--steps", str(steps), "--warmup", "3",
           "--no-cpu-baseline", "--no-saturated", "--no-other-configs", "--no-secondary", "--num-envs", str(num_envs),
           "--minibatch-size", str(minibatch)]
    if forced:
        cmd.append("--force-multi-gpu-path")
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO)
    if out.returncode != 0:
        raise SystemExit("bench failed: %s\n%s" % (" ".join(cmd), out.stderr[-2000:]))
    d = json.loads(out.stdout.strip().splitlines()[-1])
    return {"num_envs": num_envs, "minibatch": minibatch, "multi_rank_path": forced, "ms_per_iteration": d["ms_per_step"],
            "rollout_ms": d["rollout_ms"], "update_ms": d["update_ms"], "update_graphs": d["ppo"]["hipgraphs_active"]["update"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    base = run(16384, 32768, False, a.steps)
    rows = {"1": base}
    weak = run(16384, 32768, True, a.steps)
    strong = {w: run(16384 // w, 32768 // w, True, a.steps) for w in (2, 4, 8)}
    opt_steps = 32
    table = []
    for w in (1, 2, 4, 8):
        ar_ms = opt_steps * ALLREDUCE_US[w] * 1e-3
        t_strong = base["ms_per_iteration"] if w == 1 else strong[w]["ms_per_iteration"] + ar_ms
        t_weak = base["ms_per_iteration"] if w == 1 else weak["ms_per_iteration"] + ar_ms
        table.append({"gpus": w, "assumed_allreduce_us_per_step": ALLREDUCE_US[w],
                      "strong": {"ms_per_iteration": t_strong, "env_steps_per_sec": 16384 * 16 / (t_strong * 1e-3),
                                 "speedup_vs_1": base["ms_per_iteration"] / t_strong},
                      "weak": {"ms_per_iteration": t_weak, "env_steps_per_sec": w * 16384 * 16 / (t_weak * 1e-3),
                               "efficiency": base["ms_per_iteration"] / t_weak}})
    print(json.dumps({"label": "PROJECTION from one MI355X: per-rank shard timings measured on the multi-rank code path "
                               "(1-rank RCCL group), all-reduce cost ASSUMED; RCCL over xGMI unmeasured",
                      "measured": {"single_gpu": base, "weak_shard_multi_rank_path": weak,
                                   "strong_shards_multi_rank_path": {str(k): v for k, v in strong.items()}},
                      "projection": table}, indent=1))


if __name__ == "__main__":
    main()
