# A/B helper for experiments on the GPU box: `source scripts/bench_ab.sh; run <label>` prints value / rollout / update of
# one short PPO bench run under the current environment (e.g. `VINE_UPD_GRAPH=step run step`).
run() { python bench.py --mode ppo --steps 10 --warmup 3 --no-cpu-baseline --no-saturated --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(\"$1 value\", round(d[\"value\"]), \"rollout_ms\", round(d[\"rollout_ms\"],2), \"update_ms\", round(d[\"update_ms\"],2))"; }
