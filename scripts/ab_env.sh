#!/bin/bash
# A/B of ONE environment switch on one box, alternating, whole PPO iterations (bench.py, no extras):
#   scripts/ab_env.sh VINE_TRUNK_HOIST 0 1 [rounds] [extra bench args]   -> prints value / rollout / update per run
VAR=$1; A=$2; B=$3; ROUNDS=${4:-3}; shift 4 2>/dev/null || shift $#
for r in $(seq $ROUNDS); do
  for v in $A $B; do
    env $VAR=$v python bench.py --mode ppo --steps 20 --warmup 3 --no-cpu-baseline --no-saturated --no-secondary --no-other-configs "$@" 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$VAR=$v value %.0f ms %.3f rollout %.3f update %.3f' % (d['value'], d['ms_per_step'], d['rollout_ms'], d['update_ms']))"
  done
done
