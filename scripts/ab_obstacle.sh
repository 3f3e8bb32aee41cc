#!/bin/bash
# A/B of an env switch on the obstacle configurations (one box): env-step kernel alone under random actions (bench.py --mode env)
# and whole PPO iterations while the policy trains (100 iterations: the vine reaches the tube after ~60).
#   scripts/ab_obstacle.sh VINE_STEP_HALF_WAVES 0 1
VAR=$1; A=$2; B=$3
for cfgname in pipe shelf; do
  if [ $cfgname = pipe ]; then OV="--override task.env.CREATE_PIPE=True"; else OV="--override task.env.CREATE_SHELF=True --override task.env.CREATE_PIPE=False --override task.env.ACTION_DELAY=1"; fi
  for v in $A $B $A $B; do
    env $VAR=$v python bench.py --mode env --steps 400 --warmup 100 --no-cpu-baseline --no-saturated --no-other-configs $OV 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$cfgname $VAR=$v env-only kernel %.1f us' % (d['roofline']['kernel_ms']*1e3))"
  done
  for v in $A $B; do
    env $VAR=$v python bench.py --mode ppo --steps 100 --warmup 3 --no-cpu-baseline --no-saturated --no-secondary --no-other-configs $OV 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$cfgname $VAR=$v 100 training iterations: value %.0f ms %.3f rollout %.3f update %.3f' % (d['value'], d['ms_per_step'], d['rollout_ms'], d['update_ms']))"
  done
done
