#!/bin/bash
# A/B two builds of the HIP library on the GPU box: scripts/ab_build.sh "<flagsA>" "<flagsB>"
for v in A B; do
  if [ $v = A ]; then F="$1"; else F="$2"; fi
  VINE_HIPCC_FLAGS="$F" python3 -c "from vine_robot_isaacgymenvs_amd import native; native.build(force=True)"
  for r in 1 2; do
    python3 bench.py --mode env --steps 1000 --warmup 100 --no-cpu-baseline --randomize 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v [$F] norand kernel_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'])"
  done
done
python3 -c "from vine_robot_isaacgymenvs_amd import native; native.build(force=True)"
