#!/bin/bash
# Workgroup-size sweep of the step kernel on the GPU box.
for t in ${VINE_SWEEP:-64 128 256 512}; do
  VINE_HIPCC_FLAGS="-DVINE_STEP_THREADS=$t" python3 -c "from vine_robot_isaacgymenvs_amd import native; native.build(force=True)"
  python3 bench.py --mode env --steps 1000 --warmup 100 --no-cpu-baseline --randomize 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('threads $t kernel_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'])"
done
python3 -c "from vine_robot_isaacgymenvs_amd import native; native.build(force=True)"
