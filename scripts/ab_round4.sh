#!/bin/bash
# Same-box A/B of round 4's changes to the PPO iteration: every round-4 switch off (= round 3's launch sequence) against the
# defaults, alternating, 20 timed iterations each.   Usage (GPU box): bash scripts/ab_round4.sh > gpurun_out/ab_round4.txt
OFF="VINE_DATASET_FUSED=0 VINE_MLP3_PREP=0 VINE_COPY_SCATTER=0 VINE_SEQ_FWD_WCACHE=0 VINE_SYNC_EACH_ITER=1 VINE_ROLLOUT_COPYBATCH=0 VINE_BWD_PHASES=0 VINE_TRUNK_PHASES=0 VINE_ROLLOUT_FIN_RIDE=0 VINE_MLP3_F32_SPLIT=0 VINE_LSTM_STEP_NSPLIT=0"
for rep in 1 2 3; do
  for mode in off on; do
    if [ $mode = off ]; then export $OFF; else unset VINE_DATASET_FUSED VINE_MLP3_PREP VINE_COPY_SCATTER VINE_SEQ_FWD_WCACHE VINE_SYNC_EACH_ITER VINE_ROLLOUT_COPYBATCH VINE_BWD_PHASES VINE_TRUNK_PHASES VINE_ROLLOUT_FIN_RIDE VINE_MLP3_F32_SPLIT VINE_LSTM_STEP_NSPLIT; fi
    python bench.py --no-cpu-baseline --no-saturated --no-secondary --no-other-configs --steps 20 > /tmp/ab_line.json 2>/dev/null
    python - <<PY
import json
d = json.loads(open("/tmp/ab_line.json").read().strip().splitlines()[-1])
print("round-4 switches %-3s  %.3f M env-steps/s  %.3f ms/iteration  rollout %.3f  update %.3f" % ("$mode", d["value"] / 1e6, d["ms_per_step"], d["rollout_ms"], d["update_ms"]))
PY
  done
done
