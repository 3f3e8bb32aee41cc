#!/bin/bash
# rocprofv3 kernel-trace + SQ / FETCH / WRITE counter passes of the env-step kernel in ANOTHER task configuration
# (run on the GPU box via gpurun).  Usage: scripts/profile_env_cfg.sh <tag> [bench args, e.g. --override task.env.CREATE_SHELF=True]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--mode env --steps 200 --warmup 60 --no-cpu-baseline --no-saturated --no-other-configs $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1 &&
python3 scripts/pmc_summary.py $OUT vine_step > $OUT/pmc_summary.json && cat $OUT/pmc_summary.json
