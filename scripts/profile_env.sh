#!/bin/bash
# rocprofv3 characterisation of the env-step kernel (run on the GPU box via gpurun).
# Usage: scripts/profile_env.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--mode env --steps 300 --warmup 30 --no-cpu-baseline --no-saturated $@"
# 1) kernel trace + stats (the same command as the bench line)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
# 2) counters, each group in its own pass (no trace domains besides kernel-trace)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
