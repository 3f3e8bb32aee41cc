#!/bin/bash
# SQ counters of every kernel of an eager PPO iteration (run on the GPU box via gpurun): scripts/profile_ppo_pmc.sh <tag>
# One counter group per pass, kernel-trace only.  Prints per kernel: launches, waves, instructions and the busy / wait
# cycle sums per launch (scripts/pmc_kernels.py).
set -o pipefail
TAG=${1:-r03}; shift
OUT=gpurun_out/ppo_pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--mode ppo --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-other-configs --no-secondary --no-saturated $@"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/bench_sq2.log 2>&1 &&
{ rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq3 -- python3 bench.py $ARGS > $OUT/bench_sq3.log 2>&1 || echo "(third pass -- matrix-pipe busy cycles, LDS conflicts -- not available)" >&2; } &&
python3 scripts/pmc_kernels.py $OUT > $OUT/summary.txt && rm -rf $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_sq3 && cat $OUT/summary.txt
