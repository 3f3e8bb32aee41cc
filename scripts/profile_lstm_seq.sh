#!/bin/bash
# rocprofv3 characterisation of the two persistent LSTM sequence kernels at the update's shapes, cold operands
# (scripts/ubench/lstm_pmc.py rotates over operand sets larger than L2 + MALL).  Counters in separate passes.
# Usage: scripts/profile_lstm_seq.sh <tag>   -> gpurun_out/prof_<tag>/{trace,pmc_*}
TAG=${1:-lstm_seq}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/ubench/lstm_pmc.py > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 scripts/ubench/lstm_pmc.py > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/ubench/lstm_pmc.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/ubench/lstm_pmc.py > $OUT/pmc_write.log 2>&1
tail -2 $OUT/trace.log
