#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes, kernel-trace only) of every kernel of real PPO iterations,
# beside the kernel trace of the PRODUCTION (graph-replayed) path of the same tree.  Run on the GPU box via gpurun:
#     scripts/profile_ppo_traffic.sh <tag> [extra bench args]
# -> gpurun_out/ppo_traffic_<tag>/{kernel_stats.csv,pmc_summary.json,summary.txt}; committed as profiles/rNN/ppo_traffic_<tag>_<N>envs_*
# The counter passes run the iteration eagerly (--no-graph: the same kernels with the same arguments, one dispatch record
# each); the in-situ durations come from the graphed trace.
set -o pipefail
TAG=${1:-x}; shift
OUT=gpurun_out/ppo_traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--mode ppo --no-cpu-baseline --no-saturated --no-secondary --no-other-configs $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $COMMON --steps 8 --warmup 2 > $OUT/bench_trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $COMMON --steps 2 --warmup 1 --no-graph > $OUT/bench_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $COMMON --steps 2 --warmup 1 --no-graph > $OUT/bench_write.log 2>&1 &&
python3 scripts/pmc_traffic_summary.py $OUT > $OUT/summary.txt && rm -rf $OUT/pmc_fetch $OUT/pmc_write && cat $OUT/summary.txt
