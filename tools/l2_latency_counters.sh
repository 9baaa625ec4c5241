#!/bin/bash
# L2 hit rate and L1 -> L2 read latency per kernel of the headline workload (two counter passes, --kernel-trace only):
#   gpurun -- 'bash tools/l2_latency_counters.sh [bench args]'   -> gpurun_out/l2_latency.txt
ROOT=$(pwd); O=$ROOT/gpurun_out/l2lat; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 4096 $@"
rm -rf $O/hit $O/lat
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -f csv -d $O/hit -o c -- $B > $O/hit.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --kernel-trace -f csv -d $O/lat -o c -- $B > $O/lat.log 2>&1
{ python3 tools/sq_summary.py $O/hit; python3 tools/sq_summary.py $O/lat; } > $ROOT/gpurun_out/l2_latency.txt
rm -rf $O/hit $O/lat
