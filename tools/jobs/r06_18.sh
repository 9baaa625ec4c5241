#!/bin/bash
# round 6, call 18: L2 hit rate / L1->L2 read latency per kernel on the final sources (S64 and S128), rocprofv3 kernel stats of the 128-ring step
O=gpurun_out; mkdir -p $O
bash tools/l2_latency_counters.sh; cp $O/l2_latency.txt $O/r06_l2_hit_and_read_latency.txt
bash tools/l2_latency_counters.sh --rings 128; cp $O/l2_latency.txt $O/r06_l2_hit_and_read_latency_s128.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/prof_r06_s128 -o r06s -- python3 bench.py --rings 128 --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_r06_s128.log 2>&1
python3 tools/rocpd_summary.py $O/prof_r06_s128/r06s_results.db > $O/r06_kernel_stats_s128.txt; rm -rf $O/prof_r06_s128
head -12 $O/r06_kernel_stats_s128.txt; grep -E "k_associate|k_build_grid|k_ring_features" $O/r06_l2_hit_and_read_latency.txt
