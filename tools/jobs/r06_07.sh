#!/bin/bash
# round 6, call 7: the profile round (rocprofv3 kernel stats + counter passes for four workloads, bench lines, stream input, short soaks)
bash tools/profile_round.sh r06 2>&1 | tail -5
O=gpurun_out
cp profiles/pmc_traffic*.json profiles/sq_issue*.json $O/ 2>/dev/null
LL_SORT_CHECK=1 LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_stats/libsortcheck.so timeout 600 python3 tools/soak_extract.py 4 > $O/r06_soak_extract_sort_check.log 2>&1; tail -2 $O/r06_soak_extract_sort_check.log
LL_SORT_CHECK=1 LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_stats/libsortcheck.so timeout 600 python3 tools/soak_extract_s64.py 48 > $O/r06_soak_extract_s64_sort_check.log 2>&1; tail -2 $O/r06_soak_extract_s64_sort_check.log
timeout 300 python3 tools/bench_latency.py > $O/r06_latency.json 2>/dev/null; tail -c 400 $O/r06_latency.json
