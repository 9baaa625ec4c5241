#!/bin/bash
# round 6, call 9: the profile round on the FINAL sources (counter files stamped with their digest), sort-check soaks, latency, a third soak seed
bash tools/profile_round.sh r06 2>&1 | tail -3
O=gpurun_out
cp profiles/pmc_traffic*.json profiles/sq_issue*.json $O/ 2>/dev/null
LL_SORT_CHECK=1 LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_stats/libsortcheck.so timeout 600 python3 tools/soak_extract.py 4 > $O/r06_soak_extract_sort_check.log 2>&1; tail -n 2 $O/r06_soak_extract_sort_check.log
LL_SORT_CHECK=1 LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_stats/libsortcheck.so timeout 600 python3 tools/soak_extract_s64.py 48 > $O/r06_soak_extract_s64_sort_check.log 2>&1; tail -n 2 $O/r06_soak_extract_s64_sort_check.log
timeout 300 python3 tools/bench_latency.py > $O/r06_latency.json 2>/dev/null
LL_SOAK_ALL_SHAPES=1 LL_SOAK_SEED=777 timeout 900 python3 tools/soak_hot_path.py 192 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|org_s" > $O/r06_soak_hot_path_all_shapes_seed_777.log; tail -n 1 $O/r06_soak_hot_path_all_shapes_seed_777.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 > $O/r06_gpu_tests.log; tail -n 1 $O/r06_gpu_tests.log
