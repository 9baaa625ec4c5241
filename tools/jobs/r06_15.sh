#!/bin/bash
# round 6, call 15: final sources -- GPU suite, association soaks (two seeds, eight shapes), what the searches visit, then the profile round
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 > $O/r06_gpu_tests.log; tail -n 1 $O/r06_gpu_tests.log
for w in "synthetic 64" "hdl64 64" "synthetic 128"; do set -- $w; LL_STATS_LIB=_stats/libstats.so timeout 300 python tools/assoc_stats.py 64 $1 $2 2>&1 | grep -v amdgpu.ids; done > $O/r06_assoc_stats.txt; cat $O/r06_assoc_stats.txt
LL_SOAK_ALL_SHAPES=1 timeout 900 python3 tools/soak_hot_path.py 192 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|org_s" > $O/r06_soak_hot_path_all_shapes.log; tail -n 1 $O/r06_soak_hot_path_all_shapes.log
LL_SOAK_ALL_SHAPES=1 LL_SOAK_SEED=4242 timeout 900 python3 tools/soak_hot_path.py 192 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|org_s" > $O/r06_soak_hot_path_all_shapes_seed_4242.log; tail -n 1 $O/r06_soak_hot_path_all_shapes_seed_4242.log
bash tools/profile_round.sh r06 2>&1 | tail -3
cp profiles/pmc_traffic*.json profiles/sq_issue*.json $O/ 2>/dev/null
LL_SORT_CHECK=1 LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_stats/libsortcheck.so timeout 600 python3 tools/soak_extract.py 4 > $O/r06_soak_extract_sort_check.log 2>&1; tail -n 2 $O/r06_soak_extract_sort_check.log
LL_SORT_CHECK=1 LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_stats/libsortcheck.so timeout 600 python3 tools/soak_extract_s64.py 48 > $O/r06_soak_extract_s64_sort_check.log 2>&1; tail -n 2 $O/r06_soak_extract_s64_sort_check.log
timeout 300 python3 tools/bench_latency.py > $O/r06_latency.json 2>/dev/null
timeout 600 python3 tools/soak_frames.py 400 200 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|org_s" > $O/r06_soak_frames.log; tail -n 1 $O/r06_soak_frames.log
