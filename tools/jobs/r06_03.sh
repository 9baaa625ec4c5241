#!/bin/bash
# round 6, call 2: the one-traversal k_associate -- index-exact suites, what the searches visit, A/B against the round-start library
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_associate_edge.py tests/test_gpu_ring_rows.py tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_distortion.py tests/test_gpu_odometry.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > $O/r06_03_pytest.log
tail -4 $O/r06_03_pytest.log
for w in "synthetic 64" "hdl64 64" "synthetic 128"; do set -- $w; LL_STATS_LIB=_stats/libstats.so timeout 300 python tools/assoc_stats.py 64 $1 $2 2>&1 | grep -v amdgpu.ids; done | tee $O/r06_03_assoc_stats.txt
bash tools/ab_once.sh > $O/r06_03_ab_s64.log 2>&1; cat $O/r06_03_ab_s64.log
bash tools/ab_once.sh --workload hdl64 > $O/r06_03_ab_hdl64.log 2>&1; cat $O/r06_03_ab_hdl64.log
bash tools/ab_once.sh --rings 128 > $O/r06_03_ab_s128.log 2>&1; cat $O/r06_03_ab_s128.log
