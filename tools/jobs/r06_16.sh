#!/bin/bash
# round 6, call 16: k_organize's first-kept search over 2 / 3 / 4 tiles per round
O=gpurun_out; mkdir -p $O
for v in os2 os4; do
  echo "== $v"; LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_ab/lib$v.so timeout 600 python -m pytest tests/test_gpu_a1_edges.py tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -2
done 2>&1 | tee $O/r06_16_pytest.log
bash tools/ab_bench.sh > $O/r06_16_ab_s64.log 2>&1; cat $O/r06_16_ab_s64.log
bash tools/ab_once.sh --workload hdl64 > $O/r06_16_ab_hdl64.log 2>&1; cat $O/r06_16_ab_hdl64.log
