#!/bin/bash
# round 6, call 5: k_ring_features with one / two waves per ring against the four-wave kernel: bit-exact suites with each library, then A/B
O=gpurun_out; mkdir -p $O
T="tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_ring_rows.py tests/test_gpu_variants.py"
for v in tree nw1 nw2; do
  if [ $v = tree ]; then unset LIGHTLOAM_HIP_LIB; else export LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_ab/lib$v.so; fi
  echo "== $v"; timeout 900 python -m pytest $T -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
done 2>&1 | tee $O/r06_05_pytest.log
unset LIGHTLOAM_HIP_LIB
bash tools/ab_once.sh > $O/r06_05_ab_s64.log 2>&1; cat $O/r06_05_ab_s64.log
bash tools/ab_once.sh --workload hdl64 > $O/r06_05_ab_hdl64.log 2>&1; cat $O/r06_05_ab_hdl64.log
