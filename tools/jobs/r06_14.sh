#!/bin/bash
# round 6, call 14: corner queries over the 5 x 5 cells as five rows; one shared minimum behind the rows
O=gpurun_out; mkdir -p $O
for v in c5 onesync c5onesync; do
  echo "== $v"; LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_ab/lib$v.so timeout 600 python -m pytest tests/test_gpu_associate_edge.py tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -2
done 2>&1 | tee $O/r06_14_pytest.log
bash tools/ab_once.sh > $O/r06_14_ab_s64.log 2>&1; cat $O/r06_14_ab_s64.log
bash tools/ab_once.sh --workload hdl64 > $O/r06_14_ab_hdl64.log 2>&1; cat $O/r06_14_ab_hdl64.log
bash tools/ab_once.sh --rings 128 > $O/r06_14_ab_s128.log 2>&1; cat $O/r06_14_ab_s128.log
