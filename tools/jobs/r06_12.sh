#!/bin/bash
# round 6, call 12: plane queries take the near cells as rows when their own cell holds fewer than T points: T = 0 / 16 / 32 (tree) / 64 / always
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_associate_edge.py tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_ring_rows.py tests/test_gpu_variants.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/r06_12_pytest.log
bash tools/ab_once.sh > $O/r06_12_ab_s64.log 2>&1; cat $O/r06_12_ab_s64.log
bash tools/ab_once.sh --workload hdl64 > $O/r06_12_ab_hdl64.log 2>&1; cat $O/r06_12_ab_hdl64.log
bash tools/ab_once.sh --rings 128 > $O/r06_12_ab_s128.log 2>&1; cat $O/r06_12_ab_s128.log
