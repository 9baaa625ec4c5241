#!/bin/bash
# round 6, call 13: the same with the corner queries' entries 3, 4 compiled out and a wave-uniform skip for the plane queries; T = 24 / 32 (tree) / 64
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_associate_edge.py tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_ring_rows.py tests/test_gpu_variants.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/r06_13_pytest.log
bash tools/ab_once.sh > $O/r06_13_ab_s64.log 2>&1; cat $O/r06_13_ab_s64.log
bash tools/ab_once.sh --workload hdl64 > $O/r06_13_ab_hdl64.log 2>&1; cat $O/r06_13_ab_hdl64.log
bash tools/ab_once.sh --rings 128 > $O/r06_13_ab_s128.log 2>&1; cat $O/r06_13_ab_s128.log
