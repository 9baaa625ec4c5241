#!/bin/bash
# round 6, call 6: the whole GPU suite + the association / frame soaks on the one-traversal k_associate
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 > $O/r06_gpu_tests.log; tail -3 $O/r06_gpu_tests.log
LL_SOAK_ALL_SHAPES=1 timeout 900 python3 tools/soak_hot_path.py 192 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|org_s" > $O/r06_soak_hot_path_all_shapes.log; tail -2 $O/r06_soak_hot_path_all_shapes.log
LL_SOAK_ALL_SHAPES=1 LL_SOAK_SEED=4242 timeout 900 python3 tools/soak_hot_path.py 192 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|org_s" > $O/r06_soak_hot_path_all_shapes_seed_4242.log; tail -2 $O/r06_soak_hot_path_all_shapes_seed_4242.log
timeout 600 python3 tools/soak_frames.py 400 200 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|org_s" > $O/r06_soak_frames.log; tail -2 $O/r06_soak_frames.log
