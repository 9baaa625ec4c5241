#!/bin/bash
# round 6, call 19: more soak on the final sources -- 16 further extract seeds, a third association seed over eight shapes
O=gpurun_out; mkdir -p $O
timeout 900 python3 tools/soak_extract.py 16 400 > $O/r06_soak_extract_seeds_400_415.log 2>&1; tail -n 1 $O/r06_soak_extract_seeds_400_415.log
LL_SOAK_ALL_SHAPES=1 LL_SOAK_SEED=777 timeout 900 python3 tools/soak_hot_path.py 192 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|org_s" > $O/r06_soak_hot_path_all_shapes_seed_777.log; tail -n 1 $O/r06_soak_hot_path_all_shapes_seed_777.log
