#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_ring_features (and the other kernels) for every library in _ab/*.so and the tree's:
#   gpurun -- 'bash tools/pmc_ab.sh [bench args]'   -> gpurun_out/pmc_ab.txt
ROOT=$(pwd); O=$ROOT/gpurun_out/pmc_ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --calibrate --batch 4096 $@"
for v in $ROOT/_ab/*.so tree; do
  if [ $v = tree ]; then unset LIGHTLOAM_HIP_LIB; n=tree; else export LIGHTLOAM_HIP_LIB=$v; n=$(basename $v .so); fi
  rm -rf $O/f_$n $O/w_$n
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/f_$n -o fetch -- $B > $O/f_$n.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/w_$n -o write -- $B > $O/w_$n.log 2>&1
  echo "== $n"; python3 tools/pmc_traffic.py $O/f_$n $O/w_$n --batch 4096 --rings 64 --source-digest x | python3 -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d['kernels'].items():
    if v['hbm_bytes_per_scan'] > 1000: print('  %-22s read %.3f MB  write %.3f MB per scan' % (k, v['hbm_read_bytes_per_launch'] / 4096 / 1e6, v['hbm_write_bytes_per_launch'] / 4096 / 1e6))"
  rm -rf $O/f_$n $O/w_$n
done
