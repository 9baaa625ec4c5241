#!/bin/bash
# round 6, call 8: k_associate compiled for fewer waves per SIMD with more loads in flight per lane
O=gpurun_out; mkdir -p $O
bash tools/ab_once.sh > $O/r06_08_ab_s64.log 2>&1; cat $O/r06_08_ab_s64.log
bash tools/ab_once.sh --workload hdl64 > $O/r06_08_ab_hdl64.log 2>&1; cat $O/r06_08_ab_hdl64.log
bash tools/ab_once.sh --rings 128 > $O/r06_08_ab_s128.log 2>&1; cat $O/r06_08_ab_s128.log
