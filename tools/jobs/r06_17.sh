#!/bin/bash
# round 6, call 17: loads in flight per lane in k_associate's nearest-neighbour scans only (1 / 3 / 4 against 2)
O=gpurun_out; mkdir -p $O
bash tools/ab_once.sh > $O/r06_17_ab_s64.log 2>&1; cat $O/r06_17_ab_s64.log
bash tools/ab_once.sh --workload hdl64 > $O/r06_17_ab_hdl64.log 2>&1; cat $O/r06_17_ab_hdl64.log
bash tools/ab_once.sh --rings 128 > $O/r06_17_ab_s128.log 2>&1; cat $O/r06_17_ab_s128.log
