#!/bin/bash
# round 6, call 4: k_associate variants (scan unroll 4, all near entries in the first traversal, lazy syncs) on three workloads; two-stream piece counts
O=gpurun_out; mkdir -p $O
bash tools/ab_once.sh > $O/r06_04_ab_s64.log 2>&1; cat $O/r06_04_ab_s64.log
bash tools/ab_once.sh --workload hdl64 > $O/r06_04_ab_hdl64.log 2>&1; cat $O/r06_04_ab_hdl64.log
bash tools/ab_once.sh --rings 128 > $O/r06_04_ab_s128.log 2>&1; cat $O/r06_04_ab_s128.log
fmt='import sys,json
d=json.loads(sys.stdin.read()); a=d["roofline"].get("association_stage") or {}
print("%.0f scans/s  %.2f ms/step | stage two-stream %s one-stream %s" % (d["value"], d["ms_per_step"], a.get("two_stream_ms_per_step"), a.get("one_stream_ms_per_step")))'
for p in 2 3 4 6 8; do
  printf "pieces %d: " $p; LIGHTLOAM_TWO_STREAM=1 LIGHTLOAM_TS_PIECES=$p timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "$fmt"
done | tee $O/r06_04_two_stream_pieces.log
printf "one stream: " | tee -a $O/r06_04_two_stream_pieces.log; timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "$fmt" | tee -a $O/r06_04_two_stream_pieces.log
