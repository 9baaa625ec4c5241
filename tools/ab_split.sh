#!/bin/bash
# fused (rounds 1-3) vs split ring kernels on ONE box: same library, LIGHTLOAM_RING_SPLIT selects the path
fmt='import sys,json
d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_step"]
print("%.0f scans/s  %.2f ms/step | " % (d["value"], d["ms_per_step"]) + " ".join("%s=%.2f" % (a[2:10],b) for a,b in k.items()))'
for rep in 1 2; do
  for v in 0 1; do
    export LIGHTLOAM_RING_SPLIT=$v
    printf "split=%s " $v; timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>&1 | grep "^{\"metric" | tail -1 | python -c "$fmt"
  done
done
