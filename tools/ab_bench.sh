#!/bin/bash
# A/B on ONE box: bench.py with the library in _ab/libA.so (built from another revision) and with the tree's library,
# alternating, so that box-to-box and run-to-run variation does not hide a few-percent difference.
fmt='import sys,json
d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_step"]
print("%.0f scans/s  %.2f ms/step | " % (d["value"], d["ms_per_step"]) + " ".join("%s=%.2f" % (a[2:6],b) for a,b in k.items()))'
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export LIGHTLOAM_HIP_LIB=$GRAFT_REPO_ROOT/_ab/libA.so; else unset LIGHTLOAM_HIP_LIB; fi
    echo -n "$v: "; timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "$fmt"
  done
done
