#!/bin/bash
# A/B/... on ONE box: bench.py with every library in _ab/*.so (built from other revisions or with other flags:
# tools/make_ab_baseline.sh) and with the tree's library ("tree"), alternating, so that box-to-box and run-to-run
# variation does not hide a few-percent difference.
fmt='import sys,json
d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_step"]
print("%.0f scans/s  %.2f ms/step | " % (d["value"], d["ms_per_step"]) + " ".join("%s=%.2f" % (a[2:10],b) for a,b in k.items()))'
for rep in 1 2; do
  for v in $GRAFT_REPO_ROOT/_ab/*.so tree; do
    if [ $v = tree ]; then unset LIGHTLOAM_HIP_LIB; else export LIGHTLOAM_HIP_LIB=$v; fi
    printf "%-10s " "$(basename $v .so):"; timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>&1 | grep "^{\"metric" | tail -1 | python -c "$fmt"
  done
done
