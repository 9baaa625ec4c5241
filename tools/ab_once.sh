#!/bin/bash
# one bench.py run per library in _ab/*.so and for the tree's (see tools/ab_bench.sh), kernel times only
fmt='import sys,json
d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_step"]
print("%.0f scans/s  %.2f ms/step | " % (d["value"], d["ms_per_step"]) + " ".join("%s=%.2f" % (a[2:10],b) for a,b in k.items()))'
for v in $GRAFT_REPO_ROOT/_ab/*.so tree; do
  if [ $v = tree ]; then unset LIGHTLOAM_HIP_LIB; else export LIGHTLOAM_HIP_LIB=$v; fi
  printf "%-10s " "$(basename $v .so):"; timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>&1 | grep -E "^\{\"metric|Error|error|assert" | tail -1 | python -c "$fmt" 2>&1 | tail -1
done
