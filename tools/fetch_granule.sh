#!/bin/bash
# gpurun -- 'bash tools/fetch_granule.sh' -> gpurun_out/fetch_granule.txt: timing of 16-byte loads at strides of 16..256 bytes and what FETCH_SIZE /
# the TCC_EA0 read-request counters report per load for each stride (which granule the fabric moves, which one the counter tallies).
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
[ -x tools/ubench/fetch_granule ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_granule tools/ubench/fetch_granule.hip
tools/ubench/fetch_granule 3 > $O/fetch_granule.txt 2>&1
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum TCC_REQ_sum" "TCC_MISS_sum TCC_HIT_sum"; do
  rm -rf $O/fg
  rocprofv3 --pmc $set --kernel-trace -f csv -d $O/fg -o c -- tools/ubench/fetch_granule 1 > $O/fg.log 2>&1
  python3 - "$O/fg" "$set" >> $O/fetch_granule.txt <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = acc.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("# pmc", sys.argv[2])
for did, d in acc.items():
    print(did, d["k"][:40], {k: v for k, v in d.items() if k != "k"})
PY
done
rm -rf $O/fg
cat $O/fetch_granule.txt
