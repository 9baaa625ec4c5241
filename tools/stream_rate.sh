#!/bin/bash
# The chip's streaming ceiling on the library's own strides:
#   gpurun -- 'bash tools/stream_rate.sh'  ->  gpurun_out/stream_rate.json (+ stream_rate_inflight.txt: measured L1->L2 read latency and
#   requests in flight per CU of the best read / copy variants, from TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum)
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
[ -x tools/ubench/stream_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/stream_rate tools/ubench/stream_rate.hip
tools/ubench/stream_rate "$@" > $O/stream_rate.json 2> $O/stream_rate.err
rm -rf $O/sr_lat
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --kernel-trace -f csv -d $O/sr_lat -o c -- tools/ubench/stream_rate --reps 1 --wgs 8 > $O/sr_lat.log 2>&1
python3 - "$O/sr_lat" > $O/stream_rate_inflight.txt <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
kt = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kt[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
acc = collections.OrderedDict()
for r in rows:
    d = acc.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"]})
    if "End_Timestamp" in r: kt.setdefault(r["Dispatch_Id"], (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9)
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("# dispatch kernel seconds lat_cycles_per_req reqs_in_flight_per_CU   (requests are 64 B on gfx950 for wide loads: x2 for 128-B lines)")
for did, d in acc.items():
    req = d.get("TCP_TCC_READ_REQ_sum", 0.0); lat = d.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0); t = kt.get(did, 0.0)
    if req <= 0 or t <= 0: continue
    cyc = t * 2.4e9
    print(did, d["k"][:60], "%.6f" % t, "%.0f" % (lat / req), "%.1f" % (lat / cyc / 256.0))
PY
rm -rf $O/sr_lat
tail -c 400 $O/stream_rate.json
