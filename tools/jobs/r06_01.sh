#!/bin/bash
# round 6, call 1: the GPU suite on the new sources, then A/B against the round-start library (_ab/libA.so) on one box
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > $O/r06_01_pytest.log
tail -3 $O/r06_01_pytest.log
bash tools/ab_bench.sh > $O/r06_01_ab_s64.log 2>&1; cat $O/r06_01_ab_s64.log
LIGHTLOAM_TWO_STREAM=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('tree one-stream: %.0f scans/s %.2f ms' % (d['value'], d['ms_per_step']))" | tee -a $O/r06_01_ab_s64.log
for st in 4 3; do
  timeout 600 python bench.py --stream-input --input-stride $st --no-cpu-baseline > $O/r06_01_stream_s64_stride$st.json 2>/dev/null
  timeout 600 python bench.py --stream-input --rings 128 --input-stride $st --no-cpu-baseline > $O/r06_01_stream_s128_stride$st.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_01_stream_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.0f scans/s' % d['value'], d['stream'])
    except Exception as e: print(f, 'failed', e)
P
timeout 300 python bench.py --input-stride 3 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > $O/r06_01_bench_stride3.json
python -c "
import json; d=json.load(open('gpurun_out/r06_01_bench_stride3.json')); print('stride3 resident: %.0f scans/s' % d['value'], d['roofline']['kernel_ms_per_step'])"
