#!/bin/bash
# round 6, call 10: what the driver runs at round end, as it runs it
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric' > $O/r06_bench_as_the_driver_runs_it.json
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_as_the_driver_runs_it.json')); r=d['roofline']
print('%.1f k scans/s %.2f ms; bound %s; kernel %s frac %.3f traffic %s; cpu %s' % (d['value']/1e3, d['ms_per_step'], r['bound'], r['kernel'], r['frac'], r['traffic'], d['cpu_baseline']['value']))"
