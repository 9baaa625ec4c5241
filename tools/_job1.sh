cd $GRAFT_REPO_ROOT
bash tools/stream_rate.sh > gpurun_out/j1_stream.log 2>&1
python3 bench.py --no-cpu-baseline > gpurun_out/j1_bench_head.json 2> gpurun_out/j1_bench_head.err
for r in 64 16; do
  LIGHTLOAM_RING_SPLIT=1 python3 tools/bench_latency.py $r 60 > gpurun_out/j1_latency_split_$r.json 2>/dev/null
  LIGHTLOAM_RING_SPLIT=0 python3 tools/bench_latency.py $r 60 > gpurun_out/j1_latency_fused_$r.json 2>/dev/null
  LIGHTLOAM_RING_SPLIT=1 python3 tools/bench_latency.py $r 60 > gpurun_out/j1_latency_split_${r}b.json 2>/dev/null
  LIGHTLOAM_RING_SPLIT=0 python3 tools/bench_latency.py $r 60 > gpurun_out/j1_latency_fused_${r}b.json 2>/dev/null
done
cat gpurun_out/j1_latency_*.json
tail -c 1500 gpurun_out/j1_bench_head.json
