cd $GRAFT_REPO_ROOT
(python bench.py --steps 1200 --warmup 3 --no-cpu-baseline --batch 4096 > gpurun_out/clk_bench.json 2>/dev/null) &
BP=$!
sleep 14
for i in $(seq 1 60); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/ \+/ /g'
  echo
  sleep 0.3
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -c 300 gpurun_out/clk_bench.json
