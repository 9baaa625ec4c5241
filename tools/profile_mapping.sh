cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_map
rocprofv3 --kernel-trace --hip-trace --stats -d gpurun_out/prof_map -o map -- python3 tools/bench_mapping.py > gpurun_out/prof_map.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_map/map_results.db | head -40
tail -c 400 gpurun_out/prof_map.log
