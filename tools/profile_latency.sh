# per-kernel durations of the node-style loop (tools/latency_breakdown.py) and of a mapping drive (tools/mapping_breakdown.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_lat gpurun_out/prof_mapb
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lat -o lat -- python3 tools/latency_breakdown.py > gpurun_out/prof_lat.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_lat/lat_results.db | head -32
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mapb -o mapb -- python3 tools/mapping_breakdown.py > gpurun_out/prof_mapb.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_mapb/mapb_results.db | head -60
