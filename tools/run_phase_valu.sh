# instruction counts of k_ring_features per phase: the early-return builds of tools/phase_stop_time.py (built on the CPU
# box into _phase/) under rocprofv3 --pmc; difference consecutive rows.  Run on the GPU box: bash tools/run_phase_valu.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LL_PHASE_DIR=_phase
for st in 0 1 2 3 4 12 13 5 7 9; do
  rm -rf $R/gpurun_out/pv$st
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace -f csv -d $R/gpurun_out/pv$st -o pv -- python3 $R/tools/phase_valu.py $st > $R/gpurun_out/pv$st.log 2>&1
  echo "stop $st: $(python3 $R/tools/sq_summary.py $R/gpurun_out/pv$st | grep k_ring)"
done
