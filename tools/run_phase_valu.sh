cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LL_PHASE_DIR=_phase
for st in 2 3 4 5 9; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -f csv -d $R/gpurun_out/pv$st -o pv -- python3 $R/tools/phase_valu.py $st > $R/gpurun_out/pv$st.log 2>&1
  echo "stop $st: $(python3 $R/tools/sq_summary.py $R/gpurun_out/pv$st | grep k_ring)"
done
