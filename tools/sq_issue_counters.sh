cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/pmc_sq3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace -f csv -d $O/pmc_sq3 -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 4096 > $O/pmc_sq3.log 2>&1
python3 tools/sq_summary.py $O/pmc_sq3
rm -rf $O/pmc_sq4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -f csv -d $O/pmc_sq4 -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 4096 > $O/pmc_sq4.log 2>&1
python3 tools/sq_summary.py $O/pmc_sq4
