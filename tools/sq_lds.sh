cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 2048"
rm -rf $O/pmc_l1 $O/pmc_l2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace -f csv -d $O/pmc_l1 -o sq -- $B > $O/pmc_l1.log 2>&1
python3 tools/sq_summary.py $O/pmc_l1 | grep -E "k_ring|k_organize|k_associate"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --kernel-trace -f csv -d $O/pmc_l2 -o sq -- $B > $O/pmc_l2.log 2>&1
python3 tools/sq_summary.py $O/pmc_l2 | grep -E "k_ring|k_organize|k_associate"
grep -i -E "error|invalid" $O/pmc_l1.log $O/pmc_l2.log | head -5
rm -rf $O/pmc_l1 $O/pmc_l2
