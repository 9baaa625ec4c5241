# SQ counters of one library (LIGHTLOAM_HIP_LIB or the tree's) at batch 2048: bash tools/sq_assoc.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-tree}
rm -rf $O/pmc_a_$T $O/pmc_b_$T
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace -f csv -d $O/pmc_a_$T -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 2048 > $O/pmc_a_$T.log 2>&1
python3 tools/sq_summary.py $O/pmc_a_$T | grep -E "k_associate|k_build_grid"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -f csv -d $O/pmc_b_$T -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 2048 > $O/pmc_b_$T.log 2>&1
python3 tools/sq_summary.py $O/pmc_b_$T | grep -E "k_associate|k_build_grid"
