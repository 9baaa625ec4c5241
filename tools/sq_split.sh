# SQ counters of the split ring kernels (k_ring_pick*, k_ring_features) at batch 2048, two passes: bash tools/sq_split.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-tree}
rm -rf $O/pmc_s1_$T $O/pmc_s2_$T
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY --kernel-trace -f csv -d $O/pmc_s1_$T -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 2048 > $O/pmc_s1_$T.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace -f csv -d $O/pmc_s2_$T -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 2048 > $O/pmc_s2_$T.log 2>&1
python3 tools/sq_summary.py $O/pmc_s1_$T | grep -E "k_ring"
python3 tools/sq_summary.py $O/pmc_s2_$T | grep -E "k_ring"
