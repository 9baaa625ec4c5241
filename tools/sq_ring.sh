# SQ counters of k_ring_features for one library (LIGHTLOAM_HIP_LIB or the tree's) at batch 2048: bash tools/sq_ring.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-tree}
rm -rf $O/pmc_r_$T
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -f csv -d $O/pmc_r_$T -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 2048 > $O/pmc_r_$T.log 2>&1
python3 tools/sq_summary.py $O/pmc_r_$T | grep -E "k_ring"
