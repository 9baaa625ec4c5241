# instruction-cache and issue counters of the extract kernels (batch 2048): bash tools/sq_icache.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -i -o -E "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CYCLES|DCACHE)[A-Z0-9_]*)" $O/counters_list.txt | sort -u | head -40
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 2048"
rm -rf $O/pmc_ic1 $O/pmc_ic2
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -f csv -d $O/pmc_ic1 -o sq -- $B > $O/pmc_ic1.log 2>&1
python3 tools/sq_summary.py $O/pmc_ic1 | grep -E "k_ring|k_classify|k_associate"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU --kernel-trace -f csv -d $O/pmc_ic2 -o sq -- $B > $O/pmc_ic2.log 2>&1
python3 tools/sq_summary.py $O/pmc_ic2 | grep -E "k_ring|k_classify|k_associate"
tail -3 $O/pmc_ic1.log $O/pmc_ic2.log
rm -rf $O/pmc_ic1 $O/pmc_ic2
