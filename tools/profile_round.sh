#!/bin/bash
# The rocprofv3 passes behind profiles/<tag>_*.  Run on the GPU box from the repo root:
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r03'
# then copy gpurun_out/<tag>_* and gpurun_out/pmc_traffic.json into profiles/.
# Counter passes are separate runs with --kernel-trace only (TCC has 4 PMC slots; FETCH_SIZE takes 3, WRITE_SIZE 2).
# pmc_traffic.json is stamped with the library's source digest, ring count, workload and batch (bench.py uses it only on a match).
TAG=${1:-r01}
ROOT=$(pwd)
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --calibrate"
rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o $TAG -- $B > $O/prof_$TAG.log 2>&1
python3 tools/rocpd_summary.py $O/prof_$TAG/${TAG}_results.db > $O/${TAG}_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch -o fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write -o write -- $B > $O/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write --batch ${BATCH:-16384} --rings 64 --workload synthetic > $O/pmc_traffic.json
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
    --kernel-trace -f csv -d $O/pmc_sq -o sq -- $B > $O/pmc_sq.log 2>&1
python3 tools/sq_summary.py $O/pmc_sq > $O/${TAG}_sq_counters.txt
python3 tools/sq_issue.py $O/pmc_sq --batch ${BATCH:-16384} --rings 64 --workload synthetic > $O/sq_issue.json
cp $O/pmc_traffic.json profiles/pmc_traffic.json          # bench.py reads roofline.traffic from here (when the stamp matches)
cp $O/sq_issue.json profiles/sq_issue.json                # ... and roofline.issue from here
# the same three counter passes for BASELINE config 3's stand-in (bench.py --workload hdl64: 8192 scans per step, ring capacity 4608)
BH="python3 bench.py --workload hdl64 --steps 5 --warmup 1 --no-cpu-baseline --calibrate"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch_h -o fetch -- $BH > $O/pmc_fetch_h.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write_h -o write -- $BH > $O/pmc_write_h.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch_h $O/pmc_write_h --batch 8192 --rings 64 --workload hdl64 > $O/pmc_traffic_hdl64.json
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
    --kernel-trace -f csv -d $O/pmc_sq_h -o sq -- $BH > $O/pmc_sq_h.log 2>&1
python3 tools/sq_summary.py $O/pmc_sq_h > $O/${TAG}_sq_counters_hdl64.txt
python3 tools/sq_issue.py $O/pmc_sq_h --batch 8192 --rings 64 --workload hdl64 > $O/sq_issue_hdl64.json
cp $O/pmc_traffic_hdl64.json profiles/pmc_traffic_hdl64.json
cp $O/sq_issue_hdl64.json profiles/sq_issue_hdl64.json
# ... and for BASELINE config 5's shape (128 rings, 4096 scans per step): its dominant kernel is k_associate
BS="python3 bench.py --rings 128 --steps 5 --warmup 1 --no-cpu-baseline --calibrate"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch_s -o fetch -- $BS > $O/pmc_fetch_s.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write_s -o write -- $BS > $O/pmc_write_s.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch_s $O/pmc_write_s --batch 4096 --rings 128 --workload synthetic > $O/pmc_traffic_s128.json
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
    --kernel-trace -f csv -d $O/pmc_sq_s -o sq -- $BS > $O/pmc_sq_s.log 2>&1
python3 tools/sq_summary.py $O/pmc_sq_s > $O/${TAG}_sq_counters_s128.txt
python3 tools/sq_issue.py $O/pmc_sq_s --batch 4096 --rings 128 --workload synthetic > $O/sq_issue_s128.json
cp $O/pmc_traffic_s128.json profiles/pmc_traffic_s128.json
cp $O/sq_issue_s128.json profiles/sq_issue_s128.json
# ... and for BASELINE config 1's shape (16 rings, 32768 scans per step), so that no committed bench line says "bound: unknown"
BV="python3 bench.py --rings 16 --steps 5 --warmup 1 --no-cpu-baseline --calibrate"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch_v -o fetch -- $BV > $O/pmc_fetch_v.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write_v -o write -- $BV > $O/pmc_write_v.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch_v $O/pmc_write_v --batch 32768 --rings 16 --workload synthetic > $O/pmc_traffic_s16.json
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
    --kernel-trace -f csv -d $O/pmc_sq_v -o sq -- $BV > $O/pmc_sq_v.log 2>&1
python3 tools/sq_summary.py $O/pmc_sq_v > $O/${TAG}_sq_counters_s16.txt
python3 tools/sq_issue.py $O/pmc_sq_v --batch 32768 --rings 16 --workload synthetic > $O/sq_issue_s16.json
cp $O/pmc_traffic_s16.json profiles/pmc_traffic_s16.json
cp $O/sq_issue_s16.json profiles/sq_issue_s16.json
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
# the self-launching N-rank path on the one GPU of this pool (two ranks on device 0, gloo for the timing reduction: RCCL cannot share a device)
python3 bench.py --gpus 2 --backend gloo --share-gpu --batch 2048 --no-cpu-baseline > $O/${TAG}_bench_gpus2_gloo_share_gpu.json 2>/dev/null
# 9 distinct scans (the default of rounds 1-3) beside the 65 of the headline: what slot-to-slot diversity costs
python3 bench.py --distinct 8 --no-cpu-baseline > $O/${TAG}_bench_distinct8.json 2>/dev/null
# BASELINE config 3's stand-in (HDL-64E true laser table, ring capacity 4608): bench line + its kernel stats
python3 bench.py --workload hdl64 --no-cpu-baseline > $O/${TAG}_bench_hdl64.json 2> $O/${TAG}_bench_hdl64.err
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_hdl64 -o ${TAG}h -- python3 bench.py --workload hdl64 --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_${TAG}_hdl64.log 2>&1
python3 tools/rocpd_summary.py $O/prof_${TAG}_hdl64/${TAG}h_results.db > $O/${TAG}_kernel_stats_hdl64.txt
# the other shapes and BASELINE config 5 (stream of scans over PCIe, 64 and 128 rings)
python3 bench.py --rings 128 --no-cpu-baseline > $O/${TAG}_bench_s128.json 2>/dev/null
python3 bench.py --rings 16 --no-cpu-baseline > $O/${TAG}_bench_s16.json 2>/dev/null
# (12-byte points = ll_params.input_stride_floats 3: the 4th float of a .bin record never crosses PCIe; the 16-byte contract beside it)
python3 bench.py --stream-input --input-stride 3 --no-cpu-baseline > $O/${TAG}_stream_input.json 2>/dev/null
python3 bench.py --stream-input --input-stride 3 --rings 128 --no-cpu-baseline > $O/${TAG}_stream_input_s128.json 2>/dev/null
python3 bench.py --stream-input --no-cpu-baseline > $O/${TAG}_stream_input_16_byte_points.json 2>/dev/null
python3 bench.py --stream-input --rings 128 --no-cpu-baseline > $O/${TAG}_stream_input_s128_16_byte_points.json 2>/dev/null
# the full soaks behind tests/test_gpu_soak.py
python3 tools/soak_extract.py 8 > $O/${TAG}_soak_extract.log 2>&1
python3 tools/soak_extract_s64.py 96 > $O/${TAG}_soak_extract_s64.log 2>&1
python3 tools/soak_hot_path.py 256 2>&1 | grep -v amdgpu.ids > $O/${TAG}_soak_hot_path.log
python3 tools/soak_frames.py 120 60 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|org_s" > $O/${TAG}_soak_frames_short.log
# keep what is merged back small: the raw counter tables and traces stay on the box
rm -rf $O/prof_$TAG $O/prof_${TAG}_hdl64 $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_fetch_h $O/pmc_write_h $O/pmc_sq_h $O/pmc_fetch_s $O/pmc_write_s $O/pmc_sq_s $O/pmc_fetch_v $O/pmc_write_v $O/pmc_sq_v $O/liblightloam_hip_phase.so
tail -n 3 $O/${TAG}_kernel_stats.txt; tail -c 600 $O/${TAG}_bench.json
