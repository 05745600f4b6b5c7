#!/bin/bash
# Collect the rocprofv3 evidence committed under profiles/ (run on the GPU box via gpurun).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_r01
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the bench command itself
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r01_bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
grep -c '"metric"' $OUT/r01_bench_under_rocprof.json || tail -5 $OUT/bench_under_rocprof.err
cp $OUT/bench_trace/*/*_kernel_stats.csv $OUT/r01_bench_kernel_stats.csv
# 1b. the wavefront form on the largest 1080p config (sponza-like, 263k triangles), 16 frames in batches of 8
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wf_trace -- python3 $R/tools/prof_frames.py sponza_like 1920 1080 16 8 3 0 1 8 > $OUT/wf_trace.log 2>&1
cp $OUT/wf_trace/*/*_kernel_stats.csv $OUT/r01_sponza_wavefront_kernel_stats.csv
# 2. PMC passes on the same workload: 64 frames as two batched dispatches of 32 (counters in their own runs)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-60)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 $R/tools/prof_frames.py cornell 1920 1080 64 8 1 0 1 32 > $OUT/pmc_$name.log 2>&1 || echo "pass failed: $set"
  python3 $R/tools/pmc_summary.py $OUT/pmc_$name >> $OUT/r01_pmc_summary.txt
done
# 3. the post pass too (FETCH/WRITE calibration on a streaming kernel)
cat > /tmp/post_frames.py <<PY
import sys; sys.path.insert(0, "$R")
import webgpu_raytracer_amd as W
b = W.WorldBridge(); b.loadScene("cornell")
r = W.WebGPURenderer(0); r.buildPipeline(8, 1); W.upload_scene(r, b, 1920, 1080)
for f in range(1, 21):
    r.compute(f); r.present()
r.sync()
PY
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/post_$set -- python3 /tmp/post_frames.py > $OUT/post_$set.log 2>&1
  python3 $R/tools/pmc_summary.py $OUT/post_$set | grep postprocess >> $OUT/r01_pmc_summary.txt
done
cat $OUT/r01_pmc_summary.txt | grep -v prepare
cut -c1-300 $OUT/r01_bench_under_rocprof.json
rm -rf $OUT/bench_trace $OUT/wf_trace $OUT/pmc_* $OUT/post_*
