#!/bin/bash
# rocprofv3 evidence of a round (run on the GPU box via gpurun; HEAD=<commit> in the environment names the code state,
# RTAG=<rNN> the file prefix under profiles/).  One counter set per run (never --pmc together with other trace domains).
# usage: RTAG=r04 HEAD=<commit> bash tools/make_profiles_round.sh [cornell|bench|big|glass|world|calib|scene <name>|all]
R=$GRAFT_REPO_ROOT
WHAT=${1:-all}
RTAG=${RTAG:-r04}
export RTAG
OUT=$R/gpurun_out/profiles_$RTAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc_pass() {  # name scene frames depth batch images counters...
  local name=$1 scene=$2 frames=$3 depth=$4 batch=$5 images=$6; shift 6
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$scene/$name -- python3 $R/tools/prof_workload.py $scene $frames $depth $batch $images > $OUT/pmc_${scene}_$name.log 2>&1 || { echo "pass failed: $scene $name"; tail -3 $OUT/pmc_${scene}_$name.log; }
}
if [ "$WHAT" = cornell ] || [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  # 1. kernel trace + stats of the bench command itself (mode `bench`: only this, after profiles/pmc_reference.json of the
  #    same kernel sources is in place, so that the line carries the counters)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-world-update --no-live-loop > $OUT/${RTAG}_bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
  grep -c '"metric"' $OUT/${RTAG}_bench_under_rocprof.json || tail -5 $OUT/bench_under_rocprof.err
  cp $OUT/bench_trace/*/*_kernel_stats.csv $OUT/${RTAG}_bench_kernel_stats.csv
  rm -rf $OUT/bench_trace
  if [ "$WHAT" = bench ]; then ls $OUT; exit 0; fi
  # 2. PMC passes, each in its own run: 2 images = 4 path-trace launches of 32 frames
  rm -rf $OUT/pmc_cornell
  pmc_pass sq1 cornell 64 8 32 2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
  pmc_pass sq2 cornell 64 8 32 2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT
  pmc_pass fetch cornell 64 8 32 2 FETCH_SIZE
  pmc_pass write cornell 64 8 32 2 WRITE_SIZE
  pmc_pass grbm cornell 64 8 32 2 GRBM_GUI_ACTIVE
  python3 $R/tools/pmc_collect.py $OUT/pmc_cornell $OUT/${RTAG}_cornell_pmc.json > /dev/null
fi
if [ "$WHAT" = world ] || [ "$WHAT" = all ]; then
  # the device-resident update(t) (rt_world_update) inside the animated live loop: per-kernel durations of 43 displayed frames
  MODES=device timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_world -- python3 $R/tools/animate_bench.py 512 256 40 > $OUT/${RTAG}_world_update.log 2>&1
  cp $OUT/trace_world/*/*_kernel_stats.csv $OUT/${RTAG}_world_update_kernel_stats.csv
  rm -rf $OUT/trace_world
  timeout -k 10 300 python3 $R/tools/animate_bench.py 512 256 40 2>&1 | grep "triangles" > $OUT/${RTAG}_animate_bench.txt
  SCENE=hall timeout -k 10 300 python3 $R/tools/animate_bench.py 512 256 40 2>&1 | grep "triangles" >> $OUT/${RTAG}_animate_bench.txt
fi
if [ "$WHAT" = scene ]; then SCENES="$2"; fi
if [ "$WHAT" = big ] || [ "$WHAT" = all ] || [ "$WHAT" = scene ]; then
  # per-kernel durations and counters: the two trace kernels of a depth one after the other (by default they share the GPU
  # on two streams and their durations overlap)
  export MI355RT_WF_OVERLAP=0
  for scene in ${SCENES:-sponza_like instanced1000}; do
    rm -rf $OUT/pmc_$scene
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$scene -- python3 $R/tools/prof_workload.py $scene 64 8 32 1 > $OUT/trace_$scene.log 2>&1
    cp $OUT/trace_$scene/*/*_kernel_stats.csv $OUT/${RTAG}_${scene}_kernel_stats.csv
    rm -rf $OUT/trace_$scene
    pmc_pass sq1 $scene 32 8 32 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
    pmc_pass tcp1 $scene 32 8 32 1 TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
    pmc_pass tcp2 $scene 32 8 32 1 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
    pmc_pass tcp3 $scene 32 8 32 1 TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
    pmc_pass tcc1 $scene 32 8 32 1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
    pmc_pass fetch $scene 32 8 32 1 FETCH_SIZE
    pmc_pass write $scene 32 8 32 1 WRITE_SIZE
    pmc_pass grbm $scene 32 8 32 1 GRBM_GUI_ACTIVE
    python3 $R/tools/pmc_collect.py $OUT/pmc_$scene $OUT/${RTAG}_${scene}_pmc.json > /dev/null
  done
fi
if [ "$WHAT" = glass ] || [ "$WHAT" = all ]; then
  # config 5 (4K, depth 16): kernel trace of one 32-frame batch, PMC passes on one 8-frame batch
  export MI355RT_WF_OVERLAP=0
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_glass -- python3 $R/tools/prof_workload.py glass_blob 32 16 32 1 3840 2160 > $OUT/trace_glass_blob.log 2>&1
  cp $OUT/trace_glass/*/*_kernel_stats.csv $OUT/${RTAG}_glass_blob_kernel_stats.csv
  rm -rf $OUT/trace_glass $OUT/pmc_glass_blob
  pmc4k() { local name=$1; shift; timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_glass_blob/$name -- python3 $R/tools/prof_workload.py glass_blob 8 16 8 1 3840 2160 > $OUT/pmc_glass_blob_$name.log 2>&1 || { echo "pass failed: glass_blob $name"; tail -3 $OUT/pmc_glass_blob_$name.log; }; }
  pmc4k sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
  pmc4k tcp1 TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
  pmc4k tcp2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
  pmc4k tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  pmc4k fetch FETCH_SIZE
  pmc4k write WRITE_SIZE
  pmc4k grbm GRBM_GUI_ACTIVE
  python3 $R/tools/pmc_collect.py $OUT/pmc_glass_blob $OUT/${RTAG}_glass_blob_pmc.json > /dev/null
  unset MI355RT_WF_OVERLAP
fi
if [ "$WHAT" = scene ]; then ls $OUT; exit 0; fi
if [ "$WHAT" = world ]; then ls $OUT; exit 0; fi
if [ "$WHAT" != calib ] && [ "$WHAT" != all ]; then
  python3 $R/tools/pmc_reference.py $OUT ${HEAD:-unknown} $OUT/pmc_reference.json > /dev/null
  ls $OUT
  exit 0
fi
mkdir -p $R/tools/bin
for t in valu_peak gather_peak issue_peak; do   # calibration binaries (git-ignored): build on the box when they did not travel
  [ -x $R/tools/bin/$t ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/bin/$t $R/tools/$t.hip 2>/dev/null
done
timeout -k 10 120 $R/tools/bin/valu_peak > $OUT/${RTAG}_valu_peak.txt 2>&1
timeout -k 10 300 $R/tools/bin/gather_peak > $OUT/${RTAG}_gather_peak.txt 2>&1
timeout -k 10 120 $R/tools/bin/issue_peak > $OUT/${RTAG}_issue_peak.txt 2>&1
timeout -k 10 200 python3 $R/tools/clock_check.py $OUT/${RTAG}_clock_check.json > $OUT/clock_check.log 2>&1
python3 $R/tools/pmc_reference.py $OUT ${HEAD:-unknown} $OUT/pmc_reference.json > /dev/null
ls $OUT
