#!/bin/bash
# PMC pass (SQ set) + wave stats for one scene: bash tools/pmc_scene.sh <scene>
R=$GRAFT_REPO_ROOT; S=${1:-sponza_like}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_s
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc_s -- python3 $R/tools/prof_frames.py $S 1920 1080 8 8 1 0 1 8 > $R/gpurun_out/pmc_s.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s | grep -v prepare
rm -rf $R/gpurun_out/pmc_s2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_s2 -- python3 $R/tools/prof_frames.py $S 1920 1080 8 8 1 0 1 8 > $R/gpurun_out/pmc_s2.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s2 | grep -v prepare
cd $R && python tools/prof_frames.py $S 1920 1080 8 8 1 1 1 8 | tail -3
