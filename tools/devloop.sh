#!/bin/bash
# GPU dev loop: parity tests, timing of both kernel forms, one PMC pass (run via gpurun).
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
python tools/prof_frames.py cornell 1920 1080 16 8 1 0 2>&1 | tail -2
python tools/prof_frames.py instanced1000 1920 1080 8 8 1 0 2>&1 | tail -2
python tools/prof_frames.py sponza_like 1920 1080 8 8 1 0 2>&1 | tail -2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/prof_frames.py cornell 1920 1080 4 8 1 0 > $R/gpurun_out/pmc1.log 2>&1; echo pmc rc=$?
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc1
