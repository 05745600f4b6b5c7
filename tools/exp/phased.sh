#!/bin/bash
# phased mixed-mode trip: parity first, then the big scenes at several workgroup shapes
mkdir -p gpurun_out/r02m
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wavefront or forms or band or traversal" > gpurun_out/r02m/phased_parity.log 2>&1 || { tail -30 gpurun_out/r02m/phased_parity.log; exit 1; }
tail -3 gpurun_out/r02m/phased_parity.log
run() { for s in sponza_like instanced1000; do timeout -k 10 120 python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 2>&1 | tail -2 | head -1; done; }
echo "== default (256 x 6)"; run
echo "== 256 x 4"; MI355RT_WF_BLOCKS_PER_CU=4 run
echo "== 512 x 3"; MI355RT_WF_BLOCK=512 MI355RT_WF_BLOCKS_PER_CU=3 run
echo "== 512 x 2"; MI355RT_WF_BLOCK=512 MI355RT_WF_BLOCKS_PER_CU=2 run
echo "== 1024 x 1"; MI355RT_WF_BLOCK=1024 MI355RT_WF_BLOCKS_PER_CU=1 run
echo "== 256 x 6, no treelet"; MI355RT_TREELET_MAX=0 run
