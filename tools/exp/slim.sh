#!/bin/bash
mkdir -p gpurun_out/r02u
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_radiometric_kat.py -x -q -m gpu > gpurun_out/r02u/parity.log 2>&1 || { tail -30 gpurun_out/r02u/parity.log; exit 1; }
tail -2 gpurun_out/r02u/parity.log
for s in sponza_like instanced1000; do timeout -k 10 120 python tools/prof_frames.py $s 1920 1080 64 8 3 0 1 32 2>&1 | tail -3 | head -2; done
timeout -k 10 200 python tools/prof_frames.py glass_blob 3840 2160 32 16 3 0 1 32 2>&1 | tail -3 | head -2
