#!/bin/bash
# calibration for the trace kernels: dependent divergent gathers (tools/gather_peak.hip) beside the node / triangle-test
# counts of one 32-frame batch of the big scenes (DETAIL counters)
mkdir -p gpurun_out/r02m
timeout -k 10 300 tools/bin/gather_peak > gpurun_out/r02m/gather_peak.txt 2>&1
cat gpurun_out/r02m/gather_peak.txt
for s in sponza_like instanced1000; do timeout -k 10 200 python tools/prof_frames.py $s 1920 1080 32 8 3 1 1 32 2>&1 | tail -4; done
