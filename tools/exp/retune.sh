#!/bin/bash
# re-sweep of the trace-kernel knobs on the final kernels: node steps per trip x refill threshold (32-frame batches)
run() { for s in sponza_like instanced1000; do timeout -k 10 120 python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 2>&1 | tail -3 | head -1 | sed 's/.*pathtrace \([0-9.]*\) ms.*/\1/' | tr '\n' ' '; done; echo; }
for st in 4 6 8 12; do for rf in 8 16 24 32 40; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_WF_STEPS_PER_TRIP=$st", "-DRT_WF_REFILL=$rf"])
PY
echo -n "steps $st refill $rf: "; run
done; done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
