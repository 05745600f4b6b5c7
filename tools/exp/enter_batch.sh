#!/bin/bash
# deferred instance entry: batch threshold sweep (RT_ENTER_BATCH), big scenes, 32-frame batches
run() { for s in sponza_like instanced1000; do timeout -k 10 120 python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 2>&1 | tail -2 | head -1; done
        timeout -k 10 200 python tools/prof_frames.py glass_blob 3840 2160 8 16 3 0 1 8 2>&1 | tail -2 | head -1; }
echo "== RT_ENTER_BATCH=16 (default build)"; run
for k in 1 8 32 48; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_ENTER_BATCH=${k}u"])
PY
echo "== RT_ENTER_BATCH=$k"; run
done
