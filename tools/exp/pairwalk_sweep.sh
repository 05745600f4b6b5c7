#!/bin/bash
# round 3: knobs of the child-pair walk in the wavefront trace kernels (32-frame batches at 1080p, path-trace stage ms per batch)
run() { for s in sponza_like instanced1000; do timeout -k 10 120 python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 2>&1 | grep "kernel ms" | sed 's/.*kernel ms (sum over the run): //' | tr '\n' ' '; done; echo; }
build() { python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags="$1".split())
PY
}
for flags in "" "-DRT_WF_STEPS_PER_TRIP=2" "-DRT_WF_STEPS_PER_TRIP=4" "-DRT_WF_STEPS_PER_TRIP=6" \
             "-DRT_PW_STACK_K=7 -DRT_WF_WAVES=5" "-DRT_PW_STACK_K=4 -DRT_WF_WAVES=6" "-DRT_PW_STACK_K=4 -DRT_WF_WAVES=4" \
             "-DRT_PW_ENTER_BATCH=8" "-DRT_PW_ENTER_BATCH=32" "-DRT_WF_REFILL=16" "-DRT_WF_REFILL=32" $EXTRA_VARIANTS; do
  build "$flags"; echo "[$flags]"; run
done
build ""
