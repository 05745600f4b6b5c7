#!/bin/bash
# round 3: knobs of the child-pair walk (LDS-DMA fetch) in the wavefront trace kernels (32-frame batches, kernel ms per batch)
# VARIANTS="flags|flags|..." [RUNS="scene w h depth;..."] bash tools/exp/pairwalk_sweep2.sh
RUNS=${RUNS:-"sponza_like 1920 1080 8;instanced1000 1920 1080 8"}
run() { echo "$RUNS" | tr ';' '\n' | while read s w h d; do timeout -k 10 200 python tools/prof_frames.py $s $w $h 32 $d 3 0 1 32 2>&1 | grep "kernel ms" | sed "s/.*kernel ms (sum over the run): /$s /" | tr '\n' ' '; done; echo; }
build() { python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags="$1".split())
PY
}
echo "$VARIANTS" | tr '|' '\n' | while read flags; do
  build "$flags"; echo "[$flags]"; run
done
build ""
