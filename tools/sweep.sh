#!/bin/bash
# ONE sweep driver for every build-flag / environment experiment quoted in DESIGN.md (tools/SWEEPS.md lists them with the
# invocation that reproduces each).  For every variant: rebuild libmi355rt.so with the extra compiler flags, run the listed
# workloads (one batched dispatch of FRAMES frames each, tools/prof_frames.py) and print the per-kernel milliseconds.
# The product library is rebuilt at the end.  Run on the GPU box (gpurun).
#   VARIANTS="flags|flags|..."    compiler flags per variant, "" = the product build
#   RUNS="scene w h depth;..."    default: the three large configs at their BASELINE sizes
#   FRAMES=32 BATCH=32 KVARIANT=3 frames per run / per dispatch, kernel form (rt_set_kernel_variant)
#   ENVS="A=1 B=2"                environment for the runs (e.g. MI355RT_WALK=node)
RUNS=${RUNS:-"sponza_like 1920 1080 8;instanced1000 1920 1080 8;glass_blob 3840 2160 16"}
FRAMES=${FRAMES:-32}; BATCH=${BATCH:-32}; KVARIANT=${KVARIANT:-3}
run() { echo "$RUNS" | tr ';' '\n' | while read s w h d; do env $ENVS timeout -k 10 200 python tools/prof_frames.py $s $w $h $FRAMES $d $KVARIANT 0 1 $BATCH 2>&1 | grep "kernel ms" | sed "s/.*kernel ms (sum over the run): /$s /" | tr '\n' ' '; done; echo; }
# the product build is restored on ANY exit (a variant library left behind would be taken for the product one)
trap 'python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True)" > /dev/null 2>&1' EXIT
build() { python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags="$1".split())
PY
}
echo "$VARIANTS" | tr '|' '\n' | while read flags; do
  build "$flags"; echo "[$flags]"; run
done
build ""
