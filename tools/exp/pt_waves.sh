# waves per SIMD of the global-memory persistent kernel on single-frame dispatches (tools/SWEEPS.md).  Builds VARIANT
# libraries in place: the product build is restored on ANY exit (a variant library left behind would be newer than every
# source and taken for the product one by _build.build_rt and by bench.py's pmc_stale check).
set -e
trap 'python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True)" > /dev/null 2>&1' EXIT
mkdir -p gpurun_out/r03p
for w in 6 5 4; do
  python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True, extra_flags=['-DRT_PT_GLOBAL_WAVES=$w'])" > /dev/null 2>&1
  echo "== RT_PT_GLOBAL_WAVES=$w"
  MODES=device timeout -k 10 200 python tools/animate_bench.py 512 256 40 2>&1 | grep "triangles skinned"
  timeout -k 10 200 python tools/prof_frames.py sponza_like 1920 1080 8 8 1 2>&1 | tail -2
done
