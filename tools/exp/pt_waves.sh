set -e
mkdir -p gpurun_out/r03p
for w in 6 5 4; do
  python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True, extra_flags=['-DRT_PT_GLOBAL_WAVES=$w'])" > /dev/null 2>&1
  echo "== RT_PT_GLOBAL_WAVES=$w"
  MODES=device timeout -k 10 200 python tools/animate_bench.py 512 256 40 2>&1 | grep "triangles skinned"
  timeout -k 10 200 python tools/prof_frames.py sponza_like 1920 1080 8 8 1 2>&1 | tail -2
done
