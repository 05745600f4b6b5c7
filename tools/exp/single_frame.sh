# Single-frame dispatches of the large scenes: persistent global-memory form at 6 / 5 / 4 waves per SIMD, wavefront form at n = 1.
O=gpurun_out/single_frame; mkdir -p $O
for w in 6 5 4; do
  python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True, extra_flags=['-DRT_PT_GLOBAL_WAVES=$w'])" > /dev/null 2>&1
  for sc in sponza_like instanced1000; do
    echo -n "waves=$w " >> $O/single_frame.log
    timeout -k 10 200 python tools/prof_frames.py $sc 1920 1080 8 8 1 0 1 1 2>&1 | grep -E "scene=" >> $O/single_frame.log
  done
done
python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True)" > /dev/null 2>&1
for sc in sponza_like instanced1000; do
  echo -n "wavefront n=1 " >> $O/single_frame.log
  timeout -k 10 200 python tools/prof_frames.py $sc 1920 1080 8 8 2 0 1 1 2>&1 | grep -E "scene=" >> $O/single_frame.log
done
cat $O/single_frame.log
