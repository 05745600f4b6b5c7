O=gpurun_out/r05b; mkdir -p $O
for flags in "-DRT_POST_AT_ENTRY_MIXED=1" "-DRT_POST_AT_ENTRY_MIXED=0"; do
  python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True, extra_flags='$flags'.split())" > /dev/null 2>&1
  echo "[$flags]" >> $O/shape.txt
  MI355RT_DEBUG_SHAPE=1 timeout -k 10 200 python tools/prof_frames.py glass_blob 1920 1080 4 16 3 0 1 4 2>&1 | grep "mi355rt\]" | sort | uniq -c >> $O/shape.txt
  MI355RT_DEBUG_SHAPE=1 timeout -k 10 200 python tools/prof_frames.py instanced1000 1920 1080 4 8 3 0 1 4 2>&1 | grep "mi355rt\]" | sort | uniq -c >> $O/shape.txt
done
python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True)" > /dev/null 2>&1
cat $O/shape.txt
