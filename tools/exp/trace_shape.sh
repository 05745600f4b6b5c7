# Resident workgroups per CU of the node-walk trace kernels x where the instance-space ray lives.
# Part 1: what fits (MI355RT_DEBUG_SHAPE); part 2: ms per 32 frames at 5..8 workgroups per CU, both forms.  Run on the GPU box.
O=gpurun_out/trace_shape; mkdir -p $O
for flags in "-DRT_POST_AT_ENTRY_MIXED=1" "-DRT_POST_AT_ENTRY_MIXED=0"; do
  python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True, extra_flags='$flags'.split())" > /dev/null 2>&1
  echo "[$flags]" >> $O/shape.txt
  MI355RT_DEBUG_SHAPE=1 timeout -k 10 200 python tools/prof_frames.py glass_blob 1920 1080 4 16 3 0 1 4 2>&1 | grep "mi355rt\]" | sort | uniq -c >> $O/shape.txt
  MI355RT_DEBUG_SHAPE=1 timeout -k 10 200 python tools/prof_frames.py instanced1000 1920 1080 4 8 3 0 1 4 2>&1 | grep "mi355rt\]" | sort | uniq -c >> $O/shape.txt
done
python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True)" > /dev/null 2>&1
cat $O/shape.txt

O=gpurun_out/trace_shape; mkdir -p $O
for flags in "-DRT_POST_AT_ENTRY_MIXED=1" "-DRT_POST_AT_ENTRY_MIXED=0"; do
  python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True, extra_flags='$flags'.split())" > /dev/null 2>&1
  for b in 5 6 7 8; do
    for run in "glass_blob 3840 2160 32 16" "instanced1000 1920 1080 32 8"; do
      set -- $run
      echo -n "[$flags] blocks_per_cu=$b $1: " >> $O/blocks.txt
      MI355RT_WF_BLOCKS_PER_CU=$b timeout -k 10 200 python tools/prof_frames.py $1 $2 $3 $4 $5 3 0 1 32 2>&1 | grep "kernel ms" >> $O/blocks.txt
    done
  done
done
python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True)" > /dev/null 2>&1
cat $O/blocks.txt
