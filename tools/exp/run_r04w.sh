O=gpurun_out/r04w; mkdir -p $O
for b in 4 5 6 7 8; do
  for run in "glass_blob 3840 2160 32 16" "instanced1000 1920 1080 32 8"; do
    set -- $run
    echo -n "blocks_per_cu=$b $1: " >> $O/blocks.txt
    MI355RT_WF_BLOCKS_PER_CU=$b timeout -k 10 200 python tools/prof_frames.py $1 $2 $3 $4 $5 3 0 1 32 2>&1 | grep "kernel ms" >> $O/blocks.txt
  done
done
cat $O/blocks.txt
