#!/bin/bash
# frames per batched dispatch: 32 vs 64 at 1080p (8 vs 16 at 4K)
for b in 32 64; do
echo "== batch $b"
timeout -k 10 120 python tools/prof_frames.py cornell 1920 1080 128 8 3 0 1 $b 2>&1 | tail -3 | head -2
for s in sponza_like instanced1000; do timeout -k 10 120 python tools/prof_frames.py $s 1920 1080 64 8 3 0 1 $b 2>&1 | tail -3 | head -2; done
done
for b in 8 16 32; do
echo "== 4K batch $b"
timeout -k 10 200 python tools/prof_frames.py glass_blob 3840 2160 32 16 3 0 1 $b 2>&1 | tail -3 | head -2
done
