#!/bin/bash
for cap in 0 100000; do
echo "== MI355RT_TREELET_MAX=$cap"
for s in sponza_like instanced1000; do MI355RT_TREELET_MAX=$cap python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 2>&1 | tail -2 | head -1; done
MI355RT_TREELET_MAX=$cap python tools/prof_frames.py glass_blob 3840 2160 8 16 3 0 1 8 2>&1 | tail -2 | head -1
done
