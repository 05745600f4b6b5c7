# The loop of a kernel change: product-build + counting parity tests, then ms per image / per 32 frames of the headline and the three large configs.
# usage (GPU box): bash tools/exp/check_and_time.sh <tag>   -> gpurun_out/<tag>/{pytest.log,time.log}
set -o pipefail
O=gpurun_out/${1:-r04q}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_product_build.py tests/test_gpu_parity.py tests/test_fuzz_parity.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
for i in 1 2; do timeout -k 10 120 python tools/prof_frames.py cornell 1920 1080 64 8 1 0 1 32 2>&1 | grep -E "scene=|kernel ms" >> $O/time.log; done
timeout -k 10 200 python tools/prof_frames.py sponza_like 1920 1080 32 8 3 0 1 32 2>&1 | grep -E "scene=|kernel ms" >> $O/time.log
timeout -k 10 200 python tools/prof_frames.py instanced1000 1920 1080 32 8 3 0 1 32 2>&1 | grep -E "scene=|kernel ms" >> $O/time.log
timeout -k 10 300 python tools/prof_frames.py glass_blob 3840 2160 32 16 3 0 1 32 2>&1 | grep -E "scene=|kernel ms" >> $O/time.log
cat $O/time.log
