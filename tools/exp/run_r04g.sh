set -o pipefail
mkdir -p gpurun_out/r04g
O=gpurun_out/r04g
timeout -k 10 900 python -m pytest tests/test_gpu_product_build.py tests/test_gpu_parity.py tests/test_fuzz_parity.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
for i in 1 2; do timeout -k 10 120 python tools/prof_frames.py cornell 1920 1080 64 8 1 0 1 32 2>&1 | grep -E "scene=|kernel ms" >> $O/time.log; done
timeout -k 10 120 python tools/prof_frames.py cornell 1920 1080 16 8 1 0 1 1 2>&1 | grep -E "scene=|kernel ms" >> $O/time.log
cat $O/time.log
timeout -k 10 400 python tools/lane_stats.py cornell 2>&1 | grep -v "amdgpu.ids\|warning\|\^\|unsigned long long i\|In file" > $O/lane_stats_cornell.txt; cat $O/lane_stats_cornell.txt
