set -o pipefail
mkdir -p gpurun_out/r04e
O=gpurun_out/r04e
timeout -k 10 900 python -m pytest tests/test_gpu_ieee.py -q -x > $O/pytest_ieee.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ieee.log
tail -n 12 $O/pytest_ieee.log
timeout -k 10 900 python -m pytest tests/test_gpu_product_build.py -q -x > $O/pytest_product.log 2>&1; echo "pytest rc=$?" >> $O/pytest_product.log
tail -n 5 $O/pytest_product.log
run() {
for i in 1 2; do timeout -k 10 120 python tools/prof_frames.py cornell 1920 1080 64 8 1 0 1 32 2>&1 | grep -E "scene=|kernel ms" >> $O/$1; done
timeout -k 10 200 python tools/prof_frames.py sponza_like 1920 1080 32 8 3 0 1 32 2>&1 | grep -E "kernel ms" >> $O/$1
timeout -k 10 200 python tools/prof_frames.py instanced1000 1920 1080 32 8 3 0 1 32 2>&1 | grep -E "kernel ms" >> $O/$1
timeout -k 10 200 python tools/prof_frames.py glass_blob 3840 2160 32 16 3 0 1 32 2>&1 | grep -E "kernel ms" >> $O/$1
}
run time_fast.log
echo FAST; cat $O/time_fast.log
