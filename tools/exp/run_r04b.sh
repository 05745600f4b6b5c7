set -o pipefail
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
timeout -k 10 900 python -m pytest tests/test_gpu_product_build.py -x -q > $O/pytest_product.log 2>&1; echo "pytest rc=$?" >> $O/pytest_product.log
tail -n 15 $O/pytest_product.log
for i in 1 2; do timeout -k 10 120 python tools/prof_frames.py cornell 1920 1080 64 8 1 0 1 32 >> $O/time_cornell.log 2>&1; done
timeout -k 10 200 python tools/prof_frames.py sponza_like 1920 1080 32 8 3 0 1 32 >> $O/time_big.log 2>&1
timeout -k 10 200 python tools/prof_frames.py instanced1000 1920 1080 32 8 3 0 1 32 >> $O/time_big.log 2>&1
timeout -k 10 200 python tools/prof_frames.py glass_blob 3840 2160 32 16 3 0 1 32 >> $O/time_big.log 2>&1
MI355RT_WALK=node timeout -k 10 200 python tools/prof_frames.py sponza_like 1920 1080 32 8 3 0 1 32 >> $O/time_big_node.log 2>&1
cat $O/time_cornell.log $O/time_big.log $O/time_big_node.log
