set -o pipefail
mkdir -p gpurun_out/r04n
O=gpurun_out/r04n
timeout -k 10 900 python -m pytest tests/test_gpu_world_update.py tests/test_gltf.py tests/test_node_host.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log
MODES=device timeout -k 10 200 python tools/animate_bench.py 512 256 30 2>&1 | grep "triangles" > $O/animate.txt
SCENE=hall MODES=device timeout -k 10 200 python tools/animate_bench.py 512 256 30 2>&1 | grep "triangles" >> $O/animate.txt
cat $O/animate.txt
timeout -k 10 200 python tools/blas_build_time.py 2>&1 | grep -v amdgpu.ids | tail -8 > $O/blas_build_time.txt; cat $O/blas_build_time.txt
