# large-node threshold / chunk size of the BLAS builder against update(t) on the device (animated 262 k-triangle glTF)
set -e
# the product build is restored on ANY exit (a variant library left behind would be taken for the product one)
trap 'python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True)" > /dev/null 2>&1' EXIT
for v in "4096u 2048u" "8192u 2048u" "16384u 2048u" "8192u 4096u" "4096u 1024u"; do
  set -- $v
  python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True, extra_flags=['-DRT_BLAS_BIG=$1', '-DRT_BLAS_CHUNK=$2'])" > /dev/null 2>&1
  echo "== kBig $1 kChunk $2"
  MODES=device timeout -k 10 200 python tools/animate_bench.py 512 256 40 2>&1 | grep "triangles skinned" | sed 's/.*update(t)/update(t)/'
done
