#!/bin/bash
# Mutation fuzzing of the two parsers that take untrusted bytes — mt_decode (PNG / JPEG) and ms_world_create_glb (glTF) —
# under AddressSanitizer + UBSan on the CPU build (GPU sanitizers are not available on this pool).
# usage: tools/fuzz/run.sh [iterations per seed]     (seeds are generated with PIL and tests/gltf_util.py)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
W=${TMPDIR:-/tmp}/mi355_fuzz
mkdir -p $W && cd $W
python3 - "$R" <<'PY'
import io, sys
R = sys.argv[1]
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
from PIL import Image
import webgpu_raytracer_amd as W, test_gltf
rng = np.random.default_rng(0)
def smooth(h, w, c):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([(127 + 100 * np.sin(x / (7.0 + k) + y / (11.0 - k)) + rng.integers(0, 20, (h, w))).clip(0, 255) for k in range(c)], -1).astype(np.uint8)
seeds = []
for mode, c in (("L", 1), ("RGB", 3), ("RGBA", 4), ("LA", 2)):
    im = Image.fromarray(smooth(33, 47, c)[..., 0] if c == 1 else smooth(33, 47, c), mode)
    for opt in (False, True):
        b = io.BytesIO(); im.save(b, "PNG", optimize=opt); seeds.append(b.getvalue())
im = Image.fromarray(smooth(40, 56, 3), "RGB")
b = io.BytesIO(); im.quantize(9).save(b, "PNG"); seeds.append(b.getvalue())
for sub in (0, 1, 2):
    for prog in (False, True):
        b = io.BytesIO(); im.save(b, "JPEG", quality=70, subsampling=sub, progressive=prog, restart_marker_blocks=2 if sub == 2 else 0); seeds.append(b.getvalue())
b = io.BytesIO(); Image.fromarray(smooth(40, 56, 1)[..., 0], "L").save(b, "JPEG"); seeds.append(b.getvalue())
for i, s in enumerate(seeds): open("img_%02d.bin" % i, "wb").write(s)
for i, s in enumerate([test_gltf.build_static(W)[0].glb(), test_gltf.build_skinned(W)[0].glb(), test_gltf.build_static(W)[0].gltf_json()]):
    open("glb_%02d.bin" % i, "wb").write(s)
PY
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I $R/include $R/tools/fuzz/fuzz_inputs.cpp \
    $R/webgpu-raytracer_amd/csrc/texture/image_decode.cpp $R/webgpu-raytracer_amd/csrc/scene/scene_compiler.cpp -pthread -o fuzz
./fuzz ${1:-1500}
