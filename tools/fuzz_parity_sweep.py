#!/usr/bin/env python3
"""Extended GPU / oracle parity sweep over random bridge-layout scenes (tests/random_scene.py): every kernel form,
batched dispatches, textures, thin lens.  usage: fuzz_parity_sweep.py [first_seed] [count]
With MI355RT_NO_LDS_STAGING=1 in the environment the same scenes go through the global-memory code paths (mixed-mode walk,
deferred instance entry) that scenes of this size otherwise never reach.  PRODUCT=1: the product build of the kernels (detailed
counters off — separate template instances, the ones bench.py times): images and the three ray counters are compared."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import webgpu_raytracer_amd as W  # noqa: E402
import oracle_lib  # noqa: E402
import parity_util as pu  # noqa: E402
import random_scene  # noqa: E402

PRODUCT = os.environ.get("PRODUCT", "0") == "1"
RAYS = ("primary_rays", "extension_rays", "shadow_rays")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    kw = dict(n_geoms=int(rng.integers(1, 7)), tris_per_geom=int(rng.integers(4, 300)), n_instances=int(rng.integers(1, 120)),
              with_textures=bool(rng.integers(0, 2)), lens=float(rng.choice([0.0, 0.0, 0.05])))
    b = random_scene.make(seed, **kw)
    w, h = int(rng.integers(17, 120)), int(rng.integers(9, 80))
    depth, spp = int(rng.integers(1, 10)), int(rng.choice([1, 1, 2, 3]))
    frames = tuple(range(1, int(rng.integers(2, 7))))
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, spp, frames, present=True)
    for variant in (0, 1, 2, 3):
        for batch in (1, 4):
            if variant == 0 and batch > 1:
                continue
            gpu = W.WebGPURenderer(0)
            gpu.setKernelVariant(variant)
            gpu.buildPipeline(depth, spp)
            W.upload_scene(gpu, b, w, h)
            gpu.setCounting(not PRODUCT)
            gpu.resetCounters()
            if batch == 1:
                for f in frames:
                    gpu.compute(f)
                    gpu.present()
            else:
                # batches change the present cadence, so compare the accumulation buffer and counters only
                for i in range(0, len(frames), batch):
                    gpu.computeBatch(frames[i:i + batch])
            gpu.sync()
            try:
                gc, cc = gpu.getCounters(), cpu.getCounters()
                if batch == 1:
                    pu.assert_parity(gpu, cpu, check_output=True, check_counters=not PRODUCT)
                else:
                    assert np.array_equal(gpu.readAccum().view(np.uint32), cpu.readAccum().view(np.uint32))
                    assert PRODUCT or gc == cc
                assert {k: gc[k] for k in RAYS} == {k: cc[k] for k in RAYS}
            except AssertionError as e:
                bad += 1
                print("MISMATCH seed=%d variant=%d batch=%d %s %dx%d depth=%d spp=%d frames=%d: %s" % (
                    seed, variant, batch, kw, w, h, depth, spp, len(frames), str(e)[:200]), flush=True)
            gpu.destroy()
    if (seed - first) % 5 == 4:
        print("seed %d done, mismatches so far: %d" % (seed, bad), flush=True)
print("sweep of %d random scenes x 7 configurations: %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
