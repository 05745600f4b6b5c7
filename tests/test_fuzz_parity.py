"""Randomised bridge-layout scenes (tests/random_scene.py): the oracle must survive them on the CPU, and on the GPU
the HIP path must agree with it bit for bit — duplicated and degenerate triangles, 7-triangle leaves, rotated /
non-uniformly scaled instances, all four materials incl. emissive non-lights, textures with repeat addressing,
thin lens."""
import numpy as np
import pytest

import parity_util as pu
import random_scene


def test_oracle_renders_random_scenes(W, oracle_lib):
    b = random_scene.make(1, n_geoms=2, tris_per_geom=20, n_instances=3)
    r = oracle_lib.OracleRenderer()
    pu.drive(r, W, b, 48, 32, 6, 1, (1, 2), present=True)
    acc = r.readAccum()
    assert (acc[..., 3] == 2).all()
    c = r.getCounters()
    assert c["primary_rays"] == 48 * 32 * 2 and c["tris_tested"] > 0
    # determinism
    r2 = oracle_lib.OracleRenderer(threads=1)
    pu.drive(r2, W, random_scene.make(1, n_geoms=2, tris_per_geom=20, n_instances=3), 48, 32, 6, 1, (1, 2), present=True)
    assert np.array_equal(acc.view(np.uint32), r2.readAccum().view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", [
    (11, dict()),
    (12, dict(n_geoms=5, tris_per_geom=120, n_instances=24)),
    (13, dict(n_geoms=1, tris_per_geom=8, n_instances=1)),
    (14, dict(with_textures=True)),
    (15, dict(lens=0.08, n_instances=12)),
    (16, dict(n_geoms=8, tris_per_geom=400, n_instances=300)),     # does not fit LDS: global-memory kernel form
])
def test_random_scene_parity(W, oracle_lib, seed, kw):
    W._build.build_rt()
    b = random_scene.make(seed, **kw)
    # live loop with present(): accumulation, history and RGBA8 output
    gpu, cpu = W.WebGPURenderer(0), oracle_lib.OracleRenderer()
    for r in (gpu, cpu):
        pu.drive(r, W, b, 96, 64, 8, 2, (1, 2, 3), present=True)
    pu.assert_parity(gpu, cpu, check_output=True)
    gpu.destroy()
    # no present(): G-buffer planes and counters; then the same frames again with the one-pixel-per-lane kernel form
    gpu, cpu = W.WebGPURenderer(0), oracle_lib.OracleRenderer()
    for variant in (1, 0):
        gpu.setKernelVariant(variant)
        for r in (gpu, cpu):
            pu.drive(r, W, b, 80, 56, 5, 1, (1, 2), present=False)
        pu.assert_parity(gpu, cpu, check_output=False)
    gpu.destroy()
