"""The PRODUCT build of the kernels (detailed counters off — what bench.py times) against the oracle.

Most parity tests run the counting variant (`setCounting(True)`), because `nodes_visited` / `tris_tested` pin the walk
step by step.  The variants without those counters are separate template instances and, since round 4, take code the
counting ones do not: the division / square-root expansions are the short sequences of csrc/k_ieee.hip.h.  These tests put every scene, kernel form
and record location (LDS / global memory) through exactly that code: accumulation buffer, G-buffer, presented image,
history and the three ray counters must equal the oracle's.
"""
import numpy as np
import pytest

import parity_util as pu
import random_scene

pytestmark = pytest.mark.gpu

RAYS = ("primary_rays", "extension_rays", "shadow_rays")


def _run(W, r, b, w, h, depth, spp, frames, batch, variant, walk=None):
    r.setKernelVariant(variant)
    if walk is not None:
        r.setWalk(walk)
    r.buildPipeline(depth, spp)
    W.upload_scene(r, b, w, h)
    r.setCounting(False)
    r.resetCounters()
    for i in range(0, len(frames), batch):
        if batch == 1:
            r.compute(frames[i])
        else:
            r.computeBatch(list(frames[i:i + batch]))
    r.sync()


def _check(gpu, cpu):
    pu.assert_parity(gpu, cpu, check_output=False, check_counters=False)
    gc, cc = gpu.getCounters(), cpu.getCounters()
    assert {k: gc[k] for k in RAYS} == {k: cc[k] for k in RAYS}
    gpu.present()
    cpu.present()
    pu.assert_parity(gpu, cpu, check_output=True, check_counters=False)


SCENES = [
    ("cornell", 128, 96, 8, 1, tuple(range(1, 9)), 4),
    ("cornell", 64, 48, 6, 2, (0, 1, 2, 3), 1),
    ("viewer_diamond", 96, 54, 8, 1, (1, 2, 3, 4), 2),
    ("viewer_diamond_1k", 96, 54, 8, 1, (1, 2, 3, 4), 4),
    ("special", 80, 56, 8, 1, (1, 2, 3, 4), 4),            # 528 light triangles, fallback leaves of 5-7 triangles
    ("mixed", 64, 48, 10, 1, (1, 2, 3, 4), 4),             # thin lens, all materials
    ("mesh", 64, 48, 8, 1, (1, 2, 3, 4), 4),
    ("instanced1000", 96, 54, 8, 1, (1, 2, 3, 4), 4),      # 15 instance entries per ray
    ("sponza_like", 64, 36, 8, 1, (1, 2, 3, 4), 4),
    ("glass_blob", 48, 27, 16, 1, (1, 2, 3, 4), 4),
]


@pytest.mark.parametrize("scene,w,h,depth,spp,frames,batch", SCENES)
@pytest.mark.parametrize("variant,walk", [(1, None), (2, 0), (2, 1)])
def test_product_build_parity(W, oracle_lib, gpu_renderer, scene, w, h, depth, spp, frames, batch, variant, walk):
    if variant == 2 and spp != 1:
        pytest.skip("the wavefront form takes SPP = 1")
    b = pu.bridge_for(W, scene)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, spp, frames, present=False)
    _run(W, gpu_renderer, b, w, h, depth, spp, frames, batch, variant, walk)
    _check(gpu_renderer, cpu)


@pytest.mark.parametrize("scene,variant", [("cornell", 1), ("cornell", 2), ("mesh", 2), ("special", 1), ("special", 2), ("mixed", 2),
                                           ("viewer_diamond", 1)])
@pytest.mark.parametrize("rayreg", ["0", "1"])
def test_product_build_global_paths_on_small_scenes(W, oracle_lib, monkeypatch, scene, variant, rayreg):
    """MI355RT_NO_LDS_STAGING=1: the records of a small scene read through the L1 — the mixed-mode walk with deferred
    instance entry that the large configs take, on scenes the oracle renders whole in seconds; both forms of the trace kernels."""
    if variant == 1 and rayreg == "1":
        pytest.skip("the persistent kernel has one form")
    monkeypatch.setenv("MI355RT_NO_LDS_STAGING", "1")
    monkeypatch.setenv("MI355RT_WF_RAYREG", rayreg)
    b = pu.bridge_for(W, scene)
    w, h, depth, frames = 72, 48, 8, (1, 2, 3, 4)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, 1, frames, present=False)
    r = W.WebGPURenderer(0)
    _run(W, r, b, w, h, depth, 1, frames, 4, variant, 0)
    _check(r, cpu)
    r.destroy()


@pytest.mark.parametrize("rayreg", ["0", "1"])
@pytest.mark.parametrize("treelet", [64, 1000])
def test_product_build_partial_node_staging(W, oracle_lib, monkeypatch, treelet, rayreg):
    """Part of the nodes in LDS, the rest behind the L1 (MI355RT_TREELET_MAX): the phased trip of the mixed-mode walk, in both
    forms of the trace kernels (MI355RT_WF_RAYREG: instance-space origin / direction posted to LDS at instance entry, or kept
    in registers and posted with every flush — the host picks by instance count, csrc/k_traverse.hip.h trav_post_at_entry)."""
    monkeypatch.setenv("MI355RT_TREELET_MAX", str(treelet))
    monkeypatch.setenv("MI355RT_WF_RAYREG", rayreg)
    b = pu.bridge_for(W, "instanced1000")
    w, h, depth, frames = 64, 40, 8, (1, 2, 3, 4)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, 1, frames, present=False)
    r = W.WebGPURenderer(0)
    _run(W, r, b, w, h, depth, 1, frames, 4, 2, 0)
    _check(r, cpu)
    r.destroy()


@pytest.mark.parametrize("seed,kw", [
    (21, dict()),
    (22, dict(n_geoms=5, tris_per_geom=120, n_instances=24)),
    (23, dict(n_geoms=1, tris_per_geom=8, n_instances=1)),
    (24, dict(with_textures=True)),
    (25, dict(lens=0.08, n_instances=12)),
    (26, dict(n_geoms=8, tris_per_geom=400, n_instances=300)),     # does not fit LDS
    (27, dict(n_geoms=3, tris_per_geom=300, n_instances=60)),
])
def test_product_build_random_scenes(W, oracle_lib, seed, kw):
    """Random bridge-layout scenes (duplicated / degenerate triangles, 7-triangle leaves, scaled instances, textures, thin
    lens) through the product build of the persistent and the wavefront form."""
    b = random_scene.make(seed, **kw)
    w, h, depth, frames = 96, 64, 8, (1, 2, 3, 4)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, 1, frames, present=False)
    want = cpu.readAccum().view(np.uint32)
    cc = cpu.getCounters()
    for variant, walk in ((1, None), (2, 0), (2, 1)):
        r = W.WebGPURenderer(0)
        _run(W, r, b, w, h, depth, 1, frames, 4, variant, walk)
        got = r.readAccum()
        assert np.array_equal(got.view(np.uint32), want), pu.describe_mismatch("accumulation (variant %d walk %s)" % (variant, walk), got, cpu.readAccum())
        gc = r.getCounters()
        assert {k: gc[k] for k in RAYS} == {k: cc[k] for k in RAYS}
        r.destroy()
