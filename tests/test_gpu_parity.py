"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bar: bit-exact accumulation buffer / G-buffer / history / RGBA8 output and
identical ray counters (north_star tolerance is 1e-4 relative per channel; bit-exact implies it)."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu


CASES = [
    # scene, w, h, depth, spp, frames
    ("cornell", 128, 128, 4, 1, (1, 2, 3, 4)),          # BASELINE config 1, reduced resolution
    ("cornell", 512, 512, 4, 1, (1, 2, 3, 4)),          # BASELINE config 1 as it is written: 512x512, 1 spp x 4 frames, depth 4, whole image
    ("cornell", 67, 45, 8, 1, (1, 2)),                   # ragged size: partial 8x8 tiles
    ("cornell", 64, 64, 8, 4, (1, 2)),                   # SPP > 1 inside one dispatch
    ("cornell", 64, 64, 1, 1, (1,)),                     # MAX_DEPTH = 1: no extension rays
    ("viewer_diamond", 160, 90, 8, 1, (1, 2, 3)),        # config 2: metal floor + 2 instances
    ("viewer_diamond_1k", 160, 90, 8, 1, (1, 2, 3)),     # config 2b: the "~1k tris" diamond BASELINE.json words (968 triangles)
    ("special", 96, 72, 8, 1, (1, 2)),                   # dielectric box + metal + 528 light triangles
    ("mixed", 96, 64, 10, 1, (1, 2)),                    # thin lens (defocus 0.3), GGX, nested dielectrics
    ("mesh", 96, 64, 8, 1, (1, 2)),                      # OBJ cube instances, dielectric, big sphere light
    ("instanced1000", 96, 54, 8, 1, (1, 2)),             # config 3: 1001 instances, 2001-node TLAS
    ("sponza_like", 64, 36, 8, 1, (1, 2)),               # config 4: 263k tris, 8 textures, metal-rough maps
    ("glass_blob", 48, 27, 16, 1, (1, 2)),               # config 5: 205k-tri dielectric, depth 16
]


@pytest.mark.parametrize("scene,w,h,depth,spp,frames", CASES)
def test_pathtrace_parity(W, oracle_lib, gpu_renderer, scene, w, h, depth, spp, frames):
    b = pu.bridge_for(W, scene)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(gpu_renderer, W, b, w, h, depth, spp, frames, present=False)
    pu.drive(cpu, W, b, w, h, depth, spp, frames, present=False)
    pu.assert_parity(gpu_renderer, cpu, check_output=False)
    c = gpu_renderer.getCounters()
    assert c["primary_rays"] == w * h * len(frames)


@pytest.mark.parametrize("scene,w,h,depth,frames", [
    ("cornell", 96, 96, 4, tuple(range(1, 21))),   # crosses frame_count 16 -> 17 (bilinear un-jitter -> nearest, k = 60)
    ("viewer_diamond", 80, 45, 8, (1, 2, 3)),
    ("cornell", 33, 21, 4, (1, 2)),                 # ragged size for the 16x16 post blocks
])
def test_present_parity_live_loop(W, oracle_lib, gpu_renderer, scene, w, h, depth, frames):
    """compute(); present() every frame exactly like the live loop (main.ts:172-173)."""
    b = pu.bridge_for(W, scene)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(gpu_renderer, W, b, w, h, depth, 1, frames, present=True)
    pu.drive(cpu, W, b, w, h, depth, 1, frames, present=True)
    pu.assert_parity(gpu_renderer, cpu, check_output=True)


@pytest.mark.parametrize("scene,w,h,depth,spp,frames", [
    ("cornell", 72, 40, 8, 2, (1, 2)),
    ("instanced1000", 64, 36, 8, 1, (1,)),
    ("mixed", 64, 48, 10, 1, (1, 2)),
])
def test_megakernel_variant_parity(W, oracle_lib, gpu_renderer, scene, w, h, depth, spp, frames):
    """The one-pixel-per-lane kernel form (kept for A/B timing) must agree with the oracle too."""
    b = pu.bridge_for(W, scene)
    cpu = oracle_lib.OracleRenderer()
    gpu_renderer.setKernelVariant(0)
    pu.drive(gpu_renderer, W, b, w, h, depth, spp, frames, present=False)
    pu.drive(cpu, W, b, w, h, depth, spp, frames, present=False)
    pu.assert_parity(gpu_renderer, cpu, check_output=False)


def test_recorder_semantics_frame_count_from_zero(W, oracle_lib, gpu_renderer):
    """VideoRecorder passes frame_count = 0, 1, 2, ... (VideoRecorder.ts:278-280): frames 0 and 1 both
    overwrite, and the post pass at frame 0 has alpha = 1/0."""
    b = pu.bridge_for(W, "cornell")
    cpu = oracle_lib.OracleRenderer()
    for r in (gpu_renderer, cpu):
        pu.drive(r, W, b, 64, 64, 4, 1, (0, 1, 2, 3), present=False)
    pu.assert_parity(gpu_renderer, cpu, check_output=False)
    acc = gpu_renderer.readAccum()
    assert acc[..., 3].max() == 3.0  # N-1 samples accumulated


def test_present_with_recorder_frame_zero(W, oracle_lib, gpu_renderer):
    """present() at frame_count 0 (the recorder's warm-up, VideoRecorder.ts:164-169): alpha = 1/0 and the average
    jitter is non-finite, so the un-jitter footprint leaves the LDS tile and the direct path must take over."""
    b = pu.bridge_for(W, "cornell")
    cpu = oracle_lib.OracleRenderer()
    for r in (gpu_renderer, cpu):
        pu.drive(r, W, b, 48, 40, 4, 1, (0, 1, 2, 3), present=True)
    pu.assert_parity(gpu_renderer, cpu, check_output=True)


def test_accumulation_round_trip_and_reset(W, gpu_renderer):
    b = pu.bridge_for(W, "cornell")
    pu.drive(gpu_renderer, W, b, 64, 48, 4, 1, (1, 2), present=False)
    acc = gpu_renderer.readAccum()
    gpu_renderer.resetAccumulation()
    assert not gpu_renderer.readAccum().any()
    gpu_renderer.writeAccum(acc)  # checkpoint / resume
    gpu_renderer.compute(3)
    resumed = gpu_renderer.readAccum()
    fresh = W.WebGPURenderer(0)
    pu.drive(fresh, W, b, 64, 48, 4, 1, (1, 2, 3), present=False)
    assert np.array_equal(resumed.view(np.uint32), fresh.readAccum().view(np.uint32))
    fresh.destroy()


def test_stripes_partition_is_bitwise_identical(W, gpu_renderer):
    """Interleaved row stripes rendered separately and summed == the full render."""
    b = pu.bridge_for(W, "cornell")
    w, h, frames = 80, 72, (1, 2, 3)
    pu.drive(gpu_renderer, W, b, w, h, 4, 1, frames, present=False)
    full = gpu_renderer.readAccum()
    full_counts = gpu_renderer.getCounters()
    total = np.zeros_like(full)
    counts = {}
    n = 3
    for rank in range(n):
        r = W.WebGPURenderer(0)
        r.setStripes(16, rank, n)
        pu.drive(r, W, b, w, h, 4, 1, frames, present=False)
        part = r.readAccum()
        rows = (np.arange(h) // 16) % n == rank
        assert not part[~rows].any()  # nothing written outside the owned stripes
        total += part
        for k, v in r.getCounters().items():
            counts[k] = counts.get(k, 0) + v
        r.destroy()
    gpu_renderer.setStripes(0, 0, 1)
    assert np.array_equal(total.view(np.uint32), full.view(np.uint32))
    assert counts == full_counts


def test_update_buffer_reports_reallocation(W, gpu_renderer):
    """updateBuffer returns needsRebind only when the buffer had to grow (1.5x policy)."""
    r = W.WebGPURenderer(0)
    small = np.zeros(20 * 10, dtype=np.uint32)
    assert r.updateBuffer("topology", small) is True        # first allocation
    assert r.updateBuffer("topology", small) is False       # fits
    assert r.updateBuffer("topology", np.zeros(20 * 14, dtype=np.uint32)) is False   # within the 1.5x slack
    assert r.updateBuffer("topology", np.zeros(20 * 40, dtype=np.uint32)) is True    # must grow
    r.destroy()


def test_passes_skip_silently_when_not_ready(W):
    """Like the reference passes, compute()/present() do nothing (and do not throw) before resources exist."""
    r = W.WebGPURenderer(0)
    assert r.compute(1) == 2  # RT_SKIPPED
    assert r.present() == 2
    with pytest.raises(W.RendererError):
        r.captureFrame()
    r.destroy()


def test_full_size_properties_1080p(W, oracle_lib, gpu_renderer):
    """BASELINE metric size (1920x1080, depth 8): properties that need no oracle run —
    sample count, finiteness, energy bounds, determinism and stripe-sum identity."""
    b = pu.bridge_for(W, "cornell")
    w, h = 1920, 1080
    pu.drive(gpu_renderer, W, b, w, h, 8, 1, (1, 2, 3, 4), present=True, detailed=False)
    a1 = gpu_renderer.readAccum()
    out1 = gpu_renderer.captureFrame()["data"].copy()
    c1 = gpu_renderer.getCounters()
    assert np.isfinite(a1).all() and (a1[..., 3] == 4.0).all() and (a1[..., :3] >= 0).all()
    assert a1[..., :3].max() <= 4 * 20.0 * 8  # no sample can exceed light radiance x depth
    assert c1["primary_rays"] == w * h * 4
    assert 0 < c1["shadow_rays"] <= c1["extension_rays"] + c1["primary_rays"]
    again = W.WebGPURenderer(0)
    pu.drive(again, W, b, w, h, 8, 1, (1, 2, 3, 4), present=True, detailed=False)
    assert np.array_equal(a1.view(np.uint32), again.readAccum().view(np.uint32))  # deterministic
    assert np.array_equal(out1, again.captureFrame()["data"])
    assert again.getCounters() == c1
    again.destroy()
    # oracle band at full size: the CPU renders only rows [512, 520) of the same 1080p frames
    cpu = oracle_lib.OracleRenderer()
    cpu.setStripes(8, 64, 135)
    pu.drive(cpu, W, b, w, h, 8, 1, (1, 2, 3, 4), present=False)
    band = cpu.readAccum()[512:520]
    assert np.array_equal(band.view(np.uint32), a1[512:520].view(np.uint32))


def test_scene_without_lights(W, oracle_lib, gpu_renderer):
    """light_count == 0: sample_light_source returns before drawing (Raytracer.wgsl:346-349), so no NEE draws,
    no shadow rays, and only directly visible emitters contribute."""
    b = pu.bridge_for(W, "cornell")

    class NoLights:
        def __init__(self, inner):
            self._b = inner
        def __getattr__(self, k):
            return getattr(self._b, k)
        lights = np.zeros(0, dtype=np.uint32)
        lightCount = 0

    nb = NoLights(b)
    cpu = oracle_lib.OracleRenderer()
    for r in (gpu_renderer, cpu):
        pu.drive(r, W, nb, 64, 64, 6, 1, (1, 2), present=True)
    pu.assert_parity(gpu_renderer, cpu, check_output=True)
    assert gpu_renderer.getCounters()["shadow_rays"] == 0


def test_reupload_larger_scene_and_resize(W, oracle_lib, gpu_renderer):
    """Scene swap on a live context: buffers grow (needsRebind), derived records are rebuilt, the screen is
    re-allocated; the result must equal a fresh oracle run of the second scene (totalFrames carried over)."""
    small, big = pu.bridge_for(W, "cornell"), pu.bridge_for(W, "special")
    cpu = oracle_lib.OracleRenderer()
    for r in (gpu_renderer, cpu):
        pu.drive(r, W, small, 48, 48, 4, 1, (1, 2), present=True)
        pu.drive(r, W, big, 80, 56, 8, 1, (1, 2, 3), present=True)   # same renderer: totalFrames keeps counting
    pu.assert_parity(gpu_renderer, cpu, check_output=True)


def _band_rows(h, div):
    return (np.arange(h) // 8) % div == 0


def test_bench_workload_parity_1080p(W, oracle_lib, gpu_renderer):
    """Exactly what bench.py times — Cornell 1920x1080, depth 8, computeBatch(1..32), computeBatch(33..64), present() —
    against the oracle on the 8-row bands (y // 8) % 27 == 0 of the same 64 frames: accumulation band bit for bit, the
    RGBA8 rows whose post-pass footprint (2-pixel halo) lies inside a band, and all six counters of a second, band-only
    GPU pass with the detailed-counter kernel variant."""
    b = pu.bridge_for(W, "cornell")
    w, h, depth, div = 1920, 1080, 8, 27
    frames = list(range(1, 65))
    gpu_renderer.buildPipeline(depth, 1)
    W.upload_scene(gpu_renderer, b, w, h)
    gpu_renderer.resetCounters()
    gpu_renderer.computeBatch(frames[:32])
    gpu_renderer.computeBatch(frames[32:])
    gpu_renderer.present()
    gpu_renderer.sync()
    acc = gpu_renderer.readAccum()
    rgba = gpu_renderer.captureFrame()["data"].copy()
    c = gpu_renderer.getCounters()
    assert np.isfinite(acc).all() and (acc[..., 3] == 64.0).all()
    assert c["primary_rays"] == w * h * 64

    cpu = oracle_lib.OracleRenderer()
    cpu.buildPipeline(depth, 1)
    W.upload_scene(cpu, b, w, h)
    cpu.setStripes(8, 0, div)
    cpu.resetCounters()
    for f in frames:
        cpu.compute(f)
    rows = _band_rows(h, div)
    assert int(rows.sum()) == 40
    cacc = cpu.readAccum()
    assert np.array_equal(cacc[rows].view(np.uint32), acc[rows].view(np.uint32)), \
        pu.describe_mismatch("accumulation band", acc[rows], cacc[rows])
    assert np.array_equal(gpu_renderer.readUniforms(), cpu.readUniforms())
    cpu.present()    # rows 2..5 of every band only read accumulation texels of that band (5x5 footprint)
    inner = rows & (np.arange(h) % 8 >= 2) & (np.arange(h) % 8 <= 5)
    crgba = cpu.captureFrame()["data"]
    assert np.array_equal(crgba[inner], rgba[inner]), pu.describe_mismatch("RGBA8 band interior", rgba[inner], crgba[inner])

    band = W.WebGPURenderer(0)          # same 64 frames, the band only, detailed counters: equals the oracle's counts
    band.buildPipeline(depth, 1)
    W.upload_scene(band, b, w, h)
    band.setStripes(8, 0, div)
    band.setCounting(True)
    band.resetCounters()
    band.computeBatch(frames[:32])
    band.computeBatch(frames[32:])
    band.sync()
    assert band.getCounters() == cpu.getCounters()
    assert np.array_equal(band.readAccum()[rows].view(np.uint32), cacc[rows].view(np.uint32))
    band.destroy()


@pytest.mark.parametrize("scene,w,h,depth,nframes,batch,div", [
    ("viewer_diamond", 1280, 720, 8, 16, 16, 9),     # config 2 at its BASELINE size and frame count (persistent kernel)
    ("instanced1000", 1920, 1080, 8, 64, 32, 135),   # config 3, full size, ALL 64 frames as the two 32-frame batches bench.py times (wavefront form)
    ("sponza_like", 1920, 1080, 8, 64, 32, 135),     # config 4, likewise
    ("glass_blob", 3840, 2160, 16, 32, 32, 270),     # config 5 at 4K, depth 16: one 32-frame batch of its 256 frames
])
def test_full_size_oracle_band(W, oracle_lib, gpu_renderer, scene, w, h, depth, nframes, batch, div):
    """BASELINE configs 2-5 at full resolution through the batched dispatch the benchmark uses (auto kernel form: the
    wavefront form for configs 3-5): the GPU renders whole frames, the oracle the 8-row bands (y // 8) % div == 0; the
    bands must agree bit for bit, the frames must be finite with the right sample count."""
    b = pu.bridge_for(W, scene)
    frames = list(range(1, nframes + 1))
    gpu_renderer.buildPipeline(depth, 1)
    W.upload_scene(gpu_renderer, b, w, h)
    gpu_renderer.resetCounters()
    for i in range(0, nframes, batch):
        gpu_renderer.computeBatch(frames[i:i + batch])
    gpu_renderer.sync()
    acc = gpu_renderer.readAccum()
    assert np.isfinite(acc).all() and (acc[..., 3] == nframes).all()
    c = gpu_renderer.getCounters()
    assert c["primary_rays"] == w * h * nframes
    cpu = oracle_lib.OracleRenderer()
    cpu.buildPipeline(depth, 1)
    W.upload_scene(cpu, b, w, h)
    cpu.setStripes(8, 0, div)
    for f in frames:
        cpu.compute(f)
    rows = _band_rows(h, div)
    cacc = cpu.readAccum()
    assert np.array_equal(cacc[rows].view(np.uint32), acc[rows].view(np.uint32)), \
        pu.describe_mismatch("accumulation band", acc[rows], cacc[rows])


@pytest.mark.parametrize("scene,w,h,depth,spp,frames,batch", [
    ("cornell", 96, 72, 8, 1, tuple(range(1, 13)), 4),      # 3 batches of 4
    ("cornell", 64, 48, 6, 2, (0, 1, 2, 3, 4), 5),          # recorder semantics (frame 0 and 1 overwrite), SPP 2
    ("special", 80, 56, 8, 1, (1, 2, 3, 4, 5, 6, 7), 3),    # ragged last batch (3 + 3 + 1)
    ("instanced1000", 64, 36, 8, 1, (1, 2, 3, 4), 4),       # global-memory kernel form
    ("mixed", 64, 48, 10, 1, (1, 2, 3), 3),                 # thin lens
])
def test_batched_dispatch_equals_sequential_computes(W, oracle_lib, gpu_renderer, scene, w, h, depth, spp, frames, batch):
    """rt_compute_batch == the same compute() calls one by one: accumulation, last frame's G-buffer, uniforms, counters."""
    b = pu.bridge_for(W, scene)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, spp, frames, present=False)
    gpu_renderer.buildPipeline(depth, spp)
    W.upload_scene(gpu_renderer, b, w, h)
    gpu_renderer.setCounting(True)
    gpu_renderer.resetCounters()
    for i in range(0, len(frames), batch):
        gpu_renderer.computeBatch(frames[i:i + batch])
    gpu_renderer.sync()
    pu.assert_parity(gpu_renderer, cpu, check_output=False)
    gpu_renderer.present()
    cpu.present()
    pu.assert_parity(gpu_renderer, cpu, check_output=True, check_counters=False)


def test_batched_dispatch_with_stripes(W, gpu_renderer):
    b = pu.bridge_for(W, "cornell")
    w, h, frames = 80, 72, tuple(range(1, 9))
    pu.drive(gpu_renderer, W, b, w, h, 6, 1, frames, present=False)
    full = gpu_renderer.readAccum()
    total = np.zeros_like(full)
    for rank in range(2):
        r = W.WebGPURenderer(0)
        r.setStripes(16, rank, 2)
        r.buildPipeline(6, 1)
        W.upload_scene(r, b, w, h)
        r.computeBatch(frames)
        total += r.readAccum()
        r.destroy()
    assert np.array_equal(total.view(np.uint32), full.view(np.uint32))


@pytest.mark.parametrize("scene,w,h,depth,frames,batch", [
    ("cornell", 96, 72, 8, (1, 2, 3, 4), 1),
    ("cornell", 64, 48, 6, (0, 1, 2, 3, 4, 5), 3),
    ("special", 80, 56, 8, (1, 2, 3), 3),
    ("mixed", 64, 48, 10, (1, 2), 2),
    ("instanced1000", 96, 54, 8, (1, 2, 3, 4), 2),
    ("viewer_diamond_1k", 96, 54, 8, (1, 2, 3, 4), 4),
    ("sponza_like", 64, 36, 8, (1, 2), 2),
    ("glass_blob", 48, 27, 16, (1, 2), 1),
    ("cornell", 33, 21, 1, (1, 2), 2),                      # MAX_DEPTH = 1: no extension rays at all
])
@pytest.mark.parametrize("walk", [1, 0])
def test_wavefront_form_parity(W, oracle_lib, gpu_renderer, scene, w, h, depth, frames, batch, walk):
    """Kernel variant 2 (shade / trace stages with the path state in HBM) against the oracle: accumulation, G-buffer,
    uniforms and all six counters; then present().  walk 1 = child-pair records (the default), 0 = the single-node walk
    kept for A/B measurements: both must reach every node the reference reaches, in its order (nodes_visited, tris_tested)."""
    b = pu.bridge_for(W, scene)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, 1, frames, present=False)
    gpu_renderer.setKernelVariant(2)
    gpu_renderer.setWalk(walk)
    gpu_renderer.buildPipeline(depth, 1)
    W.upload_scene(gpu_renderer, b, w, h)
    gpu_renderer.setCounting(True)
    gpu_renderer.resetCounters()
    for i in range(0, len(frames), batch):
        if batch == 1:
            gpu_renderer.compute(frames[i])
        else:
            gpu_renderer.computeBatch(frames[i:i + batch])
    gpu_renderer.sync()
    pu.assert_parity(gpu_renderer, cpu, check_output=False)
    gpu_renderer.present()
    cpu.present()
    pu.assert_parity(gpu_renderer, cpu, check_output=True, check_counters=False)


@pytest.mark.parametrize("scene,treelet,variant", [
    ("sponza_like", 64, 2), ("sponza_like", 100000, 2), ("instanced1000", 300, 2), ("instanced1000", 100000, 1),
    ("glass_blob", 1000, 2),
])
def test_partial_node_staging_parity(W, oracle_lib, monkeypatch, scene, treelet, variant):
    """MI355RT_TREELET_MAX = n stages the first n traversal nodes in LDS beside the L1-served rest (not the default:
    nodes are staged all or not at all).  That path interleaves LDS-resident and global node steps and defers instance
    entries; per ray it must still do exactly what the oracle does, counters included."""
    monkeypatch.setenv("MI355RT_TREELET_MAX", str(treelet))
    b = pu.bridge_for(W, scene)
    w, h, depth, frames = 64, 40, 8, (1, 2, 3, 4)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, 1, frames, present=False)
    r = W.WebGPURenderer(0)
    r.setKernelVariant(variant)
    r.buildPipeline(depth, 1)
    W.upload_scene(r, b, w, h)
    r.setCounting(True)
    r.resetCounters()
    r.computeBatch(list(frames))
    r.sync()
    pu.assert_parity(r, cpu, check_output=False)


@pytest.mark.parametrize("scene,variant", [("cornell", 1), ("cornell", 2), ("mesh", 2), ("special", 1), ("mixed", 2), ("viewer_diamond", 3)])
def test_global_paths_on_small_scenes(W, oracle_lib, monkeypatch, scene, variant):
    """MI355RT_NO_LDS_STAGING=1: small scenes through the code the big ones use — nodes, triangle records and instance
    rows read through the L1, instance entries deferred and batched — with the oracle's result, counters included."""
    monkeypatch.setenv("MI355RT_NO_LDS_STAGING", "1")
    b = pu.bridge_for(W, scene)
    w, h, depth, frames = 72, 48, 8, (1, 2, 3, 4)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, 1, frames, present=False)
    r = W.WebGPURenderer(0)
    r.setKernelVariant(variant)
    r.buildPipeline(depth, 1)
    W.upload_scene(r, b, w, h)
    r.setCounting(True)
    r.resetCounters()
    r.computeBatch(list(frames))
    r.sync()
    pu.assert_parity(r, cpu, check_output=False)
    r.destroy()


@pytest.mark.parametrize("env,value", [("MI355RT_WF_OVERLAP", "0"), ("MI355RT_WF_BLOCK", "512"), ("MI355RT_SHADE_BLOCKS_PER_CU", "3")])
def test_wavefront_launch_knobs_keep_parity(W, oracle_lib, monkeypatch, env, value):
    """The launch-shape knobs of the wavefront form (trace kernels on one stream instead of two, 512-thread trace
    workgroups, fewer shade workgroups) change scheduling only: accumulation and counters still equal the oracle's."""
    monkeypatch.setenv(env, value)
    b = pu.bridge_for(W, "instanced1000")
    w, h, depth, frames = 80, 48, 8, (1, 2, 3, 4, 5)
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, depth, 1, frames, present=False)
    r = W.WebGPURenderer(0)
    r.setKernelVariant(2)
    r.buildPipeline(depth, 1)
    W.upload_scene(r, b, w, h)
    r.setCounting(True)
    r.resetCounters()
    r.computeBatch(list(frames[:3]))
    r.computeBatch(list(frames[3:]))
    r.sync()
    pu.assert_parity(r, cpu, check_output=False)
    r.destroy()


@pytest.mark.parametrize("variant", [2, 3])
def test_wavefront_form_with_stripes(W, gpu_renderer, variant):
    """Sharded render of a scene that does not fit LDS (auto = wavefront form): the ranks' stripes sum to the
    unsharded image bit for bit, in batched and single dispatches."""
    b = pu.bridge_for(W, "sponza_like")
    w, h, frames = 80, 72, tuple(range(1, 6))
    gpu_renderer.setKernelVariant(1)
    pu.drive(gpu_renderer, W, b, w, h, 6, 1, frames, present=False)
    full = gpu_renderer.readAccum()
    total = np.zeros_like(full)
    for rank in range(3):
        r = W.WebGPURenderer(0)
        r.setKernelVariant(variant)
        r.setStripes(16, rank, 3)
        r.buildPipeline(6, 1)
        W.upload_scene(r, b, w, h)
        r.computeBatch(frames[:4])     # auto: >= 4 frames per dispatch -> wavefront form
        r.compute(frames[4])           # auto: a single frame -> persistent kernel
        total += r.readAccum()
        r.destroy()
    assert np.array_equal(total.view(np.uint32), full.view(np.uint32))


def test_kernel_forms_agree_bitwise(W):
    """Megakernel (0), persistent (1), wavefront (2) and auto (3) produce the same accumulation buffer, G-buffer and ray
    counters on a scene that takes the global-memory path."""
    b = pu.bridge_for(W, "instanced1000")
    w, h, frames = 160, 90, (1, 2, 3)
    ref = None
    for variant in (0, 1, 2, 3):
        r = W.WebGPURenderer(0)
        r.setKernelVariant(variant)
        r.buildPipeline(8, 1)
        W.upload_scene(r, b, w, h)
        r.resetCounters()
        for f in frames:
            r.compute(f)
        r.sync()
        got = (r.readAccum().view(np.uint32).copy(), r.readGBuffer()[1].view(np.uint32).copy(), r.getCounters())
        r.destroy()
        if ref is None:
            ref = got
        else:
            assert np.array_equal(ref[0], got[0]), "variant %d accumulation differs" % variant
            assert np.array_equal(ref[1], got[1])
            assert ref[2] == got[2]


def test_read_gbuffer_after_resize_reads_the_new_planes(W, gpu_renderer):
    """Lookahead keeps pointers to the G-buffer planes of the batch it traced; rt_resize frees them.  A readGBuffer() between
    the resize (to a LARGER screen: a stale pointer would be read out of bounds) and the next compute() must read the fresh,
    zero-initialised planes of the new size (ResourceManager.ts:97-142), as without lookahead."""
    b = pu.bridge_for(W, "cornell")
    r = gpu_renderer
    r.buildPipeline(4, 1)
    W.upload_scene(r, b, 48, 32)
    r.setLookahead(8)
    r.compute(1)
    r.compute(2)        # served from the frames traced ahead: the G-buffer of this frame lives in a batch plane
    r.sync()
    assert r.readGBuffer()[2].min() < 1.0          # something was hit
    r.updateScreenSize(160, 120)
    alb, nid, dep = r.readGBuffer()
    assert alb.shape == (120, 160, 4) and not alb.any() and not nid.any() and not dep.any()
    r.compute(1)        # and the renderer goes on at the new size
    r.sync()
    assert r.readGBuffer()[2].shape == (120, 160) and r.readGBuffer()[2].min() < 1.0


def test_malformed_scene_arrays_are_refused_not_followed(W, gpu_renderer):
    """WebGPU's robust buffer access keeps the reference alive on a bad index; a HIP kernel would fault or spin. Every
    index the kernels follow is checked once per upload (k_validate_scene) and compute() fails with the reason."""
    b = pu.bridge_for(W, "mixed")
    r = gpu_renderer
    r.buildPipeline(4, 1)
    W.upload_scene(r, b, 48, 32)
    r.compute(1)
    r.sync()
    n_verts, n_tris = len(b.vertices) // 4, len(b.mesh_topology) // 20
    n_inst = len(b.instances) // 36

    def expect_refusal(what):
        with pytest.raises(W.RendererError, match="out of range"):
            r.compute(2)
        r.sync()
        assert what in str(r.L.rt_last_error(r.ctx))

    topo = np.array(b.mesh_topology, dtype=np.uint32)
    topo[20 * 5 + 1] = n_verts                      # a vertex id one past the end
    r.updateBuffer("topology", topo)
    expect_refusal("1 triangles")
    r.updateBuffer("topology", b.mesh_topology)
    blas = np.array(b.blas, dtype=np.float32)
    blas[8 * 2 + 3:8 * 2 + 4].view(np.uint32)[0] = 1  # skip pointer that goes backwards: the walk would never end
    r.updateCombinedBVH(b.tlas, blas)
    expect_refusal("1 BLAS nodes")
    blas = np.array(b.blas, dtype=np.float32)
    leaf = int(np.flatnonzero(blas.view(np.uint32)[7::8])[0])
    blas.view(np.uint32)[leaf * 8 + 7] = ((n_tris - 1) << 3) | 3   # a leaf whose triangle range runs past the topology
    r.updateCombinedBVH(b.tlas, blas)
    expect_refusal("1 BLAS nodes")
    tlas = np.array(b.tlas, dtype=np.float32)
    tl = int(np.flatnonzero(tlas.view(np.uint32)[7::8])[0])
    tlas.view(np.uint32)[tl * 8 + 7] = (n_inst << 3) | 1           # a TLAS leaf naming an instance that does not exist
    r.updateCombinedBVH(tlas, b.blas)
    expect_refusal("1 TLAS nodes")
    r.updateCombinedBVH(b.tlas, b.blas)
    inst = np.array(b.instances, dtype=np.float32)
    inst.view(np.uint32)[32] = len(b.blas) // 8 + 5                # BLAS offset outside the node buffer
    r.updateBuffer("instance", inst)
    expect_refusal("1 instances")
    r.updateBuffer("instance", b.instances)
    lights = np.array(b.lights, dtype=np.uint32)
    lights[1] = n_tris
    r.updateBuffer("lights", lights)
    expect_refusal("1 light references")
    r.updateBuffer("lights", b.lights)
    r.compute(2)                                                   # the intact scene renders again
    r.sync()


@pytest.mark.parametrize("scene", ["cornell", "viewer_diamond", "mixed", "instanced1000", "sponza_like", "glass_blob"])
def test_traversal_array_is_the_same_tree_renumbered(W, gpu_renderer, scene):
    """tnodes (csrc/k_treelet.hip.h): a permutation of the bridge's nodes with both successors explicit. Following them
    from the roots must walk every BLAS and the TLAS in the ORIGINAL pre-order, with the original boxes and leaf words;
    the largest-area nodes come first."""
    import ctypes
    b = pu.bridge_for(W, scene)
    r = gpu_renderer
    r.buildPipeline(4, 1)
    W.upload_scene(r, b, 32, 16)
    tl, bl = np.asarray(b.tlas, np.float32).reshape(-1, 8), np.asarray(b.blas, np.float32).reshape(-1, 8)
    nodes = np.concatenate([tl, bl])
    n, n_tlas = len(nodes), len(tl)
    inst = np.asarray(b.instances, np.float32).reshape(-1, 36).view(np.uint32)
    tn = np.zeros((n, 8), np.float32)
    new = np.zeros(n, np.uint32)
    roots = np.zeros(len(inst), np.uint32)
    vp = ctypes.c_void_p
    got = r.L.rt_debug_read_traversal_nodes(r.ctx, tn.ctypes.data_as(vp), new.ctypes.data_as(vp), roots.ctypes.data_as(vp), n)
    assert got == n
    assert sorted(new.tolist()) == list(range(n)) and new[0] == 0           # a permutation; the TLAS root stays node 0
    u, tu = nodes.view(np.uint32), tn.view(np.uint32)
    END, INNER = 0xffffffff, 0x80000000
    # boxes travel with their node; leaves keep their word; inner nodes point at their first child, skips at the original target
    assert np.array_equal(tu[new][:, [0, 1, 2, 4, 5, 6]], u[:, [0, 1, 2, 4, 5, 6]])
    leaf = u[:, 7] != 0
    assert np.array_equal(tu[new[leaf], 7], u[leaf, 7])
    inner = np.flatnonzero(~leaf)
    reach = np.zeros(n, bool)           # nodes inside a TLAS / BLAS range that a walk can reach
    reach[:int(u[0, 3])] = True
    skip_target = np.full(n, -1, np.int64)
    skip_target[:n_tlas] = np.where(u[:n_tlas, 3] < u[0, 3], u[:n_tlas, 3].astype(np.int64), -1)
    for off in sorted(set(inst[:, 32].tolist())):
        root = n_tlas + off
        end = root + int(u[root, 3])
        reach[root:end] = True
        tgt = root + u[root:end, 3].astype(np.int64)
        skip_target[root:end] = np.where(tgt < end, tgt, -1)
    ri = inner[reach[inner]]
    assert np.array_equal(tu[new[ri], 7], INNER | new[ri + 1])
    rr = np.flatnonzero(reach)
    expect = np.where(skip_target[rr] >= 0, new[np.maximum(skip_target[rr], 0)], END).astype(np.uint32)
    assert np.array_equal(tu[new[rr], 3], expect)
    assert np.array_equal(roots, new[n_tlas + inst[:, 32]])
    # treelet first: the head of the array holds the largest boxes (single-BLAS scenes: the weight is the box area)
    if scene == "sponza_like":
        ext = tn[:, 4:7] - tn[:, 0:3]
        area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0])
        assert int((area[:5120] >= area[5120:].max()).sum()) >= 4000


@pytest.mark.parametrize("scene", ["cornell", "viewer_diamond", "viewer_diamond_1k", "special", "mixed", "mesh", "instanced1000",
                                   "sponza_like", "glass_blob"])
def test_pair_records_equal_the_numpy_restatement(W, gpu_renderer, scene):
    """The child-pair records the trace kernels walk (csrc/k_pairs.hip.h, built on the GPU at upload) against
    tests/pair_layout.py, byte for byte — the arrays tests/test_pairwalk_model.py walks on the host against the oracle."""
    import ctypes
    import pair_layout
    b = pu.bridge_for(W, scene)
    r = gpu_renderer
    r.buildPipeline(4, 1)
    W.upload_scene(r, b, 32, 16)
    pairs, troot, inst_root = pair_layout.build(b.tlas, b.blas, b.instances)
    got_pairs = np.zeros((len(pairs) + 1, 16), np.float32)
    got_roots = np.zeros((len(inst_root) + 1, 8), np.float32)
    vp = ctypes.c_void_p
    n = r.L.rt_debug_read_pairs(r.ctx, got_pairs.ctypes.data_as(vp), got_roots.ctypes.data_as(vp), len(got_pairs))
    assert n == len(pairs)
    assert np.array_equal(got_pairs[:n].view(np.uint32), pairs.view(np.uint32))
    assert np.array_equal(got_roots[:-1].view(np.uint32), inst_root.view(np.uint32))
    assert np.array_equal(got_roots[-1].view(np.uint32), troot.view(np.uint32))


@pytest.mark.parametrize("scene,w,h,depth", [("cornell", 96, 72, 6), ("instanced1000", 64, 36, 8)])
def test_lookahead_live_loop_parity(W, oracle_lib, gpu_renderer, scene, w, h, depth):
    """rt_set_lookahead: consecutive compute(f); present() calls are traced ahead as batches (1, 2, 4, 8, ... frames).  Every
    frame's accumulation buffer, G-buffer and presented image must be those of one dispatch per frame — checked against the
    oracle after frames in the middle of a traced-ahead batch, across a camera change (which drops what was traced ahead)
    and across a restart of the frame count."""
    b = pu.bridge_for(W, scene)
    cpu = oracle_lib.OracleRenderer()
    gpu_renderer.setLookahead(16)
    for r in (gpu_renderer, cpu):
        r.buildPipeline(depth, 1)
        W.upload_scene(r, b, w, h)

    def frames(lo, hi, present):
        for r in (gpu_renderer, cpu):
            for f in range(lo, hi + 1):
                r.compute(f)
                if present:
                    r.present()
            r.sync()

    frames(1, 6, True)             # dispatches of 1, 2 (2-3), 4 (4-7) frames: frame 6 sits inside a batch
    pu.assert_parity(gpu_renderer, cpu, check_output=True, check_counters=False)
    frames(7, 11, False)           # 7 consumed, 8-15 traced ahead, 11 consumed
    pu.assert_parity(gpu_renderer, cpu, check_output=False, check_counters=False)   # includes the G-buffer of frame 11
    cam = np.array(b.cameraData, dtype=np.float32).copy()
    cam[0] += 0.05                 # the camera moves: frames 12-15 traced ahead are stale and must not be used
    for r in (gpu_renderer, cpu):
        r.updateSceneUniforms(cam, 0, b.lightCount)
        r.resetAccumulation()
    frames(1, 9, True)
    pu.assert_parity(gpu_renderer, cpu, check_output=True, check_counters=False)
    for r in (gpu_renderer, cpu):  # frame count restarts without any other change: not a continuation
        r.resetAccumulation()
    frames(1, 3, False)
    pu.assert_parity(gpu_renderer, cpu, check_output=False, check_counters=False)
    gpu_renderer.setLookahead(0)
