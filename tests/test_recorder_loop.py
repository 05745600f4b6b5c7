"""The offline frame loop (VideoRecorder cadence): CPU test of the cadence logic with a recording stand-in,
GPU test of image parity with the oracle driven by the same loop and a deterministic clock."""
import numpy as np
import pytest

import parity_util as pu


class FakeClock:
    """Every call advances 30 ms: a 20-dispatch batch 'takes' 60 ms, so presents are throttled like on a real GPU."""
    def __init__(self, step=0.030):
        self.t, self.step = 0.0, step
    def __call__(self):
        self.t += self.step
        return self.t


class Recording:
    def __init__(self):
        self.calls = []
    def __getattr__(self, name):
        def f(*a):
            self.calls.append((name,) + tuple(list(x) if isinstance(x, range) else x
                                              for x in a if isinstance(x, (int, str, range))))
            return False
        return f


def test_cadence_matches_the_reference_recorder(W):
    b = pu.bridge_for(W, "cornell")
    rec = Recording()
    loop = W.FrameLoop(rec, b, 64, 48, batch=20, clock=FakeClock())
    loop.warm_up()
    names = [c[0] for c in rec.calls]
    # updateSceneBuffers order (VideoRecorder.ts:231-268), then 5 x {compute(k), present, fence, reset}
    assert names[:9] == ["updateCombinedBVH", "updateBuffer", "updateCombinedGeometry", "updateBuffer", "updateBuffer",
                         "updateBuffer", "updateSceneUniforms", "recreateBindGroup", "resetAccumulation"] or \
        names[:8] == ["updateCombinedBVH", "updateBuffer", "updateCombinedGeometry", "updateBuffer", "updateBuffer",
                      "updateBuffer", "updateSceneUniforms", "resetAccumulation"]
    warm = [c for c in rec.calls if c[0] == "compute"]
    assert [c[1] for c in warm] == [0, 1, 2, 3, 4]
    rec.calls.clear()
    presents = loop.render_frame(64)
    batches = [c[1] for c in rec.calls if c[0] == "computeBatch"]   # the batch loop is issued as batched dispatches
    assert [f for bt in batches for f in bt] == list(range(64))    # frame_count = 0 .. N-1 (recorder semantics)
    assert all(1 <= len(bt) <= 50 for bt in batches)
    assert presents == sum(1 for c in rec.calls if c[0] == "present") >= 1
    assert rec.calls[-2][0] == "present" and rec.calls[-1][0] == "sync"   # the finished batch always presents
    assert 1 <= loop.current_batch_size <= 50      # hard cap of 50 dispatches per batch


@pytest.mark.gpu
def test_recorder_loop_image_parity(W, oracle_lib):
    W._build.build_rt()
    b = pu.bridge_for(W, "cornell")
    imgs = []
    for r in (W.WebGPURenderer(0), oracle_lib.OracleRenderer()):
        r.buildPipeline(4, 1)
        r.updateScreenSize(64, 48)
        loop = W.FrameLoop(r, b, 64, 48, batch=4, clock=FakeClock())
        frames = loop.render_frames(2, fps=30, spp=8)
        imgs.append((frames, r.readAccum()))
    (gf, ga), (cf, ca) = imgs
    assert np.array_equal(ga.view(np.uint32), ca.view(np.uint32))
    assert ga[..., 3].max() == 7.0                  # N-1 samples: frames 0 and 1 both overwrite
    for g, c in zip(gf, cf):
        assert np.array_equal(g, c)


def test_live_loop_follows_main_ts(W):
    """renderFrame of src/main.ts:119-181: world.update every `update_interval` frames at t = total / interval / 60, the
    re-sync order BVH -> instance -> draw_commands -> (geometry, topology, lights) -> uniforms -> reset, then
    compute(frameCount) + present with frameCount restarting at 1."""
    class Bridge:
        def __init__(self):
            self.hasNewData = self.hasNewGeometry = True
            self.updates = []
            self.lightCount = 2
            for k in ("tlas", "blas", "instances", "draw_commands", "vertices", "normals", "uvs", "mesh_topology", "lights", "cameraData"):
                setattr(self, k, k)

        def update(self, t):
            self.updates.append(t)
            self.hasNewData = self.hasNewGeometry = True

        def updateCamera(self, w, h):
            pass

    rec, b = Recording(), Bridge()
    loop = W.LiveLoop(rec, b, 64, 48, update_interval=3)
    for _ in range(8):
        loop.render_frame()
    assert rec.calls[0][0] == "setLookahead"      # the loop lets the library trace a still scene's consecutive frames ahead
    rec.calls = rec.calls[1:]
    # ... and tells it how many frames are left until the world moves again, so that nothing is traced ahead in vain
    assert [c[1] for c in rec.calls if c[0] == "setLookaheadLimit"] == [3, 2, 1, 3, 2, 1, 3, 2]
    rec.calls = [c for c in rec.calls if c[0] != "setLookaheadLimit"]
    names = [c[0] for c in rec.calls]
    first_sync = ["updateCombinedBVH", "updateBuffer", "updateBuffer", "updateCombinedGeometry", "updateBuffer", "updateBuffer",
                  "updateSceneUniforms", "resetAccumulation"]
    assert names[:8] == first_sync and names[8:10] == ["compute", "present"]
    kinds = [c[1] for c in rec.calls[:8] if c[0] == "updateBuffer"]
    assert kinds == ["instance", "draw_commands", "topology", "lights"]
    assert [c[1] for c in rec.calls if c[0] == "compute"] == [1, 2, 3, 1, 2, 3, 1, 2]      # accumulation restarts after an update
    assert b.updates == [3 / 3 / 60, 6 / 3 / 60]
    assert names.count("resetAccumulation") == 3
    # update_interval <= 0: the world is never advanced, one sync, frameCount keeps counting
    rec2, b2 = Recording(), Bridge()
    loop2 = W.LiveLoop(rec2, b2, 64, 48, update_interval=0)
    for _ in range(4):
        loop2.render_frame()
    rec2.calls = [c for c in rec2.calls if c[0] != "setLookahead"]
    assert b2.updates == [] and [c[1] for c in rec2.calls if c[0] == "compute"] == [1, 2, 3, 4]


# ------------------------------------------------------------------------------------------------ frames to disk, job split
def test_job_list_is_the_hosts_queue(W):
    """src/main.ts:278-290: for (f = 0; f < totalFrames; f += batchSize) push {start: f, count: min(batchSize, total - f)}"""
    assert W.job_list(90, 20) == [(0, 20), (20, 20), (40, 20), (60, 20), (80, 10)]
    assert W.job_list(5, 20) == [(0, 5)] and W.job_list(0, 20) == [] and W.job_list(3, 1) == [(0, 1), (1, 1), (2, 1)]


def test_written_png_decodes_to_the_frame(W, tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)
    path = str(tmp_path / "f.png")
    W.write_png(path, img)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(path).convert("RGBA")), img)      # an independent decoder
    W._build.build_tex()
    assert np.array_equal(W.textures.decode_image(open(path, "rb").read()), img)  # and this repo's own


def _runner(W, make_renderer, out_dir, fmt="png"):
    def make_bridge():
        b = W.WorldBridge()
        b.loadScene("cornell")
        return b
    return W.FrameJobRunner(make_renderer, make_bridge, 48, 32, fps=30, spp=6, depth=4, batch=4, out_dir=out_dir, fmt=fmt,
                            clock=FakeClock)


def _frames_on_disk(d):
    import os
    return {n: open(os.path.join(d, n), "rb").read() for n in sorted(os.listdir(d)) if n.startswith("frame_")}


def test_frame_range_sharding_is_rank_independent(W, oracle_lib, tmp_path):
    """Frame ranges over ranks (the reference's distribution unit): every frame file of a 3-rank run equals the 1-rank run's.
    The CPU oracle stands in for the renderer here (same method surface); the GPU version is below."""
    one, three = str(tmp_path / "one"), str(tmp_path / "three")
    _runner(W, oracle_lib.OracleRenderer, one).run(5, rank=0, world=1, job_batch=2)
    for rank in range(3):
        _runner(W, oracle_lib.OracleRenderer, three).run(5, rank=rank, world=3, job_batch=2)
    a, b = _frames_on_disk(one), _frames_on_disk(three)
    assert sorted(a) == ["frame_%06d.png" % i for i in range(5)] and a == b
    import json, os
    m = [json.load(open(os.path.join(three, "manifest_rank%d.json" % r))) for r in range(3)]
    assert [x["jobs"] for x in m] == [[[0, 2]], [[2, 2]], [[4, 1]]]


@pytest.mark.gpu
def test_frame_jobs_on_the_gpu_equal_the_oracle_and_the_single_rank_run(W, oracle_lib, tmp_path):
    W._build.build_rt()
    gpu1, gpu2, cpu = str(tmp_path / "g1"), str(tmp_path / "g2"), str(tmp_path / "c")
    _runner(W, lambda: W.WebGPURenderer(0), gpu1, fmt="raw").run(5, rank=0, world=1, job_batch=2)
    for rank in range(2):
        _runner(W, lambda: W.WebGPURenderer(0), gpu2, fmt="raw").run(5, rank=rank, world=2, job_batch=2)
    _runner(W, oracle_lib.OracleRenderer, cpu, fmt="raw").run(5, rank=0, world=1, job_batch=2)
    a, b, c = _frames_on_disk(gpu1), _frames_on_disk(gpu2), _frames_on_disk(cpu)
    assert len(a) == 5 and a == b, "a frame depends on the rank that rendered it"
    assert a == c, "HIP frames differ from the oracle's"
    assert all(len(v) == 48 * 32 * 4 for v in a.values())
