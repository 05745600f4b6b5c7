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
    assert b2.updates == [] and [c[1] for c in rec2.calls if c[0] == "compute"] == [1, 2, 3, 4]
