"""Device-resident World::update(t) (rt_world_update, SURVEY.md §8f N1 — VERDICT r02 item 5): every bridge array the
GPU derives inside the renderer's buffers is byte-identical to the scene compiler's host update (the restatement of
rust-shader-tools/src/lib.rs:149-270, rebuilder.rs:9-190, bvh/blas.rs, bvh/tlas.rs:58-111), and a frame traced from the
device-resident world is the frame traced from the uploaded arrays."""
import numpy as np
import pytest

import test_gltf

BRIDGE_ARRAYS = test_gltf.BRIDGE_ARRAYS


def _same(r, cpu_b, tag):
    for k in BRIDGE_ARRAYS:
        want = np.asarray(getattr(cpu_b, k)).view(np.uint32).reshape(-1)
        got = r.worldRead(k).view(np.uint32).reshape(-1)
        assert want.shape == got.shape, (tag, k, want.shape, got.shape)
        if not np.array_equal(want, got):
            bad = np.nonzero(want != got)[0]
            raise AssertionError("%s %s: %d of %d words differ, first at %d (want %08x got %08x)"
                                 % (tag, k, len(bad), len(want), bad[0], want[bad[0]], got[bad[0]]))


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cornell", "mixed", "special", "instanced1000", "instanced16384", "sponza_like", "glass_blob",
                                   "skinned_tube", "skinned_small", "character_in_hall"])
def test_device_update_equals_the_host_update(W, scene):
    """Every bridge array, byte for byte: the static scenes (1 to 16 384 instances - the most the device TLAS takes, a lattice
    full of equal centres: the stable sort, rotation and light lists), the two 200 k+ triangle meshes (large-node levels of the builder) and skinned, animated glTFs at three times."""
    r = W.WebGPURenderer(0)
    glb, name, times = None, scene, (0.0,)
    if scene == "skinned_tube":
        glb, name, times = test_gltf.big_skinned_glb(W)[0], "viewer", (0.0, 0.4, 1.7)
    elif scene == "character_in_hall":       # a large static mesh (kept between frames) + a small skinned one
        glb, name, times = test_gltf.character_in_hall_glb(W, (160, 96))[0], "viewer", (0.0, 0.4, 1.7)
    elif scene == "skinned_small":
        glb, name, times = test_gltf.build_skinned(W)[0].glb(), "viewer", (0.0, 0.3, 0.9)
    cpu_b, dev_b = W.WorldBridge(), W.WorldBridge()
    dev_b.setDeviceUpdater(r)
    cpu_b.loadScene(name, glbData=glb)
    dev_b.loadScene(name, glbData=glb)
    for t in times:
        cpu_b.update(t)
        dev_b.update(t)
        assert dev_b.deviceResident, dev_b.deviceWarning
        _same(r, cpu_b, "%s t=%g" % (scene, t))
    # a second pass over the same times: the builds now launch the level counts learnt from the first
    for t in times[::-1]:
        cpu_b.update(t)
        dev_b.update(t)
        assert dev_b.deviceResident, dev_b.deviceWarning
        _same(r, cpu_b, "%s again t=%g" % (scene, t))
    r.destroy()


def nasty_skinned_glb(seed, n_verts=700, n_tris=1500):
    """A triangle soup with two skinned primitives (two geometries, one skin): joint indices beyond the skin's two joints,
    all-zero weight rows (-> identity), weights that do not sum to one, zero-length normals, a scale animation that goes
    through zero and up to 1e20 (degenerate boxes, huge extents), duplicate vertices and zero-area triangles."""
    import gltf_util as G
    f32 = np.float32
    rng = np.random.default_rng(seed)
    b = G.GltfBuilder()
    prims = []
    for k in range(2):
        pos = (rng.random((n_verts, 3), dtype=f32) * f32(2) - f32(1)) * f32(0.4 + 0.3 * k)
        pos[::17] = pos[0]                                   # duplicates
        nrm = rng.standard_normal((n_verts, 3)).astype(f32)
        nrm[::13] = 0                                        # normalize_or_zero
        idx = rng.integers(0, n_verts, n_tris * 3).astype(np.uint32)
        idx[:30] = idx[0]                                    # zero-area triangles
        joints = rng.integers(0, 4, (n_verts, 4)).astype(np.uint16)       # 2 and 3 are beyond the skin
        weights = rng.random((n_verts, 4), dtype=f32)
        weights[rng.random(n_verts) < 0.2] = 0               # no influence at all
        weights[rng.random((n_verts, 4)) < 0.3] = 0
        acc = dict(POSITION=b.accessor(pos, G.F32, "VEC3", minmax=True), NORMAL=b.accessor(nrm, G.F32, "VEC3"),
                   JOINTS_0=b.accessor(joints, G.U16, "VEC4"), WEIGHTS_0=b.accessor(weights, G.F32, "VEC4"))
        prims.append({"attributes": acc, "indices": b.accessor(idx, G.U32, "SCALAR")})
    b.doc["meshes"] = [{"primitives": prims}]
    ibm = np.stack([np.eye(4, dtype=f32), np.eye(4, dtype=f32)])
    ibm[1][1, 3] = -0.5
    b.doc["nodes"] = [{"mesh": 0, "skin": 0}, {"children": [2]}, {"translation": [0, 0.5, 0]}]
    b.doc["skins"] = [{"joints": [1, 2], "inverseBindMatrices": b.accessor(np.stack([m.T for m in ibm]), G.F32, "MAT4")}]
    s, c = np.sin(0.6), np.cos(0.6)
    t_in = b.accessor(np.array([0, 1, 2, 3], f32), G.F32, "SCALAR", minmax=True)
    rot = b.accessor(np.array([[0, 0, -s, c], [s, 0, 0, c], [0, s, 0, c], [0, 0, -s, c]], f32), G.F32, "VEC4")
    sc = b.accessor(np.array([[1, 1, 1], [0, 0, 0], [1e20, 1, 1e-20], [1, 1, 1]], f32), G.F32, "VEC3")
    b.doc["animations"] = [{"samplers": [{"input": t_in, "output": rot}, {"input": t_in, "output": sc}],
                            "channels": [{"sampler": 0, "target": {"node": 2, "path": "rotation"}},
                                         {"sampler": 1, "target": {"node": 1, "path": "scale"}}]}]
    return b.glb()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_update_on_degenerate_skinned_input(W, seed):
    """The corner cases of the skinning loop and of the builders (see nasty_skinned_glb), at times that put the scale
    animation at 1, near 0, at exactly 0, on the way to 1e20 and at 1e20: the device either produces the host's bytes or
    refuses the frame (a NaN instance box) and the host path takes it."""
    r = W.WebGPURenderer(0)
    glb = nasty_skinned_glb(seed)
    cpu_b, dev_b = W.WorldBridge(), W.WorldBridge()
    dev_b.setDeviceUpdater(r)
    cpu_b.loadScene("viewer", glbData=glb)
    dev_b.loadScene("viewer", glbData=glb)
    taken = 0
    for t in (0.0, 0.37, 0.98, 1.0, 1.5, 2.0, 2.6, 0.2):
        cpu_b.update(t)
        dev_b.update(t)
        if dev_b.deviceResident:
            taken += 1
            _same(r, cpu_b, "seed %d t=%g" % (seed, t))
        else:
            assert "NaN" in dev_b.deviceWarning, dev_b.deviceWarning
            for k in BRIDGE_ARRAYS:
                assert np.array_equal(np.asarray(getattr(cpu_b, k)).view(np.uint32), np.asarray(getattr(dev_b, k)).view(np.uint32)), (t, k)
    assert taken >= 4
    r.destroy()


@pytest.mark.gpu
def test_frames_from_the_device_world_equal_frames_from_uploaded_arrays(W):
    """The live loop with the device updater against the live loop that uploads the host arrays: accumulation buffers
    and counters bit for bit, over frames in which the world advances."""
    glb = test_gltf.big_skinned_glb(W, 48, 24)[0]
    out = []
    for device in (False, True):
        r = W.WebGPURenderer(0)
        r.buildPipeline(4, 1)
        b = W.WorldBridge(zero_copy=True)
        if device:
            b.setDeviceUpdater(r)
        b.loadScene("viewer", glbData=glb)
        W.upload_scene(r, b, 256, 144)
        loop = W.LiveLoop(r, b, 256, 144, update_interval=2, lookahead=0)
        imgs = []
        for _ in range(7):
            loop.render_frame()
            imgs.append(r.readAccum().copy())
        if device:
            assert b.deviceResident, b.deviceWarning
        out.append((imgs, r.getCounters()))
        r.destroy()
    for k, (a, d) in enumerate(zip(out[0][0], out[1][0])):
        assert np.array_equal(a.view(np.uint32), d.view(np.uint32)), k
    assert list(out[0][1]) == list(out[1][1])


@pytest.mark.gpu
def test_descriptions_the_device_path_refuses(W):
    """rt_world_update called directly with hand-made frames: what it does not take comes back as an error with the
    reason (the scene compiler then runs the host path), and a frame it takes is right afterwards."""
    import ctypes
    from test_world_device_hook import Frame, Geometry
    r = W.WebGPURenderer(0)
    f32 = np.float32
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], f32)
    nrm = np.tile(np.array([0, 0, 1], f32), (4, 1))
    joints = np.zeros((4, 4), np.uint32)
    weights = np.zeros((4, 4), f32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    attr = np.zeros((2, 16), f32)
    attr[:, :3] = 0.5
    fp, up = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)

    def geometry(n_tris=2):
        g = Geometry()
        g.positions, g.normals, g.uvs = pos.ctypes.data_as(fp), nrm.ctypes.data_as(fp), None
        g.joints, g.weights = joints.ctypes.data_as(up), weights.ctypes.data_as(fp)
        g.indices, g.attributes = idx.ctypes.data_as(up), attr.ctypes.data_as(fp)
        g.n_verts, g.n_uvs, g.n_tris, g.skin = 4, 0, n_tris, -1
        return g

    def frame(epoch, geos, inst):
        fr = Frame()
        arr = (Geometry * len(geos))(*geos)
        fr.static_epoch, fr.n_geometries, fr.n_instances, fr.n_skins = epoch, len(geos), len(inst), 0
        fr.geometries, fr.instances = arr, inst.ctypes.data_as(fp)
        return fr, arr

    def instance(geometry_id, m=None):
        row = np.zeros(36, f32)
        m = np.eye(4, dtype=f32) if m is None else m
        row[:16] = m.T.reshape(-1)
        row[16:32] = np.linalg.inv(np.nan_to_num(m)).T.reshape(-1).astype(f32)
        row[32:].view(np.uint32)[:] = (0, 0, geometry_id, 0)
        return row

    def call(fr):
        return r.L.rt_world_update(r.ctx, ctypes.byref(fr[0])), r.L.rt_last_error(r.ctx).decode()

    rc, why = call(frame(1001, [geometry()], np.stack([instance(0)])))
    assert rc >= 0, why
    nodes = r.worldRead("blas").reshape(-1, 8)
    assert len(nodes) == 1 and nodes[0, 7:8].view(np.uint32)[0] == 2          # one leaf: first 0, count 2
    bad = np.eye(4, dtype=f32)
    bad[0, 3] = np.nan
    rc, why = call(frame(1002, [geometry()], np.stack([instance(0, bad)])))
    assert rc < 0 and "NaN" in why
    rc, why = call(frame(1003, [geometry(), geometry(0)], np.stack([instance(0), instance(1)])))
    assert rc < 0 and "no triangles" in why
    rc, why = call(frame(1004, [geometry()], np.stack([instance(3)])))
    assert rc < 0 and "missing or empty" in why
    rc, why = call(frame(1005, [geometry()], np.stack([instance(0)] * 16385)))
    assert rc < 0 and "16 384" in why
    idx[5] = 9                                                               # vertex index out of range
    rc, why = call(frame(1006, [geometry()], np.stack([instance(0)])))
    assert rc < 0 and "out of range" in why
    idx[5] = 3
    rc, why = call(frame(1007, [geometry()], np.stack([instance(0), instance(0)])))
    assert rc >= 0, why
    assert len(r.worldRead("tlas")) // 8 == 3 and len(r.worldRead("instances")) // 36 == 2
    # ---- the per-call skin arguments are checked against the static description on EVERY call (round 4)
    joints[:, 0] = 0
    weights[:, 0] = 1.0
    mats = np.tile(np.eye(4, dtype=f32).reshape(-1), 2)                       # two joints, identity

    def skinned_frame(epoch, n_skins, first):
        g = geometry()
        g.skin = 0
        fr, arr = frame(epoch, [g], np.stack([instance(0)]))
        fr.n_skins = n_skins
        keep = np.array(first, np.uint32) if first is not None else None
        fr.skin_first = keep.ctypes.data_as(up) if keep is not None else None
        fr.joint_mats = mats.ctypes.data_as(fp)
        return fr, arr, keep

    rc, why = call(skinned_frame(2001, 1, [0, 2]))
    assert rc >= 0, why
    rc, why = call(skinned_frame(2001, 0, None))                              # fewer skins under the same static_epoch
    assert rc < 0 and "number of skins changed" in why
    rc, why = call(skinned_frame(2001, 1, None))
    assert rc < 0 and "null skin table" in why
    rc, why = call(skinned_frame(2001, 1, [1, 2]))
    assert rc < 0 and "does not start at joint 0" in why
    rc, why = call(skinned_frame(2002, 2, [0, 2, 1]))
    assert rc < 0 and "not non-decreasing" in why
    rc, why = call(skinned_frame(2003, 1, [0, 2]))                            # and a good frame is taken afterwards
    assert rc >= 0, why
    r.destroy()


@pytest.mark.gpu
def test_static_geometry_cache(W):
    """Geometries without a skin keep their BLAS / rows between frames (rt_world_set_static_cache, default on): the arrays
    are those of a full rebuild; a host upload into the renderer in between (another scene!) drops the cache; with the
    cache off every frame rebuilds everything."""
    r = W.WebGPURenderer(0)
    glb = test_gltf.big_skinned_glb(W, 64, 32)[0]
    cpu_b, dev_b = W.WorldBridge(), W.WorldBridge()
    dev_b.setDeviceUpdater(r)
    cpu_b.loadScene("viewer", glbData=glb)
    dev_b.loadScene("viewer", glbData=glb)
    other = W.WorldBridge()
    other.loadScene("mixed")
    for k, t in enumerate((0.0, 0.5, 0.9, 1.3, 1.8, 0.1)):
        if k == 3:
            W.upload_scene(r, other, 64, 48)            # somebody else's arrays land in the scene buffers
        if k == 4:
            r.setWorldStaticCache(False)
        cpu_b.update(t)
        dev_b.update(t)
        assert dev_b.deviceResident, dev_b.deviceWarning
        _same(r, cpu_b, "cache step %d" % k)
    r.setWorldStaticCache(True)
    for scene in ("sponza_like", "instanced1000"):      # all-static worlds: the second update only redoes TLAS and packing
        cpu_b.loadScene(scene)
        dev_b.loadScene(scene)
        for t in (0.0, 0.3, 0.6):
            cpu_b.update(t)
            dev_b.update(t)
            assert dev_b.deviceResident, dev_b.deviceWarning
            _same(r, cpu_b, "%s cached t=%g" % (scene, t))
    r.destroy()


@pytest.mark.gpu
def test_animated_live_loop_traces_nothing_in_vain(W):
    """A world that moves every 4th frame, lookahead on: the loop tells the library when the run ends
    (rt_set_lookahead_limit), so the images are those of one dispatch per frame AND exactly the displayed frames were
    traced (with lookahead the ray counters count a frame when it is traced)."""
    glb = test_gltf.big_skinned_glb(W, 48, 24)[0]
    out = []
    for look in (0, 32):
        r = W.WebGPURenderer(0)
        r.buildPipeline(4, 1)
        b = W.WorldBridge(zero_copy=True)
        b.setDeviceUpdater(r)
        b.loadScene("viewer", glbData=glb)
        W.upload_scene(r, b, 256, 144)
        loop = W.LiveLoop(r, b, 256, 144, update_interval=4, lookahead=look)
        r.resetCounters()
        imgs = []
        for _ in range(16):                      # four whole runs of four frames
            loop.render_frame()
            imgs.append(r.readAccum().copy())
        out.append((imgs, r.getCounters()))
        r.destroy()
    for k, (a, d) in enumerate(zip(out[0][0], out[1][0])):
        assert np.array_equal(a.view(np.uint32), d.view(np.uint32)), k
    assert out[0][1]["primary_rays"] == 16 * 256 * 144
    assert dict(out[0][1]) == dict(out[1][1]), (out[0][1], out[1][1])


@pytest.mark.gpu
def test_device_updater_refuses_a_destroyed_renderer(W):
    r = W.WebGPURenderer(0)
    b = W.WorldBridge()
    b.setDeviceUpdater(r)
    b.loadScene("cornell")
    b.update(0.0)
    assert b.deviceResident, b.deviceWarning
    r.destroy()
    with pytest.raises(RuntimeError):
        b.update(0.1)                # the renderer behind the updater is gone
    b.setDeviceUpdater(None)
    b.update(0.1)
    assert not b.deviceResident


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["axis_rule", "stable_ties", "costlier_half_first", "translated"])
def test_device_tlas_equals_the_hand_derived_arrays(W, case):
    """k_tlas (csrc/world_update.hip.h) against the TLAS node arrays and instance orders worked out by hand from
    bvh/tlas.rs:58-111 (tests/golden/tlas_kat.json; derivations in tests/test_bvh_independent.py, where the CPU builder is
    held to the same arrays): the axis rule that is not "longest axis", the STABLE sort on equal centres, the costlier half
    first, boxes of transformed instances.  rt_world_update is called directly with a hand-made frame: one geometry per
    instance, a single triangle (x0,y0,z0) (x1,y1,z0) (x0,y0,z1) whose box is the instance's box."""
    import ctypes
    import json
    import os
    from test_world_device_hook import Frame, Geometry
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tlas_kat.json")))[case]
    f32 = np.float32
    fp, up = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)
    keep, geos, rows = [], [], []
    for gi, b in enumerate(kat["boxes"]):
        x0, y0, z0, x1, y1, z1 = b
        pos = np.array([[x0, y0, z0], [x1, y1, z0], [x0, y0, z1]], f32)
        nrm = np.tile(np.array([0, 0, 1], f32), (3, 1))
        joints, weights = np.zeros((3, 4), np.uint32), np.zeros((3, 4), f32)
        idx, attr = np.array([0, 1, 2], np.uint32), np.full((1, 16), 0.5, f32)
        keep += [pos, nrm, joints, weights, idx, attr]
        g = Geometry()
        g.positions, g.normals, g.uvs = pos.ctypes.data_as(fp), nrm.ctypes.data_as(fp), None
        g.joints, g.weights = joints.ctypes.data_as(up), weights.ctypes.data_as(fp)
        g.indices, g.attributes = idx.ctypes.data_as(up), attr.ctypes.data_as(fp)
        g.n_verts, g.n_uvs, g.n_tris, g.skin = 3, 0, 1, -1
        geos.append(g)
        m = np.eye(4, dtype=f32)
        if "translations" in kat:
            m[:3, 3] = kat["translations"][gi]
        row = np.zeros(36, f32)
        row[:16] = m.T.reshape(-1)
        row[16:32] = np.linalg.inv(m).T.reshape(-1).astype(f32)
        row[32:].view(np.uint32)[:] = (0, 0, gi, 0)
        rows.append(row)
    inst = np.stack(rows)
    arr = (Geometry * len(geos))(*geos)
    fr = Frame()
    fr.static_epoch, fr.n_geometries, fr.n_instances, fr.n_skins = 7001, len(geos), len(inst), 0
    fr.geometries, fr.instances = arr, inst.ctypes.data_as(fp)
    r = W.WebGPURenderer(0)
    rc = r.L.rt_world_update(r.ctx, ctypes.byref(fr))
    assert rc >= 0, r.L.rt_last_error(r.ctx).decode()
    tlas = r.worldRead("tlas").reshape(-1, 8)
    u = tlas.view(np.uint32)
    got = [{"min": tlas[i, 0:3].tolist(), "skip": int(u[i, 3]), "max": tlas[i, 4:7].tolist(), "data": int(u[i, 7])} for i in range(len(tlas))]
    assert got == kat["nodes"]
    out_inst = r.worldRead("instances").reshape(-1, 36)
    assert [int(v) for v in out_inst[:, 34].view(np.uint32)] == kat["order"]      # instance_id = the geometry = the input index
    r.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("n,seed", [(2, 1), (3, 2), (63, 3), (64, 4), (65, 5), (129, 10), (1000, 6), (1024, 7), (1025, 8), (4097, 9)])
def test_device_tlas_on_random_and_tied_instance_sets(W, monkeypatch, n, seed, generic):
    """k_tlas / k_tlas_small against the CPU builder (ms_build_tlas) on instance sets built to be awkward for the sort: random boxes, many
    exactly equal centres (ties decided by the current order, depth after depth), centres at -0 / +0, counts around the
    wave, workgroup and power-of-two sizes of the bitonic network.  One single-triangle geometry per instance, translated."""
    import ctypes
    from test_world_device_hook import Frame, Geometry
    from test_bvh_independent import cpu_build_tlas
    if generic:                 # up to 1 024 instances take k_tlas_small (one position per lane); this forces k_tlas for them too
        if n > 1024:
            pytest.skip("k_tlas is what runs above 1 024 instances anyway")
        monkeypatch.setenv("MI355RT_TLAS_GENERIC", "1")
    f32 = np.float32
    rng = np.random.default_rng(seed)
    lo = rng.integers(-8, 8, (n, 3)).astype(f32) * f32(0.25)
    if seed % 2 == 0:
        lo[:, rng.integers(0, 3)] = 0                  # every centre equal on one axis
    lo[rng.random(n) < 0.3] = lo[0]                     # 30 % of the instances share one box
    ext = rng.integers(1, 4, (n, 3)).astype(f32) * f32(0.5)
    if n >= 3:
        lo[1], ext[1] = (-0.5, 0.25, 0.0), (1.0, 0.5, 1.0)      # centre x = +0
        lo[2], ext[2] = (0.5, 0.25, 0.0), (-1.0, 0.5, 1.0)      # box given max-first: min/max swap, centre x = (0.5 + -0.5) / 2 = 0
    boxes = np.concatenate([np.minimum(lo, lo + ext), np.maximum(lo, lo + ext)], axis=1)
    want_nodes, want_order = cpu_build_tlas(W, boxes.tolist())
    fp, up = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)
    keep, geos, rows = [], [], []
    nrm = np.tile(np.array([0, 0, 1], f32), (3, 1))
    joints, weights = np.zeros((3, 4), np.uint32), np.zeros((3, 4), f32)
    idx, attr = np.array([0, 1, 2], np.uint32), np.full((1, 16), 0.5, f32)
    for gi in range(n):
        x0, y0, z0, x1, y1, z1 = boxes[gi]
        pos = np.array([[x0, y0, z0], [x1, y1, z0], [x0, y0, z1]], f32)
        keep.append(pos)
        g = Geometry()
        g.positions, g.normals, g.uvs = pos.ctypes.data_as(fp), nrm.ctypes.data_as(fp), None
        g.joints, g.weights = joints.ctypes.data_as(up), weights.ctypes.data_as(fp)
        g.indices, g.attributes = idx.ctypes.data_as(up), attr.ctypes.data_as(fp)
        g.n_verts, g.n_uvs, g.n_tris, g.skin = 3, 0, 1, -1
        geos.append(g)
        row = np.zeros(36, f32)
        row[:16] = np.eye(4, dtype=f32).reshape(-1)
        row[16:32] = np.eye(4, dtype=f32).reshape(-1)
        row[32:].view(np.uint32)[:] = (0, 0, gi, 0)
        rows.append(row)
    inst = np.stack(rows)
    arr = (Geometry * n)(*geos)
    fr = Frame()
    fr.static_epoch, fr.n_geometries, fr.n_instances, fr.n_skins = 8000 + seed, n, n, 0
    fr.geometries, fr.instances = arr, inst.ctypes.data_as(fp)
    r = W.WebGPURenderer(0)
    rc = r.L.rt_world_update(r.ctx, ctypes.byref(fr))
    assert rc >= 0, r.L.rt_last_error(r.ctx).decode()
    tlas = r.worldRead("tlas").reshape(-1, 8)
    u = tlas.view(np.uint32)
    got = [{"min": tlas[i, 0:3].tolist(), "skip": int(u[i, 3]), "max": tlas[i, 4:7].tolist(), "data": int(u[i, 7])} for i in range(len(tlas))]
    assert [int(v) for v in r.worldRead("instances").reshape(-1, 36)[:, 34].view(np.uint32)] == want_order
    assert got == want_nodes
    r.destroy()
