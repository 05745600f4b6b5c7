"""The scene compiler's side of the device-resident update(t) (ms_world_set_device_updater, include/mi355scene.h): what
it hands the updater — the static description and the frame's joint matrices (rt_world_frame, mi355rt_layout.h) — is
enough to reproduce the host update: a numpy restatement of rebuilder.rs:36-91 fed ONLY from the frame gives the
vertices and normals the host path computes.  No GPU."""
import ctypes

import numpy as np

import test_gltf

f32 = np.float32


class Geometry(ctypes.Structure):
    _fields_ = [("positions", ctypes.POINTER(ctypes.c_float)), ("normals", ctypes.POINTER(ctypes.c_float)),
                ("uvs", ctypes.POINTER(ctypes.c_float)), ("joints", ctypes.POINTER(ctypes.c_uint32)),
                ("weights", ctypes.POINTER(ctypes.c_float)), ("indices", ctypes.POINTER(ctypes.c_uint32)),
                ("attributes", ctypes.POINTER(ctypes.c_float)), ("n_verts", ctypes.c_uint32), ("n_uvs", ctypes.c_uint32),
                ("n_tris", ctypes.c_uint32), ("skin", ctypes.c_int32)]


class Frame(ctypes.Structure):
    _fields_ = [("static_epoch", ctypes.c_uint64), ("n_geometries", ctypes.c_uint32), ("n_instances", ctypes.c_uint32),
                ("n_skins", ctypes.c_uint32), ("pad", ctypes.c_uint32), ("geometries", ctypes.POINTER(Geometry)),
                ("instances", ctypes.POINTER(ctypes.c_float)), ("skin_first", ctypes.POINTER(ctypes.c_uint32)),
                ("joint_mats", ctypes.POINTER(ctypes.c_float))]


UPDATER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(Frame))


def _arr(p, n, dtype):
    return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype).copy() if n else np.zeros(0, dtype)


def _snapshot(fr):
    """Copy everything the frame points at (the pointers only live during the call)."""
    f = fr.contents
    geos = []
    for g in range(f.n_geometries):
        d = f.geometries[g]
        geos.append(dict(pos=_arr(d.positions, d.n_verts * 3, f32).reshape(-1, 3), nrm=_arr(d.normals, d.n_verts * 3, f32).reshape(-1, 3),
                         uv=_arr(d.uvs, d.n_uvs * 2, f32).reshape(-1, 2), joints=_arr(d.joints, d.n_verts * 4, np.uint32).reshape(-1, 4),
                         weights=_arr(d.weights, d.n_verts * 4, f32).reshape(-1, 4), idx=_arr(d.indices, d.n_tris * 3, np.uint32),
                         attr=_arr(d.attributes, d.n_tris * 16, f32).reshape(-1, 16), skin=d.skin))
    first = _arr(f.skin_first, f.n_skins + 1, np.uint32)
    mats = _arr(f.joint_mats, int(first[-1]) * 16 if f.n_skins else 0, f32).reshape(-1, 4, 4)   # [joint][col][row]
    inst = _arr(f.instances, f.n_instances * 36, f32).reshape(-1, 36)
    return dict(epoch=f.static_epoch, geos=geos, skin_first=first, mats=mats, inst=inst)


def _skin(geo, first, mats):
    """rebuilder.rs:36-91 in numpy f32, one operation at a time (no fused multiply-add)."""
    pos, nrm = geo["pos"].copy(), geo["nrm"].copy()
    if geo["skin"] < 0:
        return pos, nrm
    jm = mats[first[geo["skin"]]:first[geo["skin"] + 1]]
    for i in range(len(pos)):
        m = np.zeros((4, 4), f32)
        for k in range(4):
            w, j = geo["weights"][i, k], geo["joints"][i, k]
            if w > 0 and j < len(jm):
                m = (m + jm[j] * w).astype(f32)
        if not m.any():
            m = np.eye(4, dtype=f32)
        p, n = pos[i].copy(), nrm[i].copy()
        q, v = np.zeros(3, f32), np.zeros(3, f32)
        for r in range(3):
            acc = f32(m[0, r] * p[0])
            acc = f32(f32(m[1, r] * p[1]) + acc)
            acc = f32(f32(m[2, r] * p[2]) + acc)
            q[r] = f32(m[3, r] + acc)
            bcc = f32(m[0, r] * n[0])
            bcc = f32(f32(m[1, r] * n[1]) + bcc)
            v[r] = f32(f32(m[2, r] * n[2]) + bcc)
        ln = np.sqrt(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2])), dtype=f32)
        with np.errstate(divide="ignore"):
            rcp = f32(1) / ln
        pos[i] = q
        nrm[i] = (v * rcp).astype(f32) if np.isfinite(rcp) and rcp > 0 else 0
    return pos, nrm


def test_frame_describes_the_update(W):
    glb = test_gltf.build_skinned(W)[0].glb()
    seen = []

    def updater(user, fr):
        seen.append(_snapshot(fr))
        return -1                       # "not taken": the host path runs, so the arrays to compare with exist

    cb = UPDATER(updater)
    b = W.WorldBridge()
    b.setDeviceUpdater(None, fn=cb)
    b.loadScene("viewer", glbData=glb)
    for t in (0.0, 0.3, 0.9):
        seen.clear()
        b.update(t)
        assert not b.deviceResident and "device updater failed" in b.deviceWarning
        assert len(seen) == 1
        fr = seen[0]
        verts = np.asarray(b.vertices).reshape(-1, 4)
        norms = np.asarray(b.normals).reshape(-1, 4)
        topo = np.asarray(b.mesh_topology).reshape(-1, 20)
        v0 = 0
        t0 = 0
        assert any(g["skin"] >= 0 for g in fr["geos"])
        for gi, g in enumerate(fr["geos"]):
            p, n = _skin(g, fr["skin_first"], fr["mats"])
            nv = len(p)
            assert np.array_equal(verts[v0:v0 + nv, :3].view(np.uint32), p.view(np.uint32)), (t, gi)
            assert np.array_equal(norms[v0:v0 + nv, :3].view(np.uint32), n.view(np.uint32)), (t, gi)
            rows = topo[t0:t0 + len(g["attr"])]
            assert (rows[:, 3] == gi).all()
            # the rows are the geometry's triangles in BLAS order: the same multiset of (indices + offset, attributes)
            want = np.concatenate([g["idx"].reshape(-1, 3) + v0, g["attr"].view(np.uint32)], 1)
            got = np.concatenate([rows[:, :3], rows[:, 4:]], 1)
            assert sorted(map(bytes, want)) == sorted(map(bytes, got)), (t, gi)
            v0 += nv
            t0 += len(g["attr"])
        assert v0 == len(verts) and t0 == len(topo)
        inst = np.asarray(b.instances).reshape(-1, 36)
        assert len(fr["inst"]) == len(inst)
        # same instances (transform, inverse, geometry) up to the TLAS order and the BLAS offsets filled in later
        key = lambda a: sorted(bytes(row[:32].tobytes() + row[33:].tobytes()) for row in a)
        assert key(fr["inst"]) == key(inst)
    epochs = {s["epoch"] for s in seen}
    b2 = W.WorldBridge()
    b2.setDeviceUpdater(None, fn=cb)
    b2.loadScene("cornell")
    b2.update(0.0)
    assert seen[-1]["epoch"] not in epochs          # another world: another static description


def test_a_taken_update_leaves_the_host_arrays_alone(W):
    calls = []

    def updater(user, fr):
        calls.append(fr.contents.n_geometries)
        return 0

    cb = UPDATER(updater)
    b = W.WorldBridge()
    b.loadScene("cornell")
    before = {k: np.asarray(getattr(b, k)).copy() for k in test_gltf.BRIDGE_ARRAYS}
    b.setDeviceUpdater(None, fn=cb)
    b.update(0.5)
    assert calls and b.deviceResident
    for k, v in before.items():
        assert np.array_equal(np.asarray(getattr(b, k)), v), k
    b.setDeviceUpdater(None)
    b.update(0.5)
    assert not b.deviceResident
