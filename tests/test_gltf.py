"""glTF / GLB input of the scene compiler (reference: rust-shader-tools/src/loader.rs, lib.rs:45-270,
rebuilder.rs:36-91).  The reference has no asset and no test for this path and its parser is an un-vendored crate, so the
checker is the numpy restatement in tests/gltf_util.py (parity unpinned) — plus GPU = oracle on the resulting arrays."""
import io
import os

import numpy as np
import pytest

import gltf_util as G

f32 = np.float32
ENV_VERTS = 24      # the viewer room: 6 quads (procedural.rs:634-791)


def quad_mesh():
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], f32)
    nrm = np.tile(np.array([0, 0, 1], f32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], f32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    return pos, nrm, uv, idx


def png_blob(W, h=8, w=16, seed=0):
    rng = np.random.default_rng(seed)
    return W.textures.encode_png(rng.integers(0, 256, (h, w, 4), dtype=np.uint8))


def build_static(W):
    b = G.GltfBuilder()
    pos, nrm, uv, idx = quad_mesh()
    inter = np.concatenate([pos, nrm], axis=1).astype(f32)            # interleaved POSITION | NORMAL, stride 24
    v = b.view(inter.tobytes(), stride=24)
    a_pos = b.accessor(pos, G.F32, "VEC3", view=v, offset=0, count=4, minmax=True)
    a_nrm = b.accessor(nrm, G.F32, "VEC3", view=v, offset=12, count=4)
    a_uv = b.accessor((uv * 65535).astype(np.uint16), G.U16, "VEC2", normalized=True)
    a_idx = b.accessor(idx, G.U16, "SCALAR")
    tri = np.array([[0, 0, 1], [0.5, 0, 1], [0, 0.5, 1]], f32)
    a_tri = b.accessor(tri, G.F32, "VEC3", minmax=True)
    lpos = pos * f32(0.25) + f32(0.5)
    a_lpos = b.accessor(lpos, G.F32, "VEC3", minmax=True)
    a_lidx = b.accessor(idx.astype(np.uint8), G.U8, "SCALAR")
    t0 = b.image_texture(png_blob(W))
    t1 = b.external_texture()
    b.doc["materials"] = [
        {"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.5,
                                  "baseColorTexture": {"index": t0}}, "normalTexture": {"index": t1}},
        {"pbrMetallicRoughness": {"metallicFactor": 0.0}, "emissiveFactor": [5.0, 5.0, 5.0], "occlusionTexture": {"index": t0}},
    ]
    b.doc["meshes"] = [{"primitives": [
        {"attributes": {"POSITION": a_pos, "NORMAL": a_nrm, "TEXCOORD_0": a_uv}, "indices": a_idx, "material": 0},
        {"attributes": {"POSITION": a_tri}},                                                   # no indices, normals, uv, material
        {"attributes": {"POSITION": a_lpos}, "indices": a_lidx, "material": 1, "mode": 4},
        {"attributes": {"POSITION": a_tri}, "mode": 1},                                        # LINES: not consumed
    ]}]
    b.doc["nodes"] = [{"name": "root", "translation": [0.1, 0.2, 0.3], "children": [1]},
                      {"name": "model", "mesh": 0, "rotation": [0, 0.258819, 0, 0.9659258], "scale": [1, 2, 1]}]
    b.doc["scenes"][0]["nodes"] = [0]
    return b, dict(pos=pos, nrm=nrm, uv=uv, tri=tri, lpos=lpos)


def rows_of_geometry(bridge, gi):
    topo = np.asarray(bridge.mesh_topology).reshape(-1, 20)
    return topo[topo[:, 3] == gi]


def test_static_glb_geometry_materials_textures(W):
    b, ref = build_static(W)
    br = W.WorldBridge()
    br.loadScene("viewer", glbData=b.glb())
    assert br.loadWarning == ""
    assert br.nodeCount == 2 and br.getAnimationList() == []
    v = np.asarray(br.vertices).reshape(-1, 4)
    n = np.asarray(br.normals).reshape(-1, 4)
    uv = np.asarray(br.uvs).reshape(-1, 2)
    assert len(v) == ENV_VERTS + 4 + 3 + 4        # no placeholder sphere when a GLB is given
    assert np.array_equal(v[ENV_VERTS:, :3], np.concatenate([ref["pos"], ref["tri"], ref["lpos"]])) and (v[:, 3] == 1).all()
    assert np.array_equal(n[ENV_VERTS:ENV_VERTS + 4, :3], ref["nrm"])
    assert (n[ENV_VERTS + 4:, :3] == np.array([0, 1, 0], f32)).all()          # default normal
    want_uv = ((ref["uv"] * 65535).astype(np.uint16).astype(f32) / f32(65535))
    assert np.array_equal(uv[ENV_VERTS:ENV_VERTS + 4], want_uv) and (uv[ENV_VERTS + 4:] == 0).all()
    # geometries: 0 env, 1 (empty) OBJ slot, 2.. one per consumed primitive
    r2, r3, r4 = (rows_of_geometry(br, g) for g in (2, 3, 4))
    assert len(r2) == 2 and len(r3) == 1 and len(r4) == 2 and len(rows_of_geometry(br, 5)) == 0
    a2 = r2[0, 4:].view(f32)
    assert np.allclose(a2[:3], [0.2, 0.4, 0.6]) and a2[3] == 0 and a2[4] == 0 and a2[5] == 0.5 and a2[6] == 1.5   # LAMBERTIAN
    assert list(a2[8:12]) == [0, -1, 1, -1] and list(a2[12:16]) == [0, 0, 0, -1]
    a3 = r3[0, 4:].view(f32)
    assert list(a3[:8]) == [1, 1, 1, 1, 1, 1, 1.5, 0] and list(a3[8:12]) == [-1] * 4          # default material -> METAL
    a4 = r4[0, 4:].view(f32)
    assert a4[3] == 3 and list(a4[12:16]) == [5, 5, 5, 0]                                       # emissive -> LIGHT, occlusion tex 0
    # indices are global vertex ids
    assert sorted(set(r2[:, :3].ravel())) == [ENV_VERTS + k for k in range(4)]
    assert sorted(r3[0, :3]) == [ENV_VERTS + 4, ENV_VERTS + 5, ENV_VERTS + 6]
    # instances: env + one per primitive; lib.rs:196-204 overwrites every instance after the first
    inst = np.asarray(br.instances).reshape(-1, 36)
    assert len(inst) == 4 and sorted(inst[:, 34].view(np.uint32)) == [0, 2, 3, 4]
    hack = np.array([[-0.7, 0, 0, 0], [0, 0.7, 0, 0], [0, 0, -0.7, 0], [0, 0, 0, 1]], f32)
    for row in inst:
        m = row[:16].reshape(4, 4).T
        if row[34:35].view(np.uint32)[0] == 0:
            assert np.array_equal(m, np.eye(4, dtype=f32))
        else:
            assert np.allclose(m, hack, atol=1e-6)
    # lights: the room's light quad + the emissive primitive
    lights = np.asarray(br.lights).reshape(-1, 2)
    assert len(lights) == 2 + 2
    # textures: encoded blobs in glTF texture order; an external image is an empty blob
    assert br.textureCount == 2
    assert br.getTexture(0) == png_blob(W) and br.getTexture(1) is None and br.getTextureRGBA(0) is None


def build_skinned(W, matrix_joint=False, weights_u8=False):
    b = G.GltfBuilder()
    ys = np.linspace(0, 1, 7, dtype=f32)
    pos = np.array([[x, y, 0] for y in ys for x in (-0.1, 0.1)], f32)
    nrm = np.tile(np.array([0, 0, 1], f32), (len(pos), 1))
    idx = []
    for k in range(len(ys) - 1):
        a = 2 * k
        idx += [a, a + 1, a + 3, a, a + 3, a + 2]
    w1 = np.clip((pos[:, 1] - 0.25) * 2, 0, 1).astype(f32)
    weights = np.stack([1 - w1, w1, np.zeros_like(w1), np.zeros_like(w1)], 1).astype(f32)
    joints = np.tile(np.array([0, 1, 0, 0], np.uint8), (len(pos), 1))
    a_pos = b.accessor(pos, G.F32, "VEC3", minmax=True)
    a_nrm = b.accessor(nrm, G.F32, "VEC3")
    a_idx = b.accessor(np.array(idx, np.uint32), G.U32, "SCALAR")
    a_j = b.accessor(joints, G.U8, "VEC4")
    if weights_u8:
        wq = np.round(weights * 255).astype(np.uint8)
        a_w = b.accessor(wq, G.U8, "VEC4", normalized=True)
        weights = wq.astype(f32) / f32(255)
    else:
        a_w = b.accessor(weights, G.F32, "VEC4")
    ibm = np.stack([np.eye(4, dtype=f32), np.eye(4, dtype=f32)])
    ibm[1][1, 3] = -0.5                                       # inverse of translate(0, 0.5, 0)
    a_ibm = b.accessor(np.stack([m.T for m in ibm]), G.F32, "MAT4")        # column-major
    b.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": a_pos, "NORMAL": a_nrm, "JOINTS_0": a_j, "WEIGHTS_0": a_w},
                                        "indices": a_idx}]}]
    j1 = {"name": "j1", "translation": [0, 0.5, 0]}
    if matrix_joint:
        j1 = {"name": "j1", "matrix": G.mat_from_srt([1, 1, 1], [0, 0, 0, 1], [0, 0.5, 0]).T.reshape(-1).tolist()}
    b.doc["nodes"] = [{"name": "body", "mesh": 0, "skin": 0}, {"name": "j0", "children": [2]}, j1]
    b.doc["skins"] = [{"joints": [1, 2], "inverseBindMatrices": a_ibm}]
    b.doc["scenes"][0]["nodes"] = [0, 1]
    # animation 0: rotation of j1 (LINEAR), translation of j0 (STEP), scale of j1 (CUBICSPLINE)
    c60, s60 = np.cos(np.pi / 6), np.sin(np.pi / 6)
    t_rot = b.accessor(np.array([0, 0.5, 1.0], f32), G.F32, "SCALAR", minmax=True)
    v_rot = b.accessor(np.array([[0, 0, 0, 1], [0, 0, s60, c60], [0, 0, 0, 1]], f32), G.F32, "VEC4")
    t_tr = b.accessor(np.array([0, 0.5], f32), G.F32, "SCALAR", minmax=True)
    v_tr = b.accessor(np.array([[0, 0, 0], [0.2, 0, 0]], f32), G.F32, "VEC3")
    t_sc = b.accessor(np.array([0, 1.0], f32), G.F32, "SCALAR", minmax=True)
    sc = np.array([[0, 0, 0], [1, 1, 1], [0, 0, 0], [0, 0, 0], [1.5, 1, 1], [0, 0, 0]], f32)     # (in, value, out) x 2 keys
    v_sc = b.accessor(sc, G.F32, "VEC3")
    b.doc["animations"] = [{"name": "bend", "samplers": [
        {"input": t_rot, "output": v_rot, "interpolation": "LINEAR"},
        {"input": t_tr, "output": v_tr, "interpolation": "STEP"},
        {"input": t_sc, "output": v_sc, "interpolation": "CUBICSPLINE"}],
        "channels": [{"sampler": 0, "target": {"node": 2, "path": "rotation"}},
                     {"sampler": 1, "target": {"node": 1, "path": "translation"}},
                     {"sampler": 2, "target": {"node": 2, "path": "scale"}}]},
        {"samplers": [{"input": t_tr, "output": v_tr}], "channels": [{"sampler": 0, "target": {"node": 1, "path": "translation"}}]}]
    anim = dict(rot=([0, 0.5, 1.0], [[0, 0, 0, 1], [0, 0, s60, c60], [0, 0, 0, 1]]), tr=([0, 0.5], [[0, 0, 0], [0.2, 0, 0]]),
                sc=([0, 1.0], sc), duration=1.0)
    return b, dict(pos=pos, nrm=nrm, joints=joints.astype(int), weights=weights, ibm=ibm, anim=anim)


def fresh_nodes():
    return [dict(t=[0, 0, 0], r=[0, 0, 0, 1], s=[1, 1, 1]), dict(t=[0, 0, 0], r=[0, 0, 0, 1], s=[1, 1, 1], children=[2]),
            dict(t=[0, 0.5, 0], r=[0, 0, 0, 1], s=[1, 1, 1])]


def expected_skinned(ref, t, nodes, anim_index=0):
    """`nodes` carries the local TRS from update to update: the reference animates the nodes in place and a channel that
    the active animation does not have keeps its last value (lib.rs:383-491)."""
    a = ref["anim"]
    if anim_index == 0:
        dur = a["duration"]
        tt = float(np.fmod(f32(t), f32(dur))) if dur > 0.001 else 0.0
        p, n, fac = G.sample_channel(a["rot"][0], a["rot"][1], "LINEAR", dur, tt)
        nodes[2]["r"] = G.quat_slerp(G.quat_normalize(p), G.quat_normalize(n), fac)
        p, n, fac = G.sample_channel(a["tr"][0], a["tr"][1], "STEP", dur, tt)
        nodes[1]["t"] = p + (n - p) * fac
        p, n, fac = G.sample_channel(a["sc"][0], a["sc"][1], "CUBICSPLINE", dur, tt)
        nodes[2]["s"] = p + (n - p) * fac
    else:
        dur = 0.5
        tt = float(np.fmod(f32(t), f32(dur)))
        p, n, fac = G.sample_channel(a["tr"][0], a["tr"][1], "LINEAR", dur, tt)
        nodes[1]["t"] = p + (n - p) * fac
    g = G.node_globals(nodes)
    jm = [(g[1] @ ref["ibm"][0]).astype(f32), (g[2] @ ref["ibm"][1]).astype(f32)]
    return G.skin_vertices(ref["pos"], ref["nrm"], ref["joints"], ref["weights"], jm)


@pytest.mark.parametrize("matrix_joint,weights_u8", [(False, False), (True, False), (False, True)])
def test_skinned_animated_glb_follows_the_restatement(W, matrix_joint, weights_u8):
    b, ref = build_skinned(W, matrix_joint, weights_u8)
    br = W.WorldBridge()
    br.loadScene("viewer", glbData=b.glb())
    assert br.loadWarning == "" and br.nodeCount == 3
    assert br.getAnimationList() == ["bend", "anim"]           # unnamed animations are called "anim" (loader.rs:347)
    nodes = fresh_nodes()
    expected_skinned(ref, 0.0, nodes)                          # World::new ends with update(0.0)
    for t in (0.0, 0.2, 0.5, 0.77, 1.3, 2.0):
        br.update(t)
        v = np.asarray(br.vertices).reshape(-1, 4)[ENV_VERTS:, :3]
        n = np.asarray(br.normals).reshape(-1, 4)[ENV_VERTS:, :3]
        wp, wn = expected_skinned(ref, t, nodes)
        assert np.allclose(v, wp, atol=2e-6), (t, np.abs(v - wp).max())
        assert np.allclose(n, wn, atol=2e-6)
    # the skinned instance sits at identity before the i > 0 overwrite; BLAS follows the deformation
    br.update(0.5)
    expected_skinned(ref, 0.5, nodes)
    blas = np.asarray(br.blas).reshape(-1, 8)
    inst = np.asarray(br.instances).reshape(-1, 36)
    model = [r for r in inst if r[34:35].view(np.uint32)[0] == 2][0]
    root = blas[int(model[32:33].view(np.uint32)[0])]
    v = np.asarray(br.vertices).reshape(-1, 4)[ENV_VERTS:, :3]
    assert np.allclose(root[:3], v.min(0), atol=1e-5) and np.allclose(root[4:7], v.max(0), atol=1e-5)   # (flat boxes are padded)
    # second animation (default LINEAR sampler), switched with setAnimation
    br.setAnimation(1)
    br.update(0.25)
    v = np.asarray(br.vertices).reshape(-1, 4)[ENV_VERTS:, :3]
    assert np.allclose(v, expected_skinned(ref, 0.25, nodes, anim_index=1)[0], atol=2e-6)   # j1 keeps its pose of t = 0.5
    # load_animation_glb appends the other file's animations (lib.rs:126-147)
    assert br.loadAnimation(b.glb()) == 2 and len(br.getAnimationList()) == 4
    assert br.loadAnimation(b"garbage") == -1 and len(br.getAnimationList()) == 4


def test_json_document_with_data_uri_and_sparse_accessor(W):
    b = G.GltfBuilder()
    pos, nrm, uv, idx = quad_mesh()
    a_pos = b.accessor(pos, G.F32, "VEC3", minmax=True)
    a_idx = b.accessor(idx, G.U16, "SCALAR")
    sp_i = b.view(np.array([1, 3], np.uint16).tobytes())
    sp_v = b.view(np.array([[2, 0, 0], [0, 3, 0]], f32).tobytes())
    b.doc["accessors"][a_pos]["sparse"] = {"count": 2, "indices": {"bufferView": sp_i, "componentType": G.U16},
                                            "values": {"bufferView": sp_v}}
    b.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": a_pos}, "indices": a_idx}]}]
    b.doc["nodes"] = [{"mesh": 0}]
    br = W.WorldBridge()
    br.loadScene("viewer", glbData=b.gltf_json())
    assert br.loadWarning == ""
    v = np.asarray(br.vertices).reshape(-1, 4)[ENV_VERTS:, :3]
    want = pos.copy()
    want[1], want[3] = [2, 0, 0], [0, 3, 0]
    assert np.array_equal(v, want)


def test_broken_glb_leaves_the_procedural_scene(W):
    b, _ = build_static(W)
    glb = b.glb()
    plain = W.WorldBridge()
    plain.loadScene("viewer")
    for bad in (glb[:40], b"glTF" + b"\0" * 30, b"{not json", glb[:20] + b"\xff" * 8 + glb[28:]):
        br = W.WorldBridge()
        br.loadScene("viewer", glbData=bad)
        assert br.loadWarning != ""
        # like the reference (lib.rs:57-67 ignores the loader's error): the room alone, without the placeholder sphere
        assert len(br.vertices) // 4 == ENV_VERTS and br.textureCount == 0 and br.getAnimationList() == []
        assert np.array_equal(np.asarray(br.vertices), np.asarray(plain.vertices)[:ENV_VERTS * 4])


@pytest.mark.gpu
def test_animated_textured_glb_renders_like_the_oracle(W, oracle_lib):
    PIL = pytest.importorskip("PIL.Image")
    b, _ = build_skinned(W)
    rng = np.random.default_rng(5)
    tex = (rng.integers(0, 256, (40, 60, 3), dtype=np.uint8) // 2 + 100).astype(np.uint8)
    jpg = io.BytesIO()
    PIL.fromarray(tex, "RGB").save(jpg, "JPEG", quality=90)
    t0 = b.image_texture(jpg.getvalue(), "image/jpeg")
    t1 = b.external_texture()
    uvs = np.stack([np.tile([0.0, 1.0], 7), np.repeat(np.linspace(0, 1, 7), 2)], 1).astype(f32)
    b.doc["meshes"][0]["primitives"][0]["attributes"]["TEXCOORD_0"] = b.accessor(uvs, G.F32, "VEC2")
    b.doc["materials"] = [{"pbrMetallicRoughness": {"metallicFactor": 0.0, "baseColorTexture": {"index": t0}},
                           "emissiveTexture": {"index": t1}}]
    b.doc["meshes"][0]["primitives"][0]["material"] = 0
    br = W.WorldBridge()
    br.loadScene("viewer", glbData=b.glb())
    assert br.textureCount == 2
    images = []
    for t in (0.0, 0.3):
        br.update(t)
        gpu, cpu = W.WebGPURenderer(0), oracle_lib.OracleRenderer()
        for r in (gpu, cpu):
            r.buildPipeline(6, 1)
            W.upload_scene(r, br, 96, 64)
            for f in (1, 2, 3):
                r.compute(f)
            r.present()
        gpu.sync()
        assert gpu.texture_warnings == []
        assert np.array_equal(gpu.readTextureLayer(1), np.full((1024, 1024, 4), 255, np.uint8))     # external image -> white
        assert np.array_equal(gpu.readAccum().view(np.uint32), cpu.readAccum().view(np.uint32))
        assert np.array_equal(gpu.captureFrame()["data"], cpu.captureFrame()["data"])
        gc, cc = gpu.getCounters(), cpu.getCounters()
        assert all(gc[k] == cc[k] for k in ("primary_rays", "extension_rays", "shadow_rays"))
        images.append(gpu.readAccum().copy())
        gpu.destroy()
    assert not np.array_equal(images[0], images[1])          # the pose changed the picture


def big_skinned_glb(W, nu=192, nv=96):
    """a skinned tube: nu x nv quads, two joints, one rotation channel"""
    b = G.GltfBuilder()
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False, dtype=f32)
    v = np.linspace(0, 1, nv + 1, dtype=f32)
    uu, vv = np.meshgrid(u, v)
    pos = np.stack([0.15 * np.cos(uu), vv, 0.15 * np.sin(uu)], -1).reshape(-1, 3).astype(f32)
    nrm = np.stack([np.cos(uu), np.zeros_like(uu), np.sin(uu)], -1).reshape(-1, 3).astype(f32)
    idx = []
    for j in range(nv):
        for i in range(nu):
            a, bq = j * nu + i, j * nu + (i + 1) % nu
            idx += [a, bq, bq + nu, a, bq + nu, a + nu]
    w1 = np.clip((pos[:, 1] - 0.3) * 2.5, 0, 1).astype(f32)
    weights = np.stack([1 - w1, w1, 0 * w1, 0 * w1], 1).astype(f32)
    joints = np.tile(np.array([0, 1, 0, 0], np.uint16), (len(pos), 1))
    acc = dict(POSITION=b.accessor(pos, G.F32, "VEC3", minmax=True), NORMAL=b.accessor(nrm, G.F32, "VEC3"),
               JOINTS_0=b.accessor(joints, G.U16, "VEC4"), WEIGHTS_0=b.accessor(weights, G.F32, "VEC4"))
    b.doc["meshes"] = [{"primitives": [{"attributes": acc, "indices": b.accessor(np.array(idx, np.uint32), G.U32, "SCALAR")}]}]
    ibm = np.stack([np.eye(4, dtype=f32), np.eye(4, dtype=f32)])
    ibm[1][1, 3] = -0.5
    b.doc["nodes"] = [{"mesh": 0, "skin": 0}, {"children": [2]}, {"translation": [0, 0.5, 0]}]
    b.doc["skins"] = [{"joints": [1, 2], "inverseBindMatrices": b.accessor(np.stack([m.T for m in ibm]), G.F32, "MAT4")}]
    s45, c45 = np.sin(np.pi / 8), np.cos(np.pi / 8)
    b.doc["animations"] = [{"name": "sway", "samplers": [{"input": b.accessor(np.array([0, 1, 2], f32), G.F32, "SCALAR", minmax=True),
                                                          "output": b.accessor(np.array([[0, 0, -s45, c45], [0, 0, s45, c45], [0, 0, -s45, c45]], f32), G.F32, "VEC4")}],
                            "channels": [{"sampler": 0, "target": {"node": 2, "path": "rotation"}}]}]
    return b.glb(), len(idx) // 3


def character_in_hall_glb(W, hall=(512, 256), nu=96, nv=48):
    """The usual animated scene: a small skinned, animated mesh (the tube of big_skinned_glb, nu x nv quads) inside a large
    STATIC one (a wavy hall[0] x hall[1]-quad sheet, 2 * hall[0] * hall[1] triangles) — two nodes, two geometries."""
    b = G.GltfBuilder()
    hu, hv = hall
    x = np.linspace(-1.2, 1.2, hu + 1, dtype=f32)
    z = np.linspace(-1.2, 1.2, hv + 1, dtype=f32)
    xx, zz = np.meshgrid(x, z)
    yy = (0.05 * np.sin(7 * xx) * np.cos(5 * zz)).astype(f32)
    hpos = np.stack([xx, yy, zz], -1).reshape(-1, 3).astype(f32)
    hnrm = np.tile(np.array([0, 1, 0], f32), (len(hpos), 1))
    a = (np.arange(hv)[:, None] * (hu + 1) + np.arange(hu)[None, :]).reshape(-1)
    hidx = np.stack([a, a + 1, a + hu + 2, a, a + hu + 2, a + hu + 1], 1).reshape(-1).astype(np.uint32)
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False, dtype=f32)
    v = np.linspace(0, 1, nv + 1, dtype=f32)
    uu, vv = np.meshgrid(u, v)
    pos = np.stack([0.15 * np.cos(uu), vv + 0.1, 0.15 * np.sin(uu)], -1).reshape(-1, 3).astype(f32)
    nrm = np.stack([np.cos(uu), np.zeros_like(uu), np.sin(uu)], -1).reshape(-1, 3).astype(f32)
    q = (np.arange(nv)[:, None] * nu + np.arange(nu)[None, :])
    qn = (np.arange(nv)[:, None] * nu + (np.arange(nu)[None, :] + 1) % nu)
    idx = np.stack([q, qn, qn + nu, q, qn + nu, q + nu], -1).reshape(-1).astype(np.uint32)
    w1 = np.clip((pos[:, 1] - 0.4) * 2.5, 0, 1).astype(f32)
    weights = np.stack([1 - w1, w1, 0 * w1, 0 * w1], 1).astype(f32)
    joints = np.tile(np.array([0, 1, 0, 0], np.uint16), (len(pos), 1))
    hall_acc = dict(POSITION=b.accessor(hpos, G.F32, "VEC3", minmax=True), NORMAL=b.accessor(hnrm, G.F32, "VEC3"))
    tube_acc = dict(POSITION=b.accessor(pos, G.F32, "VEC3", minmax=True), NORMAL=b.accessor(nrm, G.F32, "VEC3"),
                    JOINTS_0=b.accessor(joints, G.U16, "VEC4"), WEIGHTS_0=b.accessor(weights, G.F32, "VEC4"))
    b.doc["meshes"] = [{"primitives": [{"attributes": hall_acc, "indices": b.accessor(hidx, G.U32, "SCALAR")}]},
                       {"primitives": [{"attributes": tube_acc, "indices": b.accessor(idx, G.U32, "SCALAR")}]}]
    ibm = np.stack([np.eye(4, dtype=f32), np.eye(4, dtype=f32)])
    ibm[1][1, 3] = -0.6
    b.doc["nodes"] = [{"mesh": 0}, {"mesh": 1, "skin": 0}, {"children": [3]}, {"translation": [0, 0.6, 0]}]
    b.doc["scenes"][0]["nodes"] = [0, 1, 2]
    b.doc["skins"] = [{"joints": [2, 3], "inverseBindMatrices": b.accessor(np.stack([m.T for m in ibm]), G.F32, "MAT4")}]
    s45, c45 = np.sin(np.pi / 8), np.cos(np.pi / 8)
    b.doc["animations"] = [{"name": "sway", "samplers": [{"input": b.accessor(np.array([0, 1, 2], f32), G.F32, "SCALAR", minmax=True),
                                                          "output": b.accessor(np.array([[0, 0, -s45, c45], [0, 0, s45, c45], [0, 0, -s45, c45]], f32), G.F32, "VEC4")}],
                            "channels": [{"sampler": 0, "target": {"node": 3, "path": "rotation"}}]}]
    return b.glb(), len(hidx) // 3, len(idx) // 3


BRIDGE_ARRAYS = ("vertices", "normals", "uvs", "mesh_topology", "tlas", "blas", "instances", "lights", "draw_commands")


@pytest.mark.gpu
def test_gpu_blas_builder_equals_the_cpu_builder(W):
    """rt_build_blas against the scene compiler's BlasBuilder: node array and triangle order byte for byte, on meshes from a
    single triangle to 263 k triangles, degenerate and flat ones included; then as the hook of update(t)."""
    r = W.WebGPURenderer(0)
    rng = np.random.default_rng(2)

    def cpu_blas(verts4, idx):
        # the CPU builder through the bridge: a one-geometry GLB (static mesh) — geometry 2 of the viewer scene
        b = G.GltfBuilder()
        b.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": b.accessor(verts4[:, :3], G.F32, "VEC3", minmax=True)},
                                            "indices": b.accessor(idx, G.U32, "SCALAR")}]}]
        b.doc["nodes"] = [{"mesh": 0}]
        br = W.WorldBridge()
        br.loadScene("viewer", glbData=b.glb())
        blas = np.asarray(br.blas).reshape(-1, 8)
        inst = np.asarray(br.instances).reshape(-1, 36)
        off = int([row for row in inst if row[34:35].view(np.uint32)[0] == 2][0][32:33].view(np.uint32)[0])
        nodes = blas[off:].copy()
        topo = np.asarray(br.mesh_topology).reshape(-1, 20)
        return nodes, topo[topo[:, 3] == 2][:, :3] - ENV_VERTS, 12      # 12 = triangles of the room before this geometry

    cases = []
    for n_tris in (1, 3, 4, 5, 9, 63, 64, 65, 129, 1000, 4095, 4096, 4097, 9000, 20000):   # around the builder's thresholds: 64 (in-wave subtrees), 4 096 (large-node kernels)
        verts = rng.random((n_tris * 3, 3), dtype=f32) * f32(2) - f32(1)
        if n_tris == 64:
            verts[:, 2] = 0.25                      # flat: padded boxes
        if n_tris == 9:
            verts[:] = verts[0]                     # all triangles identical: no split possible -> fat leaf (count > 7 quirk)
        cases.append((np.concatenate([verts, np.ones((len(verts), 1), f32)], 1), np.arange(n_tris * 3, dtype=np.uint32)))
    grid = np.stack(np.meshgrid(np.linspace(-1, 1, 40, dtype=f32), np.linspace(-1, 1, 40, dtype=f32)), -1).reshape(-1, 2)
    gv = np.concatenate([grid, np.zeros((len(grid), 1), f32), np.ones((len(grid), 1), f32)], 1)       # z = +0 / -0 mix
    gv[::3, 2] = -0.0
    gi = []
    for j in range(39):
        for i in range(39):
            a = j * 40 + i
            gi += [a, a + 1, a + 41, a, a + 41, a + 40]
    cases.append((gv, np.array(gi, np.uint32)))
    for verts4, idx in cases:
        nodes, order = r.buildBlas(verts4, idx)
        want_nodes, want_tris, topo0 = cpu_blas(verts4, idx)
        # the bridge stores leaf `first` as a global topology index: undo that for the comparison
        wn = want_nodes.copy()
        data = wn[:, 7].view(np.uint32)
        leaf = data != 0
        data[leaf] = (((data[leaf] >> 3) - topo0) << 3) | (data[leaf] & 7)
        assert nodes.shape == wn.shape, (len(idx) // 3, nodes.shape, wn.shape)
        assert np.array_equal(nodes.view(np.uint32), wn.view(np.uint32)), len(idx) // 3
        assert np.array_equal(idx.reshape(-1, 3)[order], want_tris)
    # as the hook of update(t): every bridge array identical, on the big static scene and on an animated one
    for scene, glb in (("sponza_like", None), ("glass_blob", None), ("viewer", big_skinned_glb(W)[0])):
        cpu_b, gpu_b = W.WorldBridge(), W.WorldBridge()
        gpu_b.setBlasBuilder(r)
        cpu_b.loadScene(scene, glbData=glb)
        gpu_b.loadScene(scene, glbData=glb)
        for t in ((0.0,) if glb is None else (0.0, 0.4, 1.7)):
            cpu_b.update(t)
            gpu_b.update(t)
            for k in BRIDGE_ARRAYS:
                assert np.array_equal(np.asarray(getattr(cpu_b, k)).view(np.uint32), np.asarray(getattr(gpu_b, k)).view(np.uint32)), (scene, t, k)
    r.destroy()


@pytest.mark.gpu
def test_builder_hook_refuses_a_destroyed_renderer(W):
    r = W.WebGPURenderer(0)
    b = W.WorldBridge()
    b.setBlasBuilder(r)
    b.loadScene("cornell")
    b.update(0.0)
    r.destroy()
    with pytest.raises(RuntimeError):
        b.update(0.1)
    b.setBlasBuilder(None)
    b.update(0.1)


def test_parallel_update_equals_single_threaded(W, tmp_path):
    """update(t) spreads skinning and topology packing over a pool of host threads; with MS_THREADS=1 (a fresh process)
    everything runs on the calling thread — the arrays must be the same, byte for byte."""
    import hashlib
    import subprocess
    import sys
    glb, n_tris = big_skinned_glb(W, 256, 128)          # 65 536 triangles: above the per-thread minimum
    path = tmp_path / "tube.glb"
    path.write_bytes(glb)
    code = ("import sys,hashlib,json;sys.path.insert(0,%r);import numpy as np;import webgpu_raytracer_amd as W;"
            "b=W.WorldBridge();b.loadScene('viewer',glbData=open(%r,'rb').read());b.update(0.4);"
            "print(json.dumps({k:hashlib.sha256(np.ascontiguousarray(getattr(b,k)).tobytes()).hexdigest() for k in %r}))"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(path), list(BRIDGE_ARRAYS)))
    import json
    import os as _os
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True,
                         env=dict(_os.environ, MS_THREADS="1")).stdout
    single = json.loads(out.strip().splitlines()[-1])
    b = W.WorldBridge()
    b.loadScene("viewer", glbData=glb)
    b.update(0.4)
    for k in BRIDGE_ARRAYS:
        assert hashlib.sha256(np.ascontiguousarray(getattr(b, k)).tobytes()).hexdigest() == single[k], k
