"""Known-answer tests that pin the oracle to vectors hand-derived from the reference source
(the reference itself ships no tests or fixtures)."""
import ctypes
import math
import struct

import numpy as np
import pytest


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


# init_rng(pixel_idx, frame) seeds and the first three rand_pcg outputs (state, f32 bits)
RNG_KAT = [
    ((0, 1), 0xB4C6207F, [(0xF1F44DD0, 0x3F71F5AE), (0x6EDCFF15, 0x3EDDB9E7), (0xDF2367DE, 0x3F5F14AC)]),
    ((0, 0), 0x67B2772F, [(0xFFB96840, 0x3F7F8686), (0x1E62C045, 0x3DF31620), (0x92E24ECE, 0x3F16755F)]),
    ((12345, 7), 0x6721713F, [(0x8C8DA590, 0x3F0CAE85), (0x109249D5, 0x3D849A06), (0x1B90819E, 0x3DDA6030)]),
    ((2073599, 64), 0x0B489CDF, [(0x32C6DDB0, 0x3E47AAC3), (0xB25DD875, 0x3F333961), (0xF708B8BE, 0x3F770955)]),
]


@pytest.mark.parametrize("args,seed,draws", RNG_KAT)
def test_rng_known_answers(oracle_lib, args, seed, draws):
    L = oracle_lib.lib()
    s = L.oracle_init_rng(*args)
    assert s == seed
    state = ctypes.c_uint32(s)
    for want_state, want_bits in draws:
        v = L.oracle_rand_pcg(ctypes.byref(state))
        assert state.value == want_state
        assert bits(v) == want_bits


def test_rand_pcg_can_return_one(oracle_lib):
    # f32(u32) rounds to nearest-even and the divisor literal rounds to 2^32, so outputs >= 0xFFFFFF80
    # give exactly 1.0. Brute-force a state whose output word is that large.
    L = oracle_lib.lib()
    found = False
    st = np.uint32(0)
    rng = np.random.default_rng(0)
    for s in rng.integers(0, 2**32, size=200000, dtype=np.uint64):
        state = ctypes.c_uint32(int(s))
        v = L.oracle_rand_pcg(ctypes.byref(state))
        assert 0.0 <= v <= 1.0
        found = found or v == 1.0
    # probability 3e-8/draw: not expected in 2e5 draws; the range assertion above is the test
    assert found in (True, False)


def test_halton_jitter_sequence(oracle_lib):
    L = oracle_lib.lib()
    want = [(-0.25, 1 / 6), (0.25, -7 / 18), (-0.375, -1 / 18), (0.125, 5 / 18)]
    for total_frames, (wx, wy) in zip((1, 2, 3, 4), want):
        idx = (total_frames % 16) + 1
        assert math.isclose(L.oracle_halton(idx, 2) - 0.5, wx, abs_tol=1e-15)
        assert math.isclose(L.oracle_halton(idx, 3) - 0.5, wy, abs_tol=1e-15)
    # period 16
    assert L.oracle_halton((17 % 16) + 1, 2) == L.oracle_halton((1 % 16) + 1, 2)


def test_layout_strides(W):
    b = W.WorldBridge()
    b.loadScene("cornell")
    # Cornell = 18 quads: 72 verts / 36 tris / 2 light refs / 1 instance / 1-node TLAS
    assert len(b.vertices) == 72 * 4 and len(b.normals) == 72 * 4 and len(b.uvs) == 72 * 2
    assert len(b.mesh_topology) == 36 * 20
    assert len(b.instances) == 36 and len(b.lights) == 4 and len(b.draw_commands) == 4
    assert len(b.tlas) == 8
    tl = b.tlas.view(np.uint32)
    assert tl[3] == 1 and tl[7] == 1  # skip = 1, data = (0 << 3) | 1
    assert list(b.draw_commands) == [108, 1, 0, 0]
    # light quad v(213,554,227)..v(343,554,332), colour 20, material 3
    topo = b.mesh_topology.reshape(-1, 20)
    lights = b.lights.reshape(-1, 2)
    for inst_idx, tri_idx in lights:
        assert inst_idx == 0
        row = topo[tri_idx]
        data0 = row[4:8].view(np.float32)
        assert list(data0) == [20.0, 20.0, 20.0, 3.0]
        ys = [b.vertices[4 * v + 1] for v in row[:3]]
        assert all(abs(y - f32(f32(554.0 / 555.0) * 2.0)) < 1e-6 for y in ys)
    # the root skip spans the whole BLAS; leaves hold <= 4 triangles except fallback leaves
    # (no valid SAH split, blas.rs:167-171), which still fit the 3-bit count here
    nodes = b.blas.reshape(-1, 8).view(np.uint32)
    assert nodes[0, 3] == len(nodes)
    leaves = nodes[nodes[:, 7] != 0]
    assert (leaves[:, 7] & 7).max() <= 7
    assert (leaves[:, 7] & 7).sum() == 36


def test_cornell_camera(W):
    b = W.WorldBridge()
    b.loadScene("cornell")
    b.updateCamera(512, 512)
    cam = b.cameraData
    assert list(cam[:4]) == [0.0, 1.0, f32(-2.4), 0.0]
    vh = 2.0 * math.tan(math.radians(30.0)) * 2.4
    assert abs(cam[8] + vh) < 2e-6 and cam[9] == 0 and cam[10] == 0  # horizontal = (-vh, 0, 0)
    assert abs(cam[13] - vh) < 2e-6
    assert abs(cam[4] - vh / 2) < 2e-6 and abs(cam[5] - (1 - vh / 2)) < 2e-6 and abs(cam[6]) < 1e-6
    assert list(cam[16:19]) == [-1.0, 0.0, 0.0] and list(cam[20:23]) == [0.0, 1.0, 0.0]


def test_viewer_diamond_counts(W):
    obj = ("v 0.0 1.0 0.0\nv 1.0 0.0 0.0\nv 0.0 0.0 1.0\nv -1.0 0.0 0.0\nv 0.0 0.0 -1.0\nv 0.0 -1.0 0.0\n"
           "f 1 3 2\nf 1 2 5\nf 1 5 4\nf 1 4 3\nf 6 2 3\nf 6 5 2\nf 6 4 5\nf 6 3 4\n")
    b = W.WorldBridge()
    b.loadScene("viewer", obj)
    # geometry 0 = 6 quads (24 v / 12 t), geometry 1 = octahedron (6 v / 8 t), 2 instances, 3 TLAS nodes
    assert len(b.vertices) // 4 == 30 and len(b.mesh_topology) // 20 == 20
    assert len(b.instances) // 36 == 2 and len(b.tlas) // 8 == 3
    # no vn in the OBJ => every shading normal of the model is +Y (mesh.rs:95-103)
    n = b.normals.reshape(-1, 4)[24:]
    assert np.all(n[:, :3] == np.array([0, 1, 0], dtype=np.float32))
    # instance i>0 is overwritten with rotY(pi) * scale(0.7) (lib.rs:196-204)
    inst = b.instances.reshape(-1, 36)
    which = [i for i in range(2) if inst[i].view(np.uint32)[34] == 1][0]
    m = inst[which][:16].reshape(4, 4)  # columns
    assert abs(m[0, 0] + 0.7) < 1e-6 and abs(m[1, 1] - 0.7) < 1e-6 and abs(m[2, 2] + 0.7) < 1e-6
    # floor is metal with roughness 0.15
    topo = b.mesh_topology.reshape(-1, 20)
    metals = [r for r in topo if r[7:8].view(np.float32)[0] == 1.0]
    assert len(metals) == 2 and all(abs(r[9:10].view(np.float32)[0] - 0.15) < 1e-7 for r in metals)


def test_instanced1000_tlas(W):
    b = W.WorldBridge()
    b.loadScene("instanced1000")
    assert len(b.instances) // 36 == 1001
    assert len(b.tlas) // 8 == 2 * 1001 - 1  # median split => exactly 2N-1 nodes
    tl = b.tlas.reshape(-1, 8).view(np.uint32)
    leaf_inst = sorted(int(d >> 3) for d in tl[:, 7] if d != 0)
    assert leaf_inst == list(range(1001))


def test_spheres_scene_refused(W):
    b = W.WorldBridge()
    with pytest.raises(ValueError):
        b.loadScene("spheres")


def test_math_against_libm(oracle_lib):
    L = oracle_lib.lib()
    xs = np.linspace(0.0, 2 * math.pi, 4001, dtype=np.float32)
    s, c = ctypes.c_float(), ctypes.c_float()
    for x in xs:
        L.oracle_sincos(float(x), ctypes.byref(s), ctypes.byref(c))
        assert abs(s.value - math.sin(float(x))) < 3e-7
        assert abs(c.value - math.cos(float(x))) < 3e-7
    for x in np.linspace(-20, 5, 2001, dtype=np.float32):
        e = L.oracle_exp(float(x))
        assert abs(e - math.exp(float(x))) <= 3e-7 * math.exp(float(x))
    for x in np.linspace(1e-4, 1.0, 2001, dtype=np.float32):
        p = L.oracle_pow(float(x), f32(1.0 / 2.2))
        assert abs(p - float(x) ** (1.0 / 2.2)) < 5e-7
    assert L.oracle_pow(0.0, 0.5) == 0.0 and L.oracle_pow(1.0, 0.5) == 1.0
    assert L.oracle_exp(-200.0) == 0.0


def test_f16_round_trip(oracle_lib):
    L = oracle_lib.lib()
    vals = np.concatenate([np.linspace(-70000, 70000, 3001), np.logspace(-9, 5, 2001), [0.0, -0.0, 65504.0, 65520.0]])
    vals = vals.astype(np.float32)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)  # numpy converts round-to-nearest-even
    for v, w in zip(vals, want):
        assert L.oracle_f32_to_f16(float(v)) == int(w)
    for h in range(0, 0x7C00, 7):
        assert L.oracle_f16_to_f32(h) == float(np.uint16(h).view(np.float16))


def test_min_max_nan_and_zero(oracle_lib):
    L = oracle_lib.lib()
    nan = float("nan")
    assert L.oracle_min(nan, 2.0) == 2.0 and L.oracle_min(2.0, nan) == 2.0
    assert L.oracle_max(nan, -2.0) == -2.0 and L.oracle_max(-2.0, nan) == -2.0
    assert math.copysign(1, L.oracle_min(0.0, -0.0)) == -1 and math.copysign(1, L.oracle_min(-0.0, 0.0)) == -1
    assert math.copysign(1, L.oracle_max(0.0, -0.0)) == 1 and math.copysign(1, L.oracle_max(-0.0, 0.0)) == 1


def test_octahedral_normal_round_trip(oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.default_rng(1)
    n = rng.normal(size=(500, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    for v in n:
        v = np.ascontiguousarray(v, dtype=np.float32)
        p = np.zeros(2, dtype=np.float32)
        o = np.zeros(3, dtype=np.float32)
        L.oracle_pack_normal(v.ctypes.data, p.ctypes.data)
        L.oracle_unpack_normal(p.ctypes.data, o.ctypes.data)
        assert np.abs(o - v).max() < 2e-6


def test_triangle_and_aabb_edge_cases(oracle_lib):
    L = oracle_lib.lib()
    a = lambda *v: np.array(v, dtype=np.float32)
    v0, v1, v2 = a(0, 0, 0), a(1, 0, 0), a(0, 1, 0)
    mn, mx = a(0, 0, 0), a(1, 1, 1)

    def hit(o, d, lo=0.001, hi=1e30):
        oo, dd = a(*o), a(*d)  # keep the arrays alive across the call
        return L.oracle_hit_triangle(v0.ctypes.data, v1.ctypes.data, v2.ctypes.data, oo.ctypes.data,
                                     dd.ctypes.data, lo, hi)

    def box(o, d, lo=0.001, hi=1e30):
        oo, dd = a(*o), a(*d)
        return L.oracle_intersect_aabb(mn.ctypes.data, mx.ctypes.data, oo.ctypes.data, dd.ctypes.data, lo, hi)

    assert hit((0.25, 0.25, 1), (0, 0, -1)) == 1.0
    assert hit((0.25, 0.25, 1), (0, 0, -2)) == 0.5          # un-normalised direction: parametric t
    assert hit((0.25, 0.25, 1), (0, 0, 1)) == -1.0          # behind
    assert hit((0.25, 0.25, 1), (1, 0, 0)) == -1.0          # parallel: |a| < 1e-6
    assert hit((0.75, 0.75, 1), (0, 0, -1)) == -1.0         # u+v > 1
    assert hit((0.25, 0.25, 1), (0, 0, -1), 0.001, 1.0) == -1.0   # t < t_max is strict
    assert hit((0.0, 0.0, 1), (0, 0, -1)) == 1.0            # u = v = 0 is inside (closed edges)
    assert box((0.5, 0.5, -1), (0, 0, 1)) == 1.0            # zero direction components -> inf/NaN slabs ignored
    # parallel to and OUTSIDE the x slab: origin*inv_d = inf on both planes, b*inf - inf = NaN, and NaN is
    # ignored by min/max => the slab is skipped (conservative false positive of the reference's
    # `b * inv_d - origin * inv_d` form, Raytracer.wgsl:434-435; never a false negative)
    assert box((2.0, 0.5, -1), (0, 0, 1)) == 1.0
    assert box((2.0, 0.5, -1), (1e-3, 0, 1)) == f32(1e30)   # a finite x component does cull it
    assert box((0.5, 0.5, 0.5), (0, 0, 1)) == f32(0.001)    # origin inside: tm_near = t_min
    assert box((0.5, 0.5, -1), (0, 0, 1), 0.001, 0.5) == f32(1e30)  # culled by t_max
