"""Independent pins of the scene compiler's BVH — none of them is a restatement of the builder:

 (a) known-answer node arrays WORKED OUT BY HAND from /root/reference/rust-shader-tools/src/bvh/blas.rs:87-217 (bins, SAH
     costs, two-pointer partition, child rotation) for two small meshes with binary-exact coordinates; the derivations
     are in the docstrings, the expected arrays in tests/golden/blas_kat.json; the same for the TLAS builder
     (bvh/tlas.rs:58-111: axis rule, STABLE sort, costlier half first; tests/golden/tlas_kat.json), checked against the CPU
     builder here and against the device kernel k_tlas in tests/test_gpu_world_update.py;
 (b) structural validity of every scene's TLAS / BLAS arrays: every triangle of a geometry sits in exactly one leaf of its
     BLAS, every node box encloses what is below it, skip pointers describe a pre-order tree;
 (c) ray queries: the restated stackless traversal over those arrays against brute force over every (instance, triangle)
     pair (triangle ranges from the draw commands, not from the node arrays) with the same hit_triangle_raw.
"""
import ctypes
import json
import os

import numpy as np
import pytest

import parity_util as pu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(REPO, "tests", "golden", "blas_kat.json")))
TLAS_KAT = json.load(open(os.path.join(REPO, "tests", "golden", "tlas_kat.json")))


def cpu_build_blas(W, verts, tris):
    lib = ctypes.CDLL(W._build.build_scene())
    v4 = np.zeros((len(verts), 4), np.float32)
    v4[:, :3] = np.asarray(verts, np.float32)
    idx = np.ascontiguousarray(np.asarray(tris, np.uint32).reshape(-1))
    n_tris = idx.size // 3
    nodes = np.zeros((2 * n_tris, 8), np.float32)
    order = np.zeros(n_tris, np.uint32)
    n_nodes = ctypes.c_uint32()
    vp = ctypes.c_void_p
    lib.ms_build_blas.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint32, vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), vp]
    rc = lib.ms_build_blas(v4.ctypes.data_as(vp), len(verts), idx.ctypes.data_as(vp), n_tris, nodes.ctypes.data_as(vp),
                           nodes.shape[0], ctypes.byref(n_nodes), order.ctypes.data_as(vp))
    assert rc == 0
    nodes = nodes[:n_nodes.value]
    u = nodes.view(np.uint32)
    return [{"min": nodes[i, 0:3].tolist(), "skip": int(u[i, 3]), "max": nodes[i, 4:7].tolist(), "data": int(u[i, 7])}
            for i in range(len(nodes))], order.tolist()


def mesh_from_boxes(boxes):
    """One triangle per (x, y, z, dx, dy, dz): vertices (x, y, z), (x+dx, y+dy, z), (x, y, z+dz) — its AABB is exactly
    [x, x+dx] x [y, y+dy] x [z, z+dz] and no extent is below the 1e-5 flat-triangle padding of blas.rs:41-52."""
    verts, tris = [], []
    for (x, y, z, dx, dy, dz) in boxes:
        b = len(verts)
        verts += [(x, y, z), (x + dx, y + dy, z), (x, y, z + dz)]
        tris.append((b, b + 1, b + 2))
    return verts, tris


def test_blas_known_answer_six_triangles_along_x(W):
    """Six unit-box triangles T0..T5 at x = 8, 0, 15, 1, 10, 13 (all boxes [x, x+1] x [0,1] x [0,1]).

    subdivide(0, 6) (blas.rs:87): node box = [0,16] x [0,1] x [0,1]; 6 > 4 so no leaf (:99).  extent = (16, 1, 1): y > x is
    false, z > x is false -> axis 0 (:106).  split_len 16, scale = 16/16 = 1 (:122), so bin = floor(centre.x) with
    centres 8.5, 0.5, 15.5, 1.5, 10.5, 13.5 -> bins 8, 0, 15, 1, 10, 13 (:124-127).
    area of a box w x 1 x 1 = 2 (w + 1 + w) = 4w + 2 (primitives.rs:41-50).
      left sweep (:142-149)   i=0: 1 tri [0,1] -> 6 | i=1..7: 2, [0,2] -> 10 | i=8,9: 3, [0,9] -> 38 | i=10..12: 4, [0,11] -> 46
      right sweep (:151-158)  i=15,14: 1, [15,16] -> 6 | i=13..11: 2, [13,16] -> 14 | i=10,9: 3, [10,16] -> 26
                              | i=8..2: 4, [8,16] -> 34 | i=1: 5, [1,16] -> 62
      cost(i) = left_area[i] * left_count[i] + right_area[i+1] * right_count[i+1] (:160-169):
        i=0: 6 + 62*5 = 316;  i=1..7: 10*2 + 34*4 = 156;  i=8,9: 38*3 + 26*3 = 192;  i=10: 46*4 + 14*2 = 212; ...
      strict `<` keeps the first minimum: best_split = 1.
    Partition (:179-199) on [0,1,2,3,4,5], i=0, j=5: T0 (bin 8) is right, T5 (13) right -> j=4, T4 (10) right -> j=3, T3 (1)
    left -> swap -> [3,1,2,0,4,5], i=1, j=2; T1 (0) left -> i=2; T2 (15) right, order[j=2] = T2 right -> j=1; stop.
    l_count = 2.  l_cost = left_area[1] * 2 = 20, r_cost = right_area[2] * 4 = 136 > l_cost -> rotate_left(2) (:209-217):
    order = [2,0,4,5,3,1], first child = the four right triangles.
      node 0: box [0,16]x[0,1]x[0,1], data 0, skip 3
      node 1: subdivide(0,4): T2,T0,T4,T5 -> box [8,16]x[0,1]x[0,1], leaf: data = (0 << 3) | 4 = 4, skip 2
      node 2: subdivide(4,2): T3,T1 -> box [0,2]x[0,1]x[0,1], leaf: data = (4 << 3) | 2 = 34, skip 3"""
    verts, tris = mesh_from_boxes([(x, 0, 0, 1, 1, 1) for x in (8, 0, 15, 1, 10, 13)])
    nodes, order = cpu_build_blas(W, verts, tris)
    assert order == KAT["six_along_x"]["order"] == [2, 0, 4, 5, 3, 1]
    assert nodes == KAT["six_along_x"]["nodes"]


def test_blas_known_answer_axis_rule_is_not_longest_axis(W):
    """Five triangles with boxes [0,2] x [y,y+1] x [z,z+1], (y, z) = T0 (0,0), T1 (0,7), T2 (2,3), T3 (2,5), T4 (1,1).

    Node box [0,2] x [0,3] x [0,8], extent (2, 3, 8).  blas.rs:106: `extent.y > extent.x` is TRUE -> axis 1, although z is
    the longest axis (a longest-axis builder would split along z and give a different tree).  split_len 3,
    scale = 16/3 (f32 5.3333335); centre.y = 0.5, 0.5, 2.5, 2.5, 1.5 -> (c * scale) = 2.67, 2.67, 13.33, 13.33, 8.0 -> bins
    2, 2, 13, 13, 8.  Bin boxes: b2 = [0,2]x[0,1]x[0,8], b8 = [0,2]x[1,2]x[1,2], b13 = [0,2]x[2,3]x[3,6].
    area(dx,dy,dz) = 2 (dx dy + dy dz + dz dx); an empty prefix has area 0 (primitives.rs:44-46).
      left:  i<2: 0 tris | i=2..7: 2, (2,1,8) -> 52 | i=8..12: 3, (2,2,8) -> 72 | i>=13: 5, (2,3,8) -> 92
      right: i>=14: 0 | i=13..9: 2, (2,1,3) -> 22 | i=8..3: 3, [0,2]x[1,3]x[1,6] = (2,2,5) -> 48 | i<=2: 5 -> 92
      cost: i=0,1 skipped (left empty); i=2..7: 52*2 + 48*3 = 248; i=8..12: 72*3 + 22*2 = 260; i=13,14 skipped.
      best_split = 2.
    Partition on [0,1,2,3,4]: T0, T1 left -> i=2; T2 right; j=4: T4 (8) right -> j=3: T3 right -> j=2: T2 right -> j=1; stop.
    l_count = 2; l_cost = 52*2 = 104 < r_cost = 48*3 = 144 -> rotate: order = [2,3,4,0,1].
      node 0: [0,2]x[0,3]x[0,8], data 0, skip 3
      node 1: T2,T3,T4 -> [0,2]x[1,3]x[1,6], data = 3, skip 2
      node 2: T0,T1 -> [0,2]x[0,1]x[0,8], data = (3 << 3) | 2 = 26, skip 3"""
    verts, tris = mesh_from_boxes([(0, y, z, 2, 1, 1) for (y, z) in ((0, 0), (0, 7), (2, 3), (2, 5), (1, 1))])
    nodes, order = cpu_build_blas(W, verts, tris)
    assert order == KAT["axis_rule"]["order"] == [2, 3, 4, 0, 1]
    assert nodes == KAT["axis_rule"]["nodes"]


def test_blas_small_meshes_become_one_leaf(W):
    """count <= 4 -> leaf at once (blas.rs:99-103): data = (0 << 3) | count, skip = 1; flat triangles are padded by 5e-6."""
    verts, tris = mesh_from_boxes([(0, 0, 0, 1, 1, 1), (4, 0, 0, 1, 1, 1), (9, 0, 0, 1, 1, 1)])
    nodes, order = cpu_build_blas(W, verts, tris)
    assert order == [0, 1, 2] and len(nodes) == 1
    assert nodes[0] == {"min": [0.0, 0.0, 0.0], "skip": 1, "max": [10.0, 1.0, 1.0], "data": 3}
    flat, _ = cpu_build_blas(W, [(0, 0, 0), (1, 0, 0), (0, 0, 1)], [(0, 1, 2)])     # size.y = 0 < 1e-5 -> +-0.5e-5
    half = float(np.float32(1e-5) * np.float32(0.5))
    assert flat[0]["min"] == [0.0, -half, 0.0] and flat[0]["max"] == [1.0, half, 1.0]


def decoded_coverage(leaf_data, n_tris):
    """Triangles the shader reaches: it decodes a leaf word as first = data >> 3, count = data & 7 (Raytracer.wgsl:469-471)."""
    cov = np.zeros(n_tris, np.int64)
    for d in leaf_data.tolist():
        cov[(d >> 3):(d >> 3) + (d & 7)] += 1
    return cov


# triangles hidden by the 3-bit leaf-count overflow of blas.rs:111-115, per scene (0 everywhere else)
UNREACHABLE = {"special": 280, "mixed": 1920, "sponza_like": 1384, "glass_blob": 8}   # "mixed": 16 bins over a 40-unit floor leave 547 / 809 / 547 centroids in ONE bin -> no split -> three giant fallback leaves
# "sponza_like" / "glass_blob" are this repo's synthetic stand-ins: blas.rs:106 picks axis y whenever extent.y > extent.x,
# so a one-cell-high strip of a REGULAR grid running along z has every centroid in ONE y-bin, cannot be split and becomes
# a fallback leaf of 8-46 triangles whose count overflows — round 2's regular sponza-like mesh hid 83 160 of its 263 176
# triangles (31.6 %) from every ray.  Round 3 jitters the grid vertices on their surfaces (scene_compiler.cpp
# add_grid_patch) and makes the column cells square: 1 384 (0.53 %) remain, with the reference's builder untouched.  The
# numbers are pinned here so that a change of the generators or of the builder shows.
SCENES = ["cornell", "viewer_diamond", "viewer_diamond_1k", "special", "mixed", "mesh", "instanced1000", "sponza_like", "glass_blob"]


def _nodes(a):
    a = np.asarray(a, np.float32).reshape(-1, 8)
    u = a.view(np.uint32)
    return a[:, 0:3], a[:, 4:7], u[:, 3].astype(np.int64), u[:, 7].astype(np.int64)


@pytest.mark.parametrize("scene", SCENES)
def test_bvh_arrays_are_a_valid_hierarchy(W, scene):
    b = pu.bridge_for(W, scene)
    pos = np.asarray(b.vertices, np.float32).reshape(-1, 4)[:, :3]
    topo = np.asarray(b.mesh_topology, np.uint32).reshape(-1, 20)
    tri_v = pos[topo[:, 0:3].astype(np.int64)]                      # (n_tris, 3 corners, xyz)
    tmin, tmax = tri_v.min(axis=1), tri_v.max(axis=1)
    inst = np.asarray(b.instances, np.float32).reshape(-1, 36)
    blas_off = inst.view(np.uint32)[:, 32].astype(np.int64)
    dc = np.asarray(b.draw_commands, np.uint32).reshape(-1, 4).astype(np.int64)
    assert len(dc) == len(inst) and (dc[:, 1] == 1).all() and (dc[:, 3] == np.arange(len(inst))).all()
    bmin, bmax, skip, data = _nodes(b.blas)
    n = len(skip)
    seen_tris, unreachable = {}, {}
    dc_range = {int(blas_off[i]): (int(dc[i, 2] // 3), int(dc[i, 2] // 3 + dc[i, 0] // 3)) for i in range(len(inst))}
    for root in sorted(set(blas_off.tolist())):
        end = root + skip[root]                                       # skip pointers are relative to the BLAS root
        assert root < end <= n
        idx = np.arange(root, end)
        # they go forward and stay inside the BLAS: a pre-order tree
        assert (root + skip[idx] <= end).all() and (root + skip[idx] > idx).all()
        leaves = idx[data[idx] != 0]
        first, count = data[leaves] >> 3, data[leaves] & 7
        # leaves tile a contiguous triangle range without gaps or overlaps ...
        o = np.argsort(first)
        f, c = first[o], count[o]
        tiled = bool((c >= 1).all() and (f[1:] == f[:-1] + c[:-1]).all())
        if scene not in UNREACHABLE:
            assert tiled
        if tiled:
            seen_tris[root] = (int(f[0]), int(f[-1] + c[-1]))
        else:
            # ... except where a fallback leaf holds more than 7 triangles: blas.rs:111-115 stores `first << 3 | count`
            # with an unmasked count, so the word decodes to another (first, count).  "special" has three such leaves
            # (its light sphere's coincident-centre triangles: raw words 210 = 0 << 3 | 210, 2139, 2169): 280 of its 564
            # triangles are unreachable, in the reference as here; a leaf of exactly 8 decodes to count 0 ("mixed").
            # Only the decode-independent invariants are checked for such a BLAS.
            cov = decoded_coverage(data[leaves], len(topo))
            unreachable[root] = int((cov[dc_range[root][0]:dc_range[root][1]] == 0).sum())
            assert cov.max() == 1
            seen_tris[root] = dc_range[root]
            first, count, leaves = first[:0], count[:0], leaves[:0]
        # every leaf box encloses its triangles; every inner box encloses its two children
        for k, (ff, cc) in zip(leaves, zip(first, count)):
            assert (bmin[k] <= tmin[ff:ff + cc].min(axis=0)).all() and (bmax[k] >= tmax[ff:ff + cc].max(axis=0)).all()
        inner = idx[data[idx] == 0]
        left = inner + 1
        right = root + skip[left]
        assert (right < end).all()
        for k, l, r in zip(inner, left, right):
            assert (bmin[k] <= np.minimum(bmin[l], bmin[r])).all() and (bmax[k] >= np.maximum(bmax[l], bmax[r])).all()
            assert root + skip[r] == root + skip[k]                     # the right child ends where its parent ends
    assert sum(unreachable.values()) == UNREACHABLE.get(scene, 0)
    # the triangle range a BLAS covers is the range the instance's draw command names (lib.rs:237-262)
    for i in range(len(inst)):
        lo, hi = seen_tris[int(blas_off[i])]
        assert (dc[i, 2] // 3, dc[i, 2] // 3 + dc[i, 0] // 3) == (lo, hi)
    # TLAS: absolute skip pointers, one instance per leaf, every instance exactly once
    tmin_, tmax_, tskip, tdata = _nodes(b.tlas)
    nt = len(tskip)
    assert tskip[0] == nt and (tskip > np.arange(nt)).all() and (tskip <= nt).all()
    leaf = tdata != 0
    assert (tdata[leaf] & 7 == 1).all()
    assert sorted((tdata[leaf] >> 3).tolist()) == list(range(len(inst)))


def _random_rays(bridge, n, seed):
    rng = np.random.default_rng(seed)
    pos = np.asarray(bridge.vertices, np.float32).reshape(-1, 4)[:, :3]
    inst = np.asarray(bridge.instances, np.float32).reshape(-1, 36)
    # ray origins inside (and a little around) the world box of the TLAS root
    root = np.asarray(bridge.tlas, np.float32).reshape(-1, 8)[0]
    ext = root[4:7] - root[0:3]
    lo, hi = root[0:3] - 0.1 * ext, root[4:7] + 0.1 * ext
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # A fifth of the rays are NEARLY axis-parallel (other components ~1e-6: huge but finite slab distances).  Exactly
    # axis-parallel rays are left out: the reference's slab form `b * inv_d - o * inv_d` (Raytracer.wgsl:76-86,427-441)
    # evaluates inf - inf there and rejects boxes the ray is inside of — a measure-zero set for a jittered camera,
    # pinned as such in test_oracle_kat.py::test_triangle_and_aabb_edge_cases.
    k = n // 5
    d[:k] = rng.normal(size=(k, 3)).astype(np.float32) * np.float32(1e-6)
    d[np.arange(k), rng.integers(0, 3, k)] = rng.choice(np.array([-1.0, 1.0], np.float32), k)
    rays = np.empty((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 0.001, d, 1e30
    return rays


@pytest.mark.parametrize("scene,n_rays", [("cornell", 100000), ("viewer_diamond", 100000), ("special", 100000),
                                          ("mixed", 100000), ("mesh", 100000), ("instanced1000", 20000),
                                          ("sponza_like", 1500), ("glass_blob", 1500)])
def test_traversal_finds_the_brute_force_closest_hit(W, oracle_lib, scene, n_rays):
    """Closest hit through TLAS + BLAS == closest hit over every triangle of every instance.  Exact: the same
    hit_triangle_raw produces both t values, so equal hits have equal bits.  A node box is computed from the vertices
    and the hit distance by Moller-Trumbore, so a triangle whose hit lies within rounding of its own box face — or of the
    current closest hit — can be culled by the slab test; such rays (the reference culls them too) must be rare and
    their two answers within 1e-4 relative of each other."""
    b = pu.bridge_for(W, scene)
    cpu = oracle_lib.OracleRenderer()
    cpu.buildPipeline(1, 1)
    W.upload_scene(cpu, b, 8, 8)
    skip = None
    if scene in UNREACHABLE:   # brute force leaves out the triangles its overflowed fallback leaves hide from the traversal
        leaf_data = np.asarray(b.blas, np.float32).reshape(-1, 8).view(np.uint32)[:, 7].astype(np.int64)
        skip = (decoded_coverage(leaf_data[leaf_data != 0], len(b.mesh_topology) // 20) == 0).astype(np.uint8)
    bvh, brute = cpu.traceVsBruteForce(_random_rays(b, n_rays, 7), skip)
    hit = brute[:, 1] >= 0
    assert hit.sum() >= min(1000, n_rays // 4), "enough of the random rays must hit something for the comparison to mean anything"
    same = (bvh[:, 0].view(np.uint32) == brute[:, 0].view(np.uint32))
    # where the distances agree the hit is the same primitive, or another one at exactly that distance
    ident = same & (bvh[:, 1] == brute[:, 1]) & (bvh[:, 2] == brute[:, 2])
    tie = same & ~ident & (brute[:, 3] > 1)
    assert (ident | tie)[same].all()
    diff = ~same
    assert diff.mean() <= 2e-4, "%d of %d rays disagree with brute force" % (diff.sum(), n_rays)
    if diff.any():
        assert (bvh[diff, 0] >= brute[diff, 0]).all()          # brute force is the true minimum
        missed = diff & (bvh[:, 1] < 0)
        rel = np.abs(bvh[diff & ~missed, 0] - brute[diff & ~missed, 0]) / brute[diff & ~missed, 0]
        assert (rel < 1e-4).all() and missed.sum() <= max(2, n_rays // 20000)


# ------------------------------------------------------------------------------------------------ TLAS known answers
def cpu_build_tlas(W, boxes, translations=None):
    lib = ctypes.CDLL(W._build.build_scene())
    n = len(boxes)
    b6 = np.ascontiguousarray(np.asarray(boxes, np.float32).reshape(n, 6))
    xf = None
    if translations is not None:
        xf = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (n, 1))
        xf[:, 12:15] = np.asarray(translations, np.float32)          # column-major: column 3 = translation
    nodes = np.zeros((2 * n, 8), np.float32)
    order = np.zeros(n, np.uint32)
    n_nodes = ctypes.c_uint32()
    vp = ctypes.c_void_p
    lib.ms_build_tlas.argtypes = [vp, vp, ctypes.c_uint32, vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), vp]
    rc = lib.ms_build_tlas(b6.ctypes.data_as(vp), xf.ctypes.data_as(vp) if xf is not None else None, n, nodes.ctypes.data_as(vp),
                           nodes.shape[0], ctypes.byref(n_nodes), order.ctypes.data_as(vp))
    assert rc == 0
    nodes = nodes[:n_nodes.value]
    u = nodes.view(np.uint32)
    return [{"min": nodes[i, 0:3].tolist(), "skip": int(u[i, 3]), "max": nodes[i, 4:7].tolist(), "data": int(u[i, 7])}
            for i in range(len(nodes))], order.tolist()


def test_tlas_known_answer_axis_rule_is_not_longest_axis(W):
    """Four instances, all x in [0, 2]: I0 y[0,1] z[0,1], I1 y[0,1] z[7,8], I2 y[2,3] z[3,4], I3 y[2,3] z[5,6].

    subdivide(0, 4) (tlas.rs:58): node 0 box = [0,2] x [0,3] x [0,8] (:62-67), extent (2, 3, 8).  tlas.rs:76:
    `extent.y > extent.x` is TRUE -> axis 1, although z is the longest axis.  Centres (box min + max) * 0.5 (primitives.rs:52):
    y = 0.5, 0.5, 2.5, 2.5 -> the (stable) sort leaves [0,1,2,3] (:78-83).  mid = 2 (:85): left {I0,I1} = [0,2]x[0,1]x[0,8],
    d = (2,1,8), area = 2 (2*1 + 1*8 + 8*2) = 52 (primitives.rs:41-50), cost 52 * 2 = 104; right {I2,I3} = [0,2]x[2,3]x[3,6],
    d = (2,1,3), area 2 (2 + 3 + 6) = 22, cost 44.  44 > 104 is false -> no rotation (:98-103).
    (A longest-axis builder sorts by z: 0.5, 7.5, 3.5, 5.5 -> [0,2,3,1], halves {I0,I2} / {I3,I1}: another tree.)
      node 1 = subdivide(0,2): [0,2]x[0,1]x[0,8], extent (2,1,8): y > x false; z > x and z > y -> axis 2; z centres 0.5, 7.5 ->
               [0,1]; both halves area 2 (2+1+2) = 10 -> no rotation.  node 2 = leaf I0: data (0 << 3) | 1 = 1, skip 3 (:69-73);
               node 3 = leaf I1: data (1 << 3) | 1 = 9, skip 4; node 1 skip 4 (:110).
      node 4 = subdivide(2,2): [0,2]x[2,3]x[3,6], extent (2,1,3) -> axis 2; 3.5, 5.5 -> [2,3]; 10 vs 10.  node 5 = leaf I2:
               data (2 << 3) | 1 = 17, skip 6; node 6 = leaf I3: data 25, skip 7; node 4 skip 7; node 0 skip 7."""
    k = TLAS_KAT["axis_rule"]
    nodes, order = cpu_build_tlas(W, k["boxes"])
    assert order == k["order"] == [0, 1, 2, 3]
    assert nodes == k["nodes"]


def test_tlas_known_answer_equal_centres_keep_their_order(W):
    """Four instances, all y, z in [0, 1]: I0 x[3,5], I1 x[0,8], I2 x[2,6], I3 x[0,2]: centres x = 4, 4, 4, 1.

    Root [0,8]x[0,1]x[0,1], extent (8,1,1): y > x false, z > x false -> axis 0.  slice.sort_by is a STABLE sort (tlas.rs:79):
    I3 (1) first, then I0, I1, I2 in their original order: [3,0,1,2] (an unstable sort may permute the three ties and give
    another tree).  mid = 2: left {I3,I0} = x[0,5], d (5,1,1), area 2 (5 + 1 + 5) = 22, cost 44; right {I1,I2} = x[0,8], area
    2 (8 + 1 + 8) = 34, cost 68.  68 > 44 -> rotate_left(2) (:99): order [1,2,3,0], l_count = r_count = 2.
      node 1 = subdivide(0,2): I1, I2: x[0,8], axis 0, centres 4, 4 -> [1,2] stays; left {I1} 34 * 1, right {I2} x[2,6] area
               2 (4 + 1 + 4) = 18: 18 > 34 false.  node 2 = leaf I1 (data 1, skip 3), node 3 = leaf I2 (data 9, skip 4); skip 4.
      node 4 = subdivide(2,2): I3, I0: x[0,5], axis 0, centres 1, 4 -> [3,0]; areas 10, 10.  node 5 = leaf I3 (data 17, skip 6),
               node 6 = leaf I0 (data 25, skip 7); node 4 skip 7; node 0 skip 7."""
    k = TLAS_KAT["stable_ties"]
    nodes, order = cpu_build_tlas(W, k["boxes"])
    assert order == k["order"] == [1, 2, 3, 0]
    assert nodes == k["nodes"]


def test_tlas_known_answer_costlier_half_goes_first(W):
    """Three instances, x in [0,2]: I0 y[0,1] z[0,1], I1 y[2,3] z[3,4], I2 y[1,2] z[7,8].

    Root [0,2]x[0,3]x[0,8], extent (2,3,8) -> axis 1 (y > x).  y centres 0.5, 2.5, 1.5 -> [0,2,1].  mid = 3 / 2 = 1: left {I0},
    d (2,1,1), area 10, cost 10; right {I2,I1} = [0,2]x[1,3]x[3,8], d (2,2,5), area 2 (4 + 10 + 10) = 48, cost 96.  96 > 10 ->
    rotate_left(1): [2,1,0], l_count = 2, r_count = 1: the costlier half becomes the FIRST child (:98-103).
      node 1 = subdivide(0,2): I2, I1: extent (2,2,5): y > x is false (2 > 2), z > x and z > y -> axis 2; z centres 7.5, 3.5 ->
               sorted [1,2], so the order is now [1,2,0]; areas 10, 10 -> no rotation.  node 2 = leaf I1 [0,2]x[2,3]x[3,4]
               (data 1, skip 3), node 3 = leaf I2 [0,2]x[1,2]x[7,8] (data 9, skip 4); node 1 skip 4.
      node 4 = subdivide(2,1): leaf I0: data (2 << 3) | 1 = 17, skip 5; node 0 skip 5."""
    k = TLAS_KAT["costlier_half_first"]
    nodes, order = cpu_build_tlas(W, k["boxes"])
    assert order == k["order"] == [1, 2, 0]
    assert nodes == k["nodes"]


def test_tlas_known_answer_boxes_are_transformed(W):
    """Two unit cubes, the first translated by (4,0,0) (AABB::transform of its 8 corners, primitives.rs:55-76, tlas.rs:24-27):
    world boxes [4,5] and [0,1] on x.  Root [0,5]x[0,1]x[0,1] -> axis 0; centres 4.5, 0.5 -> [1,0]; areas 6, 6 -> no rotation.
    node 1 = leaf I1 (data 1, skip 2), node 2 = leaf I0 (data 9, skip 3); and a single instance is one leaf with skip 1."""
    k = TLAS_KAT["translated"]
    nodes, order = cpu_build_tlas(W, k["boxes"], k["translations"])
    assert order == k["order"] == [1, 0]
    assert nodes == k["nodes"]
    one, o1 = cpu_build_tlas(W, [[1, 2, 3, 4, 6, 8]])
    assert o1 == [0] and one == [{"min": [1.0, 2.0, 3.0], "skip": 1, "max": [4.0, 6.0, 8.0], "data": 1}]
