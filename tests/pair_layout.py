"""numpy restatement of the child-pair re-layout of the node arrays (webgpu-raytracer_amd/csrc/k_pairs.hip.h builds
the same arrays on the GPU at upload time): test infrastructure.  From the bridge's TLAS / BLAS arrays
(bvh/mod.rs StacklessBVHNode: pre-order, inner node's first child = next element, skip pointers) it makes

  pairs      (n_inner, 16) f32   one record per INNER node X, in array order: {L.min, wordL} {L.max, 0} {R.min, wordR}
                                 {R.max, skipX};  L = X + 1, R = the node L's skip pointer names;
                                 word(C) = 0x80000000 | index of C's own record (C inner) or C's leaf word;
                                 skipX = index of the record whose RIGHT child follows X's subtree in pre-order, or
                                 0xffffffff when that leaves the TLAS / the BLAS
  troot      (8,) f32            {TLAS root.min, word(root)} {TLAS root.max, 0}
  inst_root  (n_inst, 8) f32     the same for the BLAS root of every instance
"""
import numpy as np

INNER = 0x80000000
END = 0xFFFFFFFF


def build(tlas, blas, instances):
    tl = np.asarray(tlas, np.float32).reshape(-1, 8)
    bl = np.asarray(blas, np.float32).reshape(-1, 8)
    nodes = np.concatenate([tl, bl]) if len(bl) else tl.copy()
    u = nodes.view(np.uint32)
    n, n_tlas = len(nodes), len(tl)
    skip = u[:, 3].astype(np.int64)
    data = u[:, 7].astype(np.int64)
    inst = np.asarray(instances, np.float32).reshape(-1, 36)
    offs = inst.view(np.uint32)[:, 32].astype(np.int64)
    roots = np.unique(offs)
    # level of every node: (first node, end) of the array it walks in
    start = np.zeros(n, np.int64)
    end = np.zeros(n, np.int64)
    if n_tlas:
        start[:n_tlas] = 0
        end[:n_tlas] = skip[0]
    if len(bl):
        local = np.arange(n - n_tlas)
        r = roots[np.clip(np.searchsorted(roots, local, side="right") - 1, 0, len(roots) - 1)]
        start[n_tlas:] = n_tlas + r
        end[n_tlas:] = n_tlas + r + skip[n_tlas + r]
    is_tlas = np.arange(n) < n_tlas
    target = np.where(is_tlas, skip, start + skip)           # absolute index of the node the skip pointer names
    inner = data == 0
    pair_of = np.cumsum(inner) - 1                            # record index of an inner node
    ids = np.nonzero(inner)[0]
    left = ids + 1
    right = target[np.minimum(left, n - 1)]
    ok = (left < n) & (right < n) & (right > left)            # nodes no instance reaches may hold anything: kept in range
    left = np.where(ok, left, 0)
    right = np.where(ok, right, 0)
    parent = np.full(n, -1, np.int64)
    parent[left[ok]] = ids[ok]
    parent[right[ok]] = ids[ok]

    def word(c):
        return np.where(inner[c], INNER | pair_of[c], data[c]).astype(np.uint32)

    succ = target[ids]
    off_level = (succ >= end[ids]) | (succ >= n)
    sp = parent[np.minimum(succ, n - 1)]
    skipx = np.where(off_level | (sp < 0), END, pair_of[np.maximum(sp, 0)]).astype(np.uint32)
    pairs = np.zeros((len(ids), 16), np.float32)
    pu = pairs.view(np.uint32)
    pairs[:, 0:3] = nodes[left, 0:3]
    pu[:, 3] = word(left)
    pairs[:, 4:7] = nodes[left, 4:7]
    pairs[:, 8:11] = nodes[right, 0:3]
    pu[:, 11] = word(right)
    pairs[:, 12:15] = nodes[right, 4:7]
    pu[:, 15] = skipx

    def root_rec(i):
        out = np.zeros(8, np.float32)
        out[0:3] = nodes[i, 0:3]
        out.view(np.uint32)[3] = word(np.array([i]))[0]
        out[4:7] = nodes[i, 4:7]
        return out

    troot = root_rec(0) if n_tlas else np.zeros(8, np.float32)
    inst_root = np.stack([root_rec(n_tlas + o) for o in offs]) if len(offs) else np.zeros((0, 8), np.float32)
    return pairs, troot, inst_root


def traversal_records(bridge):
    """tri_geom (n_tris, 12): {v0, 0} {e1, 0} {e2, 0}; inst_trav (n_inst, 16): rows 0..2 of the inverse, then a tail row —
    k_prepare_tris / k_prepare_instances of the HIP library in numpy (the f32 subtractions the shader does per test)."""
    pos = np.asarray(bridge.vertices, np.float32).reshape(-1, 4)[:, :3]
    topo = np.asarray(bridge.mesh_topology, np.uint32).reshape(-1, 20)
    v0, v1, v2 = (pos[topo[:, k].astype(np.int64)] for k in range(3))
    tri = np.zeros((len(topo), 12), np.float32)
    tri[:, 0:3] = v0
    tri[:, 4:7] = v1 - v0
    tri[:, 8:11] = v2 - v0
    inst = np.asarray(bridge.instances, np.float32).reshape(-1, 36)
    inv = inst[:, 16:32].reshape(-1, 4, 4)                  # column-major: inv[i, c, r]
    it = np.zeros((len(inst), 16), np.float32)
    for r in range(3):
        it[:, 4 * r:4 * r + 4] = inv[:, :, r]
    return tri, it
