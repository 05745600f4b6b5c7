"""Randomised scenes emitted directly in the bridge layout — a fuzzing input for GPU/oracle parity.
Everything the renderer consumes is produced here with numpy: triangles with all four material types, several
geometries with their own stackless BLAS (random leaf sizes 1..7, random tree shapes), instances with random
affine transforms, a stackless TLAS, light references, optional textures."""
import numpy as np


class Bridge:
    """Duck-typed WorldBridge: the ten arrays + camera."""
    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.hasNewData = self.hasNewGeometry = True

    @property
    def lightCount(self):
        return len(self.lights) // 2

    @property
    def textureCount(self):
        return 0 if self.textures is None else len(self.textures)

    def getTextureRGBA(self, i):
        return self.textures[i]

    def getTexture(self, i):
        """encoded form (what the reference's bridge hands out): PNG of the layer"""
        import webgpu_raytracer_amd as W
        return W.textures.encode_png(self.textures[i])

    def updateCamera(self, w, h):
        pass


def _build_bvh(boxes_min, boxes_max, order, rng, leaf_max, leaf_payload):
    """Stackless DFS-preorder BVH over items `order`; returns list of (min, max, skip, data) with local skips.
    leaf_payload(first, count) -> data word; leaves hold 1..leaf_max consecutive items of `order`."""
    nodes = []

    def rec(lo, hi):
        idx = len(nodes)
        nodes.append(None)
        mn = boxes_min[order[lo:hi]].min(axis=0)
        mx = boxes_max[order[lo:hi]].max(axis=0)
        n = hi - lo
        if n <= leaf_max and (n == 1 or rng.random() < 0.6):
            nodes[idx] = [mn, mx, 0, leaf_payload(lo, n)]
        else:
            axis = int(rng.integers(0, 3))
            sub = order[lo:hi]
            c = (boxes_min[sub, axis] + boxes_max[sub, axis])
            order[lo:hi] = sub[np.argsort(c, kind="stable")]
            mid = lo + int(rng.integers(1, n))
            nodes[idx] = [mn, mx, 0, 0]
            rec(lo, mid)
            rec(mid, hi)
        nodes[idx][2] = len(nodes)

    rec(0, len(order))
    return nodes


def _pack(nodes):
    out = np.zeros((len(nodes), 8), dtype=np.float32)
    u = out.view(np.uint32)
    for i, (mn, mx, skip, data) in enumerate(nodes):
        out[i, 0:3] = mn
        out[i, 4:7] = mx
        u[i, 3] = skip
        u[i, 7] = data
    return out.reshape(-1)


def make(seed, n_geoms=3, tris_per_geom=40, n_instances=6, with_textures=False, lens=0.0):
    rng = np.random.default_rng(seed)
    verts, norms, uvs, topo_rows, blas_all = [], [], [], [], []
    blas_offsets, geom_boxes, geom_tri_ranges = [], [], []
    node_off = 0
    n_tex = 2 if with_textures else 0
    for g in range(n_geoms):
        nt = int(tris_per_geom * rng.uniform(0.5, 1.5))
        centers = rng.uniform(-0.8, 0.8, size=(nt, 1, 3))
        tri = (centers + rng.normal(scale=0.18, size=(nt, 3, 3))).astype(np.float32)
        if nt > 4:
            tri[1] = tri[0]                      # duplicate triangle: exact t ties
            tri[2, 2] = tri[2, 1]                # degenerate (zero area) triangle
        v_off = sum(len(v) for v in verts)
        verts.append(tri.reshape(-1, 3))
        n = rng.normal(size=(nt * 3, 3))
        norms.append((n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32))
        uvs.append(rng.uniform(-1.5, 2.5, size=(nt * 3, 2)).astype(np.float32))
        bmin, bmax = tri.min(axis=1) - 1e-4, tri.max(axis=1) + 1e-4
        order = np.arange(nt)
        topo_start = sum(len(t) for t in topo_rows)
        nodes = _build_bvh(bmin, bmax, order, rng, 7, lambda first, count: ((first + topo_start) << 3) | count)
        for pos, t in enumerate(order):
            row = np.zeros(20, dtype=np.uint32)
            f = row.view(np.float32)
            row[0:3] = v_off + 3 * t + np.arange(3)
            row[3] = g
            mat = int(rng.choice([0, 0, 1, 2, 3] if pos % 9 else [3]))
            f[4:7] = rng.uniform(0.2, 1.0, 3) * (8.0 if mat == 3 else 1.0)
            f[7] = float(mat)
            f[8] = rng.uniform(0, 1) if mat == 1 else 0.0              # metallic
            f[9] = rng.choice([0.0, 0.004, 0.05, 0.3, 1.0])            # roughness (incl. < 0.01 "specular")
            f[10] = rng.uniform(1.1, 2.0)                              # ior
            tex = lambda p: float(rng.integers(0, n_tex)) if (n_tex and rng.random() < p) else -1.0
            f[12:16] = [tex(0.5), tex(0.3), tex(0.3), tex(0.2)]
            f[16:19] = rng.uniform(0, 2, 3) if rng.random() < 0.1 else 0.0   # emissive colour
            f[19] = -1.0
            topo_rows.append(row[None])
        blas_all.append(_pack(nodes))
        blas_offsets.append(node_off)
        node_off += len(nodes)
        geom_boxes.append((bmin.min(axis=0), bmax.max(axis=0)))
        geom_tri_ranges.append((topo_start, nt))
    topo = np.concatenate(topo_rows).reshape(-1)
    V = np.concatenate(verts)
    vertices = np.concatenate([V, np.ones((len(V), 1), np.float32)], axis=1).astype(np.float32).reshape(-1)
    N = np.concatenate(norms)
    normals = np.concatenate([N, np.zeros((len(N), 1), np.float32)], axis=1).astype(np.float32).reshape(-1)
    uv = np.concatenate(uvs).reshape(-1)

    # instances: random rotation * non-uniform scale + translation; inverse in float64 then rounded to f32
    inst_rows, wmin, wmax = [], [], []
    for i in range(n_instances):
        g = int(rng.integers(0, n_geoms))
        q = rng.normal(size=(3, 3))
        r, _ = np.linalg.qr(q)
        m = np.eye(4)
        m[:3, :3] = r @ np.diag(rng.uniform(0.3, 0.9, 3))
        m[:3, 3] = rng.uniform(-1.2, 1.2, 3)
        m32 = m.astype(np.float32)
        inv32 = np.linalg.inv(m32.astype(np.float64)).astype(np.float32)
        row = np.zeros(36, dtype=np.float32)
        row[0:16] = m32.T.reshape(-1)            # column-major
        row[16:32] = inv32.T.reshape(-1)
        row.view(np.uint32)[32:36] = [blas_offsets[g], 0, g, 0]
        inst_rows.append(row)
        lo, hi = geom_boxes[g]
        corners = np.array([[x, y, z, 1.0] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
        wc = (corners @ m32.astype(np.float64).T)[:, :3]
        wmin.append(wc.min(axis=0) - 1e-3)
        wmax.append(wc.max(axis=0) + 1e-3)
    wmin, wmax = np.array(wmin), np.array(wmax)
    order = np.arange(n_instances)
    tl = _build_bvh(wmin, wmax, order, rng, 1, lambda first, count: (first << 3) | 1)
    instances = np.concatenate([inst_rows[i] for i in order])          # sorted like the reference's TLAS build
    lights, draw = [], []
    for si, i in enumerate(order):
        g = int(inst_rows[i].view(np.uint32)[34])
        start, cnt = geom_tri_ranges[g]
        draw += [cnt * 3, 1, start * 3, si]
        rows = topo.reshape(-1, 20)[start:start + cnt]
        for k, rw in enumerate(rows):
            if rw[7:8].view(np.float32)[0] == 3.0:
                lights += [si, start + k]
    textures = None
    if with_textures:
        textures = [rng.integers(0, 256, size=(1024, 1024, 4), dtype=np.uint8) for _ in range(n_tex)]

    cam = np.zeros(24, dtype=np.float32)
    eye = np.array([0.0, 0.2, -4.2], dtype=np.float32)
    cam[0:3], cam[3] = eye, lens
    h = np.array([3.6, 0, 0], np.float32)
    v = np.array([0, 2.7, 0], np.float32)
    cam[4:7] = eye + np.array([0, 0, 3.0], np.float32) - h / 2 - v / 2
    cam[8:11], cam[12:15] = h, v
    cam[16:19], cam[20:23] = [1, 0, 0], [0, 1, 0]
    return Bridge(vertices=vertices, normals=normals, uvs=uv, mesh_topology=topo, tlas=_pack(tl),
                  blas=np.concatenate(blas_all), instances=instances, lights=np.array(lights, dtype=np.uint32),
                  draw_commands=np.array(draw, dtype=np.uint32), cameraData=cam, textures=textures)
