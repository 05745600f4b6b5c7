"""Drive the HIP renderer and the CPU oracle with the same call sequence and compare every
parity artefact of SURVEY.md §8(d): accumulation buffer, G-buffer planes, history, RGBA8 output,
uniform block and the deterministic ray counters."""
import numpy as np

DIAMOND_OBJ = ("v 0.0 1.0 0.0\nv 1.0 0.0 0.0\nv 0.0 0.0 1.0\nv -1.0 0.0 0.0\nv 0.0 0.0 -1.0\nv 0.0 -1.0 0.0\n"
               "f 1 3 2\nf 1 2 5\nf 1 5 4\nf 1 4 3\nf 6 2 3\nf 6 5 2\nf 6 4 5\nf 6 3 4\n")

_bridges = {}


def diamond_obj_subdivided(n=11):
    """BASELINE.json words config 2 as a "~1k tris" diamond; public/diamond.obj is the 8-triangle octahedron.  Config 2b:
    the same octahedron with every face cut into n x n congruent triangles (n = 11: 968 triangles, 486 vertices on the
    flat faces — same shape, a deeper BLAS), as OBJ text for the `viewer` scene."""
    base_v = [(0, 1, 0), (1, 0, 0), (0, 0, 1), (-1, 0, 0), (0, 0, -1), (0, -1, 0)]
    base_f = [(1, 3, 2), (1, 2, 5), (1, 5, 4), (1, 4, 3), (6, 2, 3), (6, 5, 2), (6, 4, 5), (6, 3, 4)]
    verts, index, faces = [], {}, []

    def vid(p):
        key = tuple(round(c * n) for c in p)          # barycentric lattice points are exact multiples of 1/n
        if key not in index:
            index[key] = len(verts) + 1
            verts.append(tuple(k / n for k in key))
        return index[key]

    for fa, fb, fc in base_f:
        a, b, c = (np.array(base_v[i - 1], dtype=np.float64) for i in (fa, fb, fc))
        grid = {}
        for i in range(n + 1):
            for j in range(n + 1 - i):
                grid[i, j] = vid(a + (b - a) * (i / n) + (c - a) * (j / n))
        for i in range(n):
            for j in range(n - i):
                faces.append((grid[i, j], grid[i + 1, j], grid[i, j + 1]))
                if j < n - i - 1:
                    faces.append((grid[i + 1, j], grid[i + 1, j + 1], grid[i, j + 1]))
    return "".join("v %.9g %.9g %.9g\n" % v for v in verts) + "".join("f %d %d %d\n" % f for f in faces)


def bridge_for(pkg, scene):
    if scene not in _bridges:
        b = pkg.WorldBridge()
        if scene == "viewer_diamond":
            b.loadScene("viewer", DIAMOND_OBJ)
        elif scene == "viewer_diamond_1k":
            b.loadScene("viewer", diamond_obj_subdivided())
        else:
            b.loadScene(scene)
        _bridges[scene] = b
    return _bridges[scene]


def drive(renderer, pkg, bridge, w, h, depth, spp, frames, present=True, detailed=True):
    renderer.buildPipeline(depth, spp)
    pkg.upload_scene(renderer, bridge, w, h)
    if hasattr(renderer, "setCounting"):
        renderer.setCounting(detailed)
    renderer.resetCounters()
    for f in frames:
        renderer.compute(f)
        if present:
            renderer.present()
    renderer.sync()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({2: np.uint16, 4: np.uint32, 1: np.uint8, 8: np.uint64}[a.dtype.itemsize])


def describe_mismatch(name, a, b):
    ne = bits(a) != bits(b)
    idx = np.argwhere(ne)
    first = tuple(idx[0])
    return "%s differs at %d of %d elements; first at %s: gpu=%r oracle=%r" % (
        name, int(ne.sum()), ne.size, first, a[first], b[first])


def assert_parity(gpu, cpu, check_output=True, check_counters=True):
    ga, ca = gpu.readAccum(), cpu.readAccum()
    assert np.array_equal(bits(ga), bits(ca)), describe_mismatch("accumulation buffer", ga, ca)
    if check_output:
        go, co = gpu.captureFrame()["data"], cpu.captureFrame()["data"]
        assert np.array_equal(go, co), describe_mismatch("RGBA8 output", go, co)
        gh, ch = gpu.readHistory(), cpu.readHistory()
        # +0 / -0 are the same value; compare with the sign of zero masked out
        gz, cz = gh.copy(), ch.copy()
        gz[gz == 0x8000] = 0
        cz[cz == 0x8000] = 0
        assert np.array_equal(gz, cz), describe_mismatch("history (rgba16f)", gh, ch)
    else:
        for name, g, c in zip(("albedo", "normal_id", "depth"), gpu.readGBuffer(), cpu.readGBuffer()):
            assert np.array_equal(bits(g), bits(c)), describe_mismatch("G-buffer " + name, g, c)
    assert np.array_equal(gpu.readUniforms(), cpu.readUniforms()), "uniform block differs"
    if check_counters:
        assert gpu.getCounters() == cpu.getCounters()
