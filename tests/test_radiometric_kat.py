"""A radiometric known answer, independent of any restatement: a closed box whose ceiling and four walls are emitters of
radiance Le and whose floor is a Lambert surface of albedo rho.  Every path that leaves the floor ends on an emitter, so
the radiance leaving the floor is exactly rho * Le (irradiance pi * Le times the BRDF rho / pi), whatever the mix of
light sampling and BSDF sampling — provided the light pdf, the cosine pdf, the pick probability of a light triangle and the
two power-heuristic weights are mutually consistent.  The reference's estimator (Raytracer.wgsl:345-427, 656-728) passes
this if and only if those pieces are what they claim to be; a mis-restated pdf or weight in the oracle shows up as a bias
of the mean, far outside the sampling error.  (GPU = oracle bit for bit, tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

import parity_util as pu  # noqa: F401  (sys.path set-up)
import random_scene


_QUAD = lambda a, b, c, d: [(a, b, c), (a, c, d)]
_BOX_FACES = {   # faces of the box [-1, 1]^3, wound so that cross(e1, e2) points INTO the box
    "floor":   _QUAD((-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1)),
    "ceiling": _QUAD((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)),
    "x-":      _QUAD((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1)),
    "x+":      _QUAD((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)),
    "z-":      _QUAD((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)),
    "z+":      _QUAD((-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1)),
}


def bridge_from_triangles(tris, mats, colours, iors=None, transform=None, eye=(0.0, 0.5, 0.0)):
    """One geometry, one instance, in the bridge layout.  tris: (n, 3, 3); mats: 0 Lambert / 2 dielectric / 3 emitter;
    colours: (n, 3) albedo or emitted radiance.  The camera sits at `eye` (object space) and looks straight down at a
    0.8 x 0.8 window one unit below it; `transform` (4x4) places geometry and camera together."""
    tris = np.asarray(tris, dtype=np.float32)
    nt = len(tris)
    rng = np.random.default_rng(1)
    bmin, bmax = tris.min(axis=1) - 1e-4, tris.max(axis=1) + 1e-4
    order = np.arange(nt)
    nodes = random_scene._build_bvh(bmin, bmax, order, rng, 4, lambda first, count: (first << 3) | count)
    rows = np.zeros((nt, 20), dtype=np.uint32)
    f = rows.view(np.float32)
    lights = []
    for pos, t in enumerate(order):
        rows[pos, 0:3] = 3 * t + np.arange(3)
        f[pos, 4:7] = colours[t]
        f[pos, 7] = float(mats[t])
        f[pos, 9] = 1.0
        f[pos, 10] = 1.5 if iors is None else iors[t]
        f[pos, 12:16] = -1.0
        f[pos, 19] = -1.0
        if mats[t] == 3:
            lights += [0, pos]
    V = tris.reshape(-1, 3)
    vertices = np.concatenate([V, np.ones((len(V), 1), np.float32)], axis=1).reshape(-1)
    N = np.repeat(np.array([np.cross(t[1] - t[0], t[2] - t[0]) for t in tris], dtype=np.float32), 3, axis=0)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    normals = np.concatenate([N, np.zeros((len(N), 1), np.float32)], axis=1).astype(np.float32).reshape(-1)
    M = np.eye(4) if transform is None else np.asarray(transform, dtype=np.float64)
    m32 = M.astype(np.float32)
    inv32 = np.linalg.inv(m32.astype(np.float64)).astype(np.float32)
    inst = np.zeros(36, dtype=np.float32)
    inst[0:16] = m32.T.reshape(-1)                        # column-major, like the bridge
    inst[16:32] = inv32.T.reshape(-1)
    lo, hi = V.min(axis=0), V.max(axis=0)
    corners = np.array([[x, y, z, 1.0] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])]) @ M.T
    tl = random_scene._pack([[(corners[:, :3].min(axis=0) - 1e-3).astype(np.float32),
                              (corners[:, :3].max(axis=0) + 1e-3).astype(np.float32), 1, (0 << 3) | 1]])
    cam = np.zeros(24, dtype=np.float32)
    A, t0 = M[:3, :3], M[:3, 3]
    e = (A @ np.array(eye, dtype=np.float64) + t0).astype(np.float32)
    h, v = (A @ np.array([0.8, 0, 0])).astype(np.float32), (A @ np.array([0, 0, 0.8])).astype(np.float32)
    cam[0:3] = e
    cam[4:7] = e + (A @ np.array([0, -1.0, 0])).astype(np.float32) - h / 2 - v / 2
    cam[8:11], cam[12:15] = h, v
    cam[16:19], cam[20:23] = h / np.linalg.norm(h), v / np.linalg.norm(v)   # forward = u x v: looking down
    return random_scene.Bridge(vertices=vertices, normals=normals, uvs=np.zeros(2 * len(V), np.float32),
                               mesh_topology=rows.reshape(-1), tlas=tl, blas=random_scene._pack(nodes), instances=inst,
                               lights=np.array(lights, dtype=np.uint32),
                               draw_commands=np.array([nt * 3, 1, 0, 0], dtype=np.uint32), cameraData=cam, textures=None)


def furnace_floor_bridge(rho, le, emitters_face_inward=True, emitting=("ceiling", "x-", "x+", "z-", "z+"), transform=None):
    """Box [-1, 1]^3: Lambert floor of albedo rho, the faces named in `emitting` are emitters of radiance le, the others
    are black (Lambert, albedo 0).  `transform` (4x4, float64) places the box as an instance; the camera moves with it."""
    tris, mats, colours = [], [], []
    for name, ts in _BOX_FACES.items():
        for t in ts:
            t = np.array(t, dtype=np.float32)
            n = np.cross(t[1] - t[0], t[2] - t[0])
            assert np.dot(n, -t.mean(axis=0)) > 0, name          # inward
            if name in emitting and not emitters_face_inward:
                t = t[[0, 2, 1]]
            tris.append(t)
            mats.append(3 if name in emitting else 0)
            colours.append(le if name in emitting else (rho if name == "floor" else np.zeros(3, np.float32)))
    return bridge_from_triangles(tris, mats, colours, transform=transform)


def _mean_radiance(W, oracle_lib, b, w, h, n_frames):
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, 8, 1, tuple(range(1, n_frames + 1)), present=False)
    acc = cpu.readAccum().astype(np.float64)
    assert (acc[..., 3] == n_frames).all()
    return acc[..., :3] / acc[..., 3:4], cpu.getCounters()


def _placed():
    """rotation about a skew axis x non-uniform scale + translation: light areas, cosines and distances all change"""
    ax = np.array([0.3, 1.0, -0.5]); ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(0.7) * K + (1 - np.cos(0.7)) * (K @ K)
    M = np.eye(4)
    M[:3, :3] = R * 1.7                                    # uniform scale keeps rho * Le the exact answer (angles are preserved)
    M[:3, 3] = [3.0, -2.0, 5.0]
    return M


@pytest.mark.parametrize("inward,transform", [(True, None), (False, None), (True, "placed")])
def test_lambert_floor_in_an_emitting_box(W, oracle_lib, inward, transform):
    """Mean radiance of the floor = rho * Le per channel.  inward = True: light sampling and BSDF sampling both reach the
    emitters and are combined by the power heuristic; False: the emitters' geometric normals point outward, the one-sided
    light pdf (Raytracer.wgsl:372-375) is zero everywhere and BSDF sampling alone must give the same answer."""
    rho = np.array([128, 204, 51], dtype=np.float32) / np.float32(255)     # exact in the G-buffer's rgba8unorm albedo
    le = np.array([2.0, 1.0, 0.5], dtype=np.float32)
    b = furnace_floor_bridge(rho, le, inward, transform=_placed() if transform else None)
    w = h = 48
    frames = tuple(range(1, 65))
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, w, h, 8, 1, frames, present=False)
    acc = cpu.readAccum().astype(np.float64)
    assert (acc[..., 3] == len(frames)).all()
    per_pixel = acc[..., :3] / acc[..., 3:4]
    mean = per_pixel.mean(axis=(0, 1))
    stderr = per_pixel.std(axis=(0, 1)) / np.sqrt(w * h)
    expected = rho.astype(np.float64) * le
    assert np.all(np.abs(mean - expected) < np.maximum(6 * stderr, 2e-3 * expected)), (mean, expected, stderr)
    assert np.all(np.abs(mean / expected - 1.0) < 0.01), (mean, expected)
    c = cpu.getCounters()
    assert c["primary_rays"] == w * h * len(frames)
    if inward:
        assert c["shadow_rays"] > 0.9 * c["primary_rays"]
    else:
        assert c["shadow_rays"] == 0


def _corner_form_factor(a, b, c):
    """differential area -> parallel rectangle a x b at distance c, the element under one CORNER of the rectangle (the
    classic closed form, Howell's catalogue of configuration factors B-4):
    F = 1 / (2 pi) [ a / sqrt(a^2 + c^2) atan(b / sqrt(a^2 + c^2)) + b / sqrt(b^2 + c^2) atan(a / sqrt(b^2 + c^2)) ]"""
    a, b = np.abs(a), np.abs(b)
    ra, rb = np.sqrt(a * a + c * c), np.sqrt(b * b + c * c)
    return (a / ra * np.arctan(b / ra) + b / rb * np.arctan(a / rb)) / (2 * np.pi)


def _ceiling_form_factor(x, z):
    """floor point (x, -1, z) -> the ceiling square [-1, 1]^2 at height 2: four corner rectangles"""
    c = 2.0
    return sum(_corner_form_factor(1 - sx * x, 1 - sz * z, c) for sx in (-1, 1) for sz in (-1, 1))   # |x|, |z| < 1


def test_lambert_floor_under_a_square_light(W, oracle_lib):
    """Only the ceiling emits; the walls are black.  The floor's radiance is rho * Le * F(x), F the point-to-rectangle
    form factor — direct light only, so this pins the light-sampling estimator (pick probability, area pdf, cosine at
    the light, shadow-ray visibility) and its MIS partner against geometry worked out on paper."""
    rho = np.array([255, 153, 102], dtype=np.float32) / np.float32(255)
    le = np.array([3.0, 2.0, 1.0], dtype=np.float32)
    b = furnace_floor_bridge(rho, le, True, emitting=("ceiling",))
    w = h = 48
    per_pixel, c = _mean_radiance(W, oracle_lib, b, w, h, 96)
    # floor point behind every pixel centre (bench camera convention: u = (x + .5) / w, v = 1 - (y + .5) / h)
    cam = b.cameraData.astype(np.float64)
    eye, ll, hv, vv = cam[0:3], cam[4:7], cam[8:11], cam[12:15]
    xs, ys = np.meshgrid((np.arange(w) + 0.5) / w, 1.0 - (np.arange(h) + 0.5) / h)
    d = ll[None, None, :] + xs[..., None] * hv + ys[..., None] * vv - eye
    t = (-1.0 - eye[1]) / d[..., 1]
    px, pz = eye[0] + t * d[..., 0], eye[2] + t * d[..., 2]
    F = _ceiling_form_factor(px, pz)
    assert 0.15 < F.min() and F.max() < 0.25          # ~0.24 under the centre of the light
    expected = F[..., None] * (rho.astype(np.float64) * le)[None, None, :]
    ratio = per_pixel.mean(axis=(0, 1)) / expected.mean(axis=(0, 1))
    stderr = per_pixel.std(axis=(0, 1)) / np.sqrt(w * h) / expected.mean(axis=(0, 1))
    assert np.all(np.abs(ratio - 1.0) < np.maximum(5 * stderr, 3e-3)), (ratio, stderr)
    # and pixel by pixel, within the noise of 96 samples: no systematic tilt across the image
    rel = (per_pixel - expected) / expected
    assert np.abs(rel[: h // 2].mean() - rel[h // 2:].mean()) < 0.02
    assert np.abs(rel[:, : w // 2].mean() - rel[:, w // 2:].mean()) < 0.02
    assert c["shadow_rays"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["box", "box_outward", "box_placed", "square_light", "glass_slab", "metal_floor"])
def test_gpu_equals_oracle_on_the_radiometric_scenes(W, oracle_lib, gpu_renderer, case):
    """The same scenes on the HIP path: bit-identical to the oracle (so the known answers above hold for it too)."""
    rho = np.array([128, 204, 51], dtype=np.float32) / np.float32(255)
    le = np.array([2.0, 1.0, 0.5], dtype=np.float32)
    b = {"box": lambda: furnace_floor_bridge(rho, le, True),
         "box_outward": lambda: furnace_floor_bridge(rho, le, False),
         "box_placed": lambda: furnace_floor_bridge(rho, le, True, transform=_placed()),
         "square_light": lambda: furnace_floor_bridge(rho, le, True, emitting=("ceiling",)),
         "glass_slab": lambda: glass_slab_bridge(le),
         "metal_floor": lambda: metal_floor_bridge(0.3)[0]}[case]()
    frames = tuple(range(1, 9))
    cpu = oracle_lib.OracleRenderer()
    pu.drive(cpu, W, b, 48, 48, 8, 1, frames, present=True)
    pu.drive(gpu_renderer, W, b, 48, 48, 8, 1, frames, present=True)
    pu.assert_parity(gpu_renderer, cpu, check_output=True)


def glass_slab_bridge(le, ior=1.5):
    """The box with an emitting ceiling, black walls and floor, and a white glass slab (y in [-0.2, 0]) spanning it: the
    camera (y = 0.5) looks down at the slab; what it sees of the ceiling is the slab's reflectance."""
    tris, mats, colours, iors = [], [], [], []
    for name, ts in _BOX_FACES.items():
        for t in ts:
            tris.append(np.array(t, dtype=np.float32))
            mats.append(3 if name == "ceiling" else 0)
            colours.append(le if name == "ceiling" else np.zeros(3, np.float32))
            iors.append(1.5)
    for y, up in ((0.0, True), (-0.2, False)):
        q = _QUAD((-1, y, -1), (-1, y, 1), (1, y, 1), (1, y, -1)) if up else _QUAD((-1, y, -1), (1, y, -1), (1, y, 1), (-1, y, 1))
        for t in q:
            t = np.array(t, dtype=np.float32)
            assert (np.cross(t[1] - t[0], t[2] - t[0])[1] > 0) == up      # outward normals of the slab
            tris.append(t)
            mats.append(2)
            colours.append(np.ones(3, np.float32))
            iors.append(ior)
    return bridge_from_triangles(tris, mats, colours, iors)


def test_glass_slab_reflectance(W, oracle_lib):
    """A white dielectric slab over a black floor under an emitting ceiling shows Le times its reflectance.  With the
    reference's Schlick reflectance at BOTH interfaces (Raytracer.wgsl:314-339; the cosine is the one on the incident
    side, inside the glass too) the geometric series of internal reflections sums to
    R = R1 + (1 - R1) R2 / (1 + R2), R1 = schlick(cos theta), R2 = schlick(cos theta') with Snell's theta'.
    Pins the branch probability of the dielectric (reflect with probability R, one RNG draw), its unit throughput, the
    refraction direction (theta' enters R2) and that glass blocks the floor's shadow rays."""
    le = np.array([4.0, 2.0, 1.0], dtype=np.float32)
    ior = 1.5
    b = glass_slab_bridge(le, ior)
    w = h = 48
    per_pixel, c = _mean_radiance(W, oracle_lib, b, w, h, 256)
    cam = b.cameraData.astype(np.float64)
    eye, ll, hv, vv = cam[0:3], cam[4:7], cam[8:11], cam[12:15]
    xs, ys = np.meshgrid((np.arange(w) + 0.5) / w, 1.0 - (np.arange(h) + 0.5) / h)
    d = ll[None, None, :] + xs[..., None] * hv + ys[..., None] * vv - eye
    cos_t = -d[..., 1] / np.linalg.norm(d, axis=-1)
    sin_in = np.sqrt(1 - cos_t ** 2) / ior
    cos_in = np.sqrt(1 - sin_in ** 2)
    r0 = ((1 - ior) / (1 + ior)) ** 2
    schlick = lambda cs: r0 + (1 - r0) * (1 - cs) ** 5
    R1, R2 = schlick(cos_t), schlick(cos_in)
    R = R1 + (1 - R1) * R2 / (1 + R2)
    assert 0.0769 < R.min() and R.max() < 0.0771          # 2 r0 / (1 + r0) = 0.076923 at these angles
    expected = R[..., None] * le.astype(np.float64)[None, None, :]
    ratio = per_pixel.mean(axis=(0, 1)) / expected.mean(axis=(0, 1))
    stderr = per_pixel.std(axis=(0, 1)) / np.sqrt(w * h) / expected.mean(axis=(0, 1))
    assert np.all(stderr < 0.01)
    assert np.all(np.abs(ratio - 1.0) < 5 * stderr), (ratio, stderr)
    assert c["shadow_rays"] > 0      # the floor tries next-event estimation; the slab is in the way


def _ggx_weight_integral(view_dirs, alpha, f0, rng, n_samples):
    """E over pixels and half vectors of the reference's GGX sampling weight (Raytracer.wgsl:271-306), evaluated
    independently in float64: h ~ D(h) (n.h) around n = +y, l = reflect(-v, h) with the UN-normalised view vector v the
    reference passes (-ray.direction), weight = F(v.h) G1(n.v) G1(n.l) (v.h) / ((n.v)(n.h)), zero when l dips below the
    surface.  Returns the mean weight per colour channel."""
    a2 = alpha * alpha
    v = view_dirs[rng.integers(0, len(view_dirs), n_samples)]                  # (N, 3), n = +y
    u = rng.random((n_samples, 2))
    phi = 2 * np.pi * u[:, 0]
    ct = np.sqrt(np.maximum(0, (1 - u[:, 1]) / (1 + (a2 - 1) * u[:, 1])))
    st = np.sqrt(np.maximum(0, 1 - ct * ct))
    h = np.stack([st * np.cos(phi), ct, st * np.sin(phi)], axis=1)
    vdh_raw = np.einsum("ij,ij->i", v, h)
    l = -v + 2 * vdh_raw[:, None] * h
    ok = l[:, 1] > 0
    ndv, ndl = np.maximum(v[:, 1], 1e-4), np.maximum(l[:, 1], 1e-4)
    ndh, vdh = np.maximum(h[:, 1], 1e-4), np.maximum(vdh_raw, 1e-4)
    g1 = lambda x: 2 * x / (x + np.sqrt(a2 + (1 - a2) * x * x))
    fres = f0[None, :] + (1 - f0[None, :]) * (np.clip(1 - vdh, 0, 1) ** 5)[:, None]
    w = np.where(ok, g1(ndv) * g1(ndl) * vdh / (ndv * ndh), 0.0)
    return (fres * w[:, None]).mean(axis=0)


def metal_floor_bridge(roughness):
    """A GGX floor (metallic 1, f0 = albedo) in the emitting box with the emitters turned away."""
    le = np.array([2.0, 1.0, 0.5], dtype=np.float32)
    f0 = np.array([255, 204, 128], dtype=np.float32) / np.float32(255)
    tris, mats, cols = [], [], []
    for name, ts in _BOX_FACES.items():
        for t in ts:
            t = np.array(t, dtype=np.float32)
            if name != "floor":
                t = t[[0, 2, 1]]                       # emitters face outward: the one-sided light pdf is zero
            tris.append(t)
            mats.append(1 if name == "floor" else 3)
            cols.append(f0 if name == "floor" else le)
    b = bridge_from_triangles(tris, mats, cols)
    rows = b.mesh_topology.reshape(-1, 20).view(np.float32)
    rows[rows[:, 7] == 1.0, 8] = 1.0                    # metallic 1: f0 = albedo
    rows[rows[:, 7] == 1.0, 9] = roughness
    return b, le, f0


@pytest.mark.parametrize("roughness", [0.05, 0.3])
def test_metal_floor_against_an_independent_integral(W, oracle_lib, roughness):
    """A GGX floor in the emitting box, emitters turned away so that BSDF sampling is the only strategy: the floor shows
    Le times the mean sampling weight.  That mean is NOT 1 — the reference's GGX loses energy (12 % at roughness 0.3) and
    takes an un-normalised view vector — so it is computed here from the formulas of Raytracer.wgsl:236-306 on their own,
    in float64 numpy, and the oracle's path tracer must land on it: pins sample_ggx, build_onb / reflect and the pickup of
    emission after a non-specular bounce."""
    b, le, f0 = metal_floor_bridge(roughness)
    w = h = 48
    per_pixel, c = _mean_radiance(W, oracle_lib, b, w, h, 64)
    assert c["shadow_rays"] == 0
    cam = b.cameraData.astype(np.float64)
    eye, ll, hv, vv = cam[0:3], cam[4:7], cam[8:11], cam[12:15]
    xs, ys = np.meshgrid((np.arange(w) + 0.5) / w, 1.0 - (np.arange(h) + 0.5) / h)
    d = (ll[None, None, :] + xs[..., None] * hv + ys[..., None] * vv - eye).reshape(-1, 3)
    weight = _ggx_weight_integral(-d, roughness, f0.astype(np.float64), np.random.default_rng(7), 4_000_000)
    ratio = per_pixel.mean(axis=(0, 1)) / (weight * le)
    stderr = per_pixel.std(axis=(0, 1)) / np.sqrt(w * h) / (weight * le)
    assert (0.85 < weight / f0).all() and (weight / f0 < 1.01).all()
    assert np.all(np.abs(ratio - 1.0) < np.maximum(5 * stderr, 2.5e-3)), (ratio, stderr, weight)


def test_textured_emitters_scale_the_known_answer(W, oracle_lib):
    """The emitting box again, every emitter carrying a base-colour texture of constant value 128 / 255: light sampling
    reads it at the sampled point (Raytracer.wgsl:383-389), a BSDF-sampled hit reads it through the albedo — both must
    scale by exactly 128 / 255, so the floor shows rho * Le * 128 / 255.  Pins the texel decode (rgba8unorm -> f32), the
    uv interpolation feeding it and that both strategies see the same emitter."""
    rho = np.array([128, 204, 51], dtype=np.float32) / np.float32(255)
    le = np.array([2.0, 1.0, 0.5], dtype=np.float32)
    b = furnace_floor_bridge(rho, le, True)
    rows = b.mesh_topology.reshape(-1, 20).view(np.float32)
    rows[rows[:, 7] == 3.0, 12] = 0.0                   # base-colour texture 0 on the emitters
    b.uvs = np.random.default_rng(2).uniform(0.05, 0.95, size=len(b.uvs)).astype(np.float32)
    b.textures = [np.full((1024, 1024, 4), 128, dtype=np.uint8)]
    per_pixel, c = _mean_radiance(W, oracle_lib, b, 48, 48, 64)
    expected = rho.astype(np.float64) * le * (128.0 / 255.0)
    ratio = per_pixel.mean(axis=(0, 1)) / expected
    stderr = per_pixel.std(axis=(0, 1)) / 48 / expected
    assert np.all(np.abs(ratio - 1.0) < np.maximum(5 * stderr, 1e-3)), (ratio, stderr)
