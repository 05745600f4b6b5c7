"""Texture ingest (reference: ResourceManager.ts:153-198).

CPU: the host decoders (include/mi355tex.h) against an independent decoder (PIL: libpng / libjpeg-turbo) — PNG is
lossless so any correct decoder agrees; for JPEG the integer IDCT, triangle upsampling and fixed-point colour
conversion are the published IJG algorithms, so the bytes agree as well (tolerance 0).  The resize rule (oracle
restatement) is checked for its properties.
GPU: k_resize_texture against the oracle, and loadTexturesFromWorld through decode + resize."""
import io
import struct
import zlib

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


@pytest.fixture()
def gpu_renderer(W):
    W._build.build_rt()
    r = W.WebGPURenderer(0)
    yield r
    r.destroy()


def smooth(rng, h, w, c):
    y, x = np.mgrid[0:h, 0:w]
    chans = [(127 + 100 * np.sin(x / (7.0 + k) + y / (11.0 - k)) + rng.integers(0, 20, (h, w))).clip(0, 255) for k in range(c)]
    return np.stack(chans, -1).astype(np.uint8)


def chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def mkpng(w, h, depth, ctype, rows, interlace=0, extra=b"", level=6):
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + extra +
            chunk(b"IDAT", zlib.compress(rows, level)) + chunk(b"IEND", b""))


def pil_rgba(blob):
    return np.asarray(PIL.open(io.BytesIO(blob)).convert("RGBA"))


@pytest.mark.parametrize("mode,c", [("L", 1), ("RGB", 3), ("RGBA", 4), ("LA", 2), ("P", 3)])
def test_png_colour_types_match_pil(W, mode, c):
    rng = np.random.default_rng(3)
    for (h, w) in ((1, 1), (5, 7), (33, 65), (64, 64)):
        a = smooth(rng, h, w, c)
        if mode == "P":
            im = PIL.fromarray(a, "RGB").quantize(17)
        else:
            im = PIL.fromarray(a[..., 0] if c == 1 else a, mode)
        for opt in (False, True):  # optimize=True: dynamic Huffman + all five filter types
            bio = io.BytesIO()
            im.save(bio, "PNG", optimize=opt, compress_level=9 if opt else 1)
            assert np.array_equal(W.textures.decode_image(bio.getvalue()), np.asarray(im.convert("RGBA")))


def test_png_low_bit_depths_16_bit_trns_and_stored_blocks(W):
    rng = np.random.default_rng(4)
    for depth in (1, 2, 4):
        w, h = 13, 5
        vals = rng.integers(0, 1 << depth, (h, w))
        rows = b""
        for y in range(h):
            bits = "".join(format(v, "0%db" % depth) for v in vals[y])
            bits += "0" * (-len(bits) % 8)
            rows += b"\x00" + int(bits, 2).to_bytes(len(bits) // 8, "big")
        png = mkpng(w, h, depth, 0, rows, level=0)   # level 0 = stored deflate blocks
        assert np.array_equal(W.textures.decode_image(png), pil_rgba(png))
    w, h = 9, 4
    v16 = rng.integers(0, 65536, (h, w, 3)).astype(">u2")
    rows = b"".join(b"\x00" + v16[y].tobytes() for y in range(h))
    key = v16[1, 2]
    got = W.textures.decode_image(mkpng(w, h, 16, 2, rows, extra=chunk(b"tRNS", key.tobytes())))
    want = np.full((h, w, 4), 255, np.uint8)
    want[..., :3] = v16.astype(np.uint16) >> 8          # the high byte
    want[(v16 == key).all(-1), 3] = 0                   # colour key compared at the file's bit depth
    assert np.array_equal(got, want)


def test_png_adam7_with_every_filter(W):
    rng = np.random.default_rng(5)
    w, h = 11, 9
    img = smooth(rng, h, w, 3)
    xs, ys, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    rows = b""
    for p in range(7):
        sub = img[ys[p]::dy[p], xs[p]::dx[p]]
        if sub.size == 0:
            continue
        prev = np.zeros(sub.shape[1] * 3, np.int32)
        for y in range(sub.shape[0]):
            cur = sub[y].reshape(-1).astype(np.int32)
            ft = (y + p) % 5
            a = np.concatenate([np.zeros(3, np.int32), cur[:-3]])
            c = np.concatenate([np.zeros(3, np.int32), prev[:-3]])
            if ft == 0:
                f = cur
            elif ft == 1:
                f = cur - a
            elif ft == 2:
                f = cur - prev
            elif ft == 3:
                f = cur - ((a + prev) >> 1)
            else:
                pp = a + prev - c
                pa, pb, pc = abs(pp - a), abs(pp - prev), abs(pp - c)
                f = cur - np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            rows += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
            prev = cur
    png = mkpng(w, h, 8, 2, rows, interlace=1)
    got = W.textures.decode_image(png)
    assert np.array_equal(got[..., :3], img) and np.array_equal(got, pil_rgba(png))


def test_corrupt_and_foreign_input_is_refused(W):
    rng = np.random.default_rng(6)
    bio = io.BytesIO()
    PIL.fromarray(smooth(rng, 20, 20, 3), "RGB").save(bio, "PNG")
    png = bio.getvalue()
    flipped = png[:60] + bytes([png[60] ^ 0x40]) + png[61:]
    for blob in (b"", b"abc", b"GIF89a" + b"\0" * 20, png[:40], png[:-20], flipped):
        with pytest.raises(W.textures.ImageDecodeError):
            W.textures.decode_image(blob)
    assert W.textures.probe(png) == 1 and W.textures.probe(b"\xff\xd8\xff\xe0") == 2 and W.textures.probe(b"BM") == 0


def test_inflate_matches_zlib(W):
    rng = np.random.default_rng(7)
    for data in (b"", b"a", bytes(1000), rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(),
                 (b"the quick brown fox " * 400)):
        for level in (0, 1, 6, 9):
            assert W.textures.inflate(zlib.compress(data, level), len(data)) == data
    with pytest.raises(W.textures.ImageDecodeError):
        W.textures.inflate(zlib.compress(b"x" * 100), 10)      # output larger than the caller allows
    bad = bytearray(zlib.compress(b"hello world" * 20))
    bad[-1] ^= 1                                                # Adler-32 mismatch
    with pytest.raises(W.textures.ImageDecodeError):
        W.textures.inflate(bytes(bad), 1000)


@pytest.mark.parametrize("progressive", [False, True])
def test_jpeg_matches_libjpeg_bit_for_bit(W, progressive):
    rng = np.random.default_rng(8)
    for (h, w) in ((8, 8), (17, 23), (64, 48), (100, 131), (1, 1), (3, 200)):
        for sub in (0, 1, 2):           # 4:4:4, 4:2:2, 4:2:0
            for q in (30, 90):
                bio = io.BytesIO()
                PIL.fromarray(smooth(rng, h, w, 3), "RGB").save(bio, "JPEG", quality=q, subsampling=sub, progressive=progressive)
                got = W.textures.decode_image(bio.getvalue())
                want = np.asarray(PIL.open(io.BytesIO(bio.getvalue())).convert("RGB"))
                assert np.array_equal(got[..., :3], want), (h, w, sub, q)
                assert (got[..., 3] == 255).all()
        bio = io.BytesIO()
        PIL.fromarray(smooth(rng, h, w, 1)[..., 0], "L").save(bio, "JPEG", quality=80, progressive=progressive)
        got = W.textures.decode_image(bio.getvalue())
        assert np.array_equal(got[..., 0], np.asarray(PIL.open(io.BytesIO(bio.getvalue())).convert("L")))


def test_jpeg_restart_intervals_and_optimised_tables(W):
    rng = np.random.default_rng(9)
    im = PIL.fromarray(smooth(rng, 70, 90, 3), "RGB")
    for kw in (dict(optimize=True, restart_marker_blocks=3), dict(restart_marker_rows=1), dict(optimize=True, subsampling=2)):
        bio = io.BytesIO()
        im.save(bio, "JPEG", quality=85, **kw)
        got = W.textures.decode_image(bio.getvalue())
        assert np.array_equal(got[..., :3], np.asarray(PIL.open(io.BytesIO(bio.getvalue())).convert("RGB")))


def test_bridge_textures_round_trip_through_png(W):
    b = W.WorldBridge()
    b.loadScene("sponza_like")
    assert b.textureCount == 8
    raw = b.getTextureRGBA(3)
    blob = b.getTexture(3)
    assert W.textures.probe(blob) == 1
    assert np.array_equal(W.textures.decode_image(blob), raw)
    assert np.array_equal(pil_rgba(blob), raw)


def oracle_resize(oracle_lib, img):
    out = np.empty((1024, 1024, 4), np.uint8)
    if img is None:
        oracle_lib.lib().oracle_resize_texture(None, 0, 0, out.ctypes.data)
    else:
        a = np.ascontiguousarray(img)
        oracle_lib.lib().oracle_resize_texture(a.ctypes.data, a.shape[1], a.shape[0], out.ctypes.data)
    return out


def test_resize_rule_properties(oracle_lib):
    rng = np.random.default_rng(10)
    full = rng.integers(0, 256, (1024, 1024, 4), dtype=np.uint8)
    assert np.array_equal(oracle_resize(oracle_lib, full), full)                 # same size: identity
    assert (oracle_resize(oracle_lib, None) == 255).all()                        # white fallback bitmap
    one = np.array([[[10, 20, 30, 40]]], np.uint8)
    assert (oracle_resize(oracle_lib, one) == one[0, 0]).all()                   # 1x1: constant
    small = rng.integers(0, 256, (2, 2, 4), dtype=np.uint8)
    up = oracle_resize(oracle_lib, small)
    assert np.array_equal(up[0, 0], small[0, 0]) and np.array_equal(up[-1, -1], small[1, 1])   # clamp to edge
    lo, hi = small.min(axis=(0, 1)), small.max(axis=(0, 1))
    assert (up >= lo).all() and (up <= hi).all()                                 # convex combination
    big = rng.integers(0, 256, (2048, 2048, 4), dtype=np.uint8)
    down = oracle_resize(oracle_lib, big).astype(np.int32)
    box = big.astype(np.int32).reshape(1024, 2, 1024, 2, 4).sum(axis=(1, 3))
    assert np.array_equal(down, (box + 2) // 4)                                  # exact 2:1: mean of the 2x2 block, ties up


@pytest.mark.gpu
def test_gpu_resize_matches_oracle(W, oracle_lib, gpu_renderer):
    rng = np.random.default_rng(11)
    shapes = [(1, 1), (2, 3), (7, 1024), (1024, 1024), (300, 517), (2048, 2048), (1500, 640), (4096, 31)]
    gpu_renderer._check(gpu_renderer.L.rt_alloc_texture_layers(gpu_renderer.ctx, len(shapes) + 1), "alloc")
    for i in range(len(shapes) + 1):
        assert (gpu_renderer.readTextureLayer(i) == 255).all()                   # layers start white
    for i, (h, w) in enumerate(shapes):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        gpu_renderer.uploadTextureImage(i, img)
        assert np.array_equal(gpu_renderer.readTextureLayer(i), oracle_resize(oracle_lib, img)), (h, w)
    gpu_renderer.uploadTextureImage(0, None)
    assert (gpu_renderer.readTextureLayer(0) == 255).all()
    with pytest.raises(W.RendererError):
        gpu_renderer.uploadTextureImage(len(shapes) + 1, np.zeros((4, 4, 4), np.uint8))


@pytest.mark.gpu
def test_load_textures_from_world_decodes_resizes_and_falls_back(W, oracle_lib, gpu_renderer):
    class Bridge:
        """what loadTexturesFromWorld needs of the bridge: textureCount + getTexture(i) (encoded bytes or None)"""
        def __init__(self, blobs):
            self.blobs = blobs
            self.textureCount = len(blobs)

        def getTexture(self, i):
            return self.blobs[i]

    rng = np.random.default_rng(12)
    a = smooth(rng, 200, 333, 4)
    b = smooth(rng, 512, 512, 3)
    png, jpg = io.BytesIO(), io.BytesIO()
    PIL.fromarray(a, "RGBA").save(png, "PNG")
    PIL.fromarray(b, "RGB").save(jpg, "JPEG", quality=90)
    gpu_renderer.loadTexturesFromWorld(Bridge([png.getvalue(), jpg.getvalue(), b"not an image", None]))
    assert np.array_equal(gpu_renderer.readTextureLayer(0), oracle_resize(oracle_lib, a))
    assert np.array_equal(gpu_renderer.readTextureLayer(1), oracle_resize(oracle_lib, pil_rgba(jpg.getvalue())))
    assert (gpu_renderer.readTextureLayer(2) == 255).all() and (gpu_renderer.readTextureLayer(3) == 255).all()
    assert len(gpu_renderer.texture_warnings) == 1 and "Failed tex 2" in gpu_renderer.texture_warnings[0]


@pytest.mark.gpu
def test_ingested_scene_textures_equal_direct_upload(W, gpu_renderer):
    """sponza_like through getTexture -> decode -> GPU resize gives the layers rt_upload_textures is given directly."""
    b = W.WorldBridge()
    b.loadScene("sponza_like")
    gpu_renderer.loadTexturesFromWorld(b)
    for i in range(b.textureCount):
        assert np.array_equal(gpu_renderer.readTextureLayer(i), b.getTextureRGBA(i))


def test_decoders_survive_mutated_input(W):
    """Byte-level mutations of valid PNG / JPEG files either decode or are refused — never crash (the sanitizer build of
    the same loop is tools/fuzz/run.sh: ASan + UBSan, 64 k mutations clean)."""
    rng = np.random.default_rng(13)
    seeds = []
    im = PIL.fromarray(smooth(rng, 24, 31, 3), "RGB")
    for kw in (dict(format="PNG"), dict(format="PNG", optimize=True), dict(format="JPEG", quality=70), dict(format="JPEG", progressive=True, subsampling=2)):
        bio = io.BytesIO()
        im.save(bio, **kw)
        seeds.append(bio.getvalue())
    decoded = refused = 0
    for seed in seeds:
        for _ in range(400):
            d = bytearray(seed)
            for _ in range(int(rng.integers(1, 6))):
                k = int(rng.integers(0, 4))
                p = int(rng.integers(0, len(d)))
                if k == 0:
                    d[p] ^= 1 << int(rng.integers(0, 8))
                elif k == 1:
                    d[p] = int(rng.integers(0, 256))
                elif k == 2:
                    del d[p:]
                    if not d:
                        d = bytearray(b"\0")
                else:
                    d[p:p] = bytes([int(rng.integers(0, 256))])
            try:
                img = W.textures.decode_image(bytes(d))
                assert img.ndim == 3 and img.shape[2] == 4
                decoded += 1
            except W.textures.ImageDecodeError:
                refused += 1
    assert decoded + refused == 1600 and refused > 0
