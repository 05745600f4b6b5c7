"""Texture ingest, host side: ctypes binding of include/mi355tex.h (PNG / JPEG -> RGBA8) and the
PNG container the bridge hands its procedural textures out in.

Reference: `ResourceManager.loadTexturesFromWorld` (src/renderer/ResourceManager.ts:153-198) asks the bridge for
ENCODED images (`bridge.getTexture(i)`, world-bridge.ts) and lets the browser decode and resize them.  Here the
decode is `mt_decode` (host C++), the resize `rt_upload_texture_image` (HIP kernel)."""
import ctypes
import os
import struct
import zlib

import numpy as np

from . import _build

_lib = None


class MtImage(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("rgba", ctypes.POINTER(ctypes.c_uint8))]


class ImageDecodeError(RuntimeError):
    pass


TEX_EXPORTED_SYMBOLS = "mt_probe mt_decode mt_free mt_last_error mt_inflate".split()


def load_tex_library(path=None):
    """dlopen libmi355tex.so and declare every symbol of include/mi355tex.h (host only, no GPU)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or _build.TEX_LIB
    if not os.path.exists(path):
        raise ImageDecodeError("image decode library not built: %s" % path)
    L = ctypes.CDLL(path)
    L.mt_probe.restype = ctypes.c_int
    L.mt_probe.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    L.mt_decode.restype = ctypes.c_int
    L.mt_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(MtImage)]
    L.mt_free.restype = None
    L.mt_free.argtypes = [ctypes.POINTER(MtImage)]
    L.mt_last_error.restype = ctypes.c_char_p
    L.mt_last_error.argtypes = []
    L.mt_inflate.restype = ctypes.c_long
    L.mt_inflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    if path == _build.TEX_LIB:
        _lib = L
    return L


def probe(blob):
    """0 unknown, 1 PNG, 2 JPEG (magic bytes, like the browser's sniffing of an untyped Blob)."""
    b = bytes(blob)
    return load_tex_library().mt_probe(b, len(b))


def decode_image(blob):
    """Encoded PNG / JPEG bytes -> (height, width, 4) uint8 straight-alpha RGBA; raises ImageDecodeError."""
    L = load_tex_library()
    b = bytes(blob)
    im = MtImage()
    rc = L.mt_decode(b, len(b), ctypes.byref(im))
    if rc != 0:
        raise ImageDecodeError("mt_decode failed (%d): %s" % (rc, L.mt_last_error().decode()))
    try:
        return np.ctypeslib.as_array(im.rgba, shape=(im.height, im.width, 4)).copy()
    finally:
        L.mt_free(ctypes.byref(im))


def inflate(data, max_size):
    L = load_tex_library()
    b = bytes(data)
    out = np.empty(max(1, max_size), dtype=np.uint8)
    n = L.mt_inflate(b, len(b), out.ctypes.data_as(ctypes.c_void_p), max_size)
    if n < 0:
        raise ImageDecodeError("mt_inflate failed (%d): %s" % (n, L.mt_last_error().decode()))
    return out[:n].tobytes()


def encode_png(rgba, level=1):
    """(h, w, 4) uint8 -> PNG bytes (colour type 6, filter 0 rows).  Stands in for the encoded images a glTF would
    carry: the procedural scenes only have raw texels, and the reference's bridge hands out encoded blobs."""
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = a.shape[:2]
    rows = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * 4)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(rows, level)) + chunk(b"IEND", b""))
