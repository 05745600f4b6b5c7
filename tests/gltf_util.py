"""A small glTF 2.0 writer (GLB or JSON + data URIs) and a numpy restatement of what the scene compiler must make of it
(loader.rs:7-354, lib.rs:149-184 + 372-491, rebuilder.rs:36-91) — the checker for tests/test_gltf.py."""
import base64
import json
import struct

import numpy as np

F32, U8, U16, U32, I8, I16 = 5126, 5121, 5123, 5125, 5120, 5122
_DT = {F32: np.float32, U8: np.uint8, U16: np.uint16, U32: np.uint32, I8: np.int8, I16: np.int16}
_NC = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


class GltfBuilder:
    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "buffers": [{}], "bufferViews": [], "accessors": [], "meshes": [], "nodes": [],
                    "materials": [], "textures": [], "images": [], "skins": [], "animations": [], "scenes": [{"nodes": []}], "scene": 0}

    def view(self, data, stride=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        self.bin += bytes(data)
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        self.doc["bufferViews"].append(v)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, array, ctype, atype, normalized=False, view=None, offset=0, count=None, minmax=False):
        a = np.ascontiguousarray(array, dtype=_DT[ctype])
        if view is None:
            view = self.view(a.tobytes())
        acc = {"bufferView": view, "byteOffset": offset, "componentType": ctype, "type": atype,
               "count": int(count if count is not None else a.size // _NC[atype])}
        if normalized:
            acc["normalized"] = True
        if minmax:
            acc["min"] = a.reshape(-1, _NC[atype]).min(0).tolist()
            acc["max"] = a.reshape(-1, _NC[atype]).max(0).tolist()
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def image_texture(self, blob, mime="image/png"):
        v = self.view(blob)
        self.doc["images"].append({"bufferView": v, "mimeType": mime})
        self.doc["textures"].append({"source": len(self.doc["images"]) - 1})
        return len(self.doc["textures"]) - 1

    def external_texture(self, uri="missing.png"):
        self.doc["images"].append({"uri": uri})
        self.doc["textures"].append({"source": len(self.doc["images"]) - 1})
        return len(self.doc["textures"]) - 1

    def _clean(self):
        d = {k: v for k, v in self.doc.items() if not (isinstance(v, list) and not v)}
        d["buffers"] = [{"byteLength": len(self.bin)}]
        return d

    def glb(self):
        d = self._clean()
        js = json.dumps(d).encode()
        js += b" " * (-len(js) % 4)
        b = bytes(self.bin) + b"\0" * (-len(self.bin) % 4)
        total = 12 + 8 + len(js) + 8 + len(b)
        return struct.pack("<4sII", b"glTF", 2, total) + struct.pack("<II", len(js), 0x4e4f534a) + js + \
            struct.pack("<II", len(b), 0x004e4942) + b

    def gltf_json(self):
        d = self._clean()
        d["buffers"] = [{"byteLength": len(self.bin),
                         "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()}]
        return json.dumps(d, indent=1).encode()


# ------------------------------------------------------------------------------------------- numpy restatement
f32 = np.float32


def quat_normalize(q):
    q = np.asarray(q, f32)
    return q * (f32(1) / np.sqrt(np.dot(q, q).astype(f32)))


def acos_approx(v):
    x = abs(f32(v))
    omx = max(f32(1) - x, f32(0))
    r = f32(-0.0012624911)
    for c in (0.0066700901, -0.0170881256, 0.0308918810, -0.0501743046, 0.0889789874, -0.2145988016, 1.5707963050):
        r = r * x + f32(c)
    r = r * np.sqrt(omx)
    return r if v >= 0 else f32(np.pi) - r


def quat_slerp(a, b, s):
    a, b, s = np.asarray(a, f32), np.asarray(b, f32), f32(s)
    d = np.dot(a, b)
    if d < 0:
        b, d = -b, -d
    if d > 1.0 - 1.1920929e-7:
        return quat_normalize(a + (b - a) * s)
    th = acos_approx(d)
    return (a * np.sin(th * (f32(1) - s)) + b * np.sin(th * s)) * (f32(1) / np.sin(th))


def mat_from_srt(s, q, t):
    x, y, z, w = [f32(v) for v in q]
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz, yy, yz, zz, wx, wy, wz = x * x2, x * y2, x * z2, y * y2, y * z2, z * z2, w * x2, w * y2, w * z2
    m = np.zeros((4, 4), f32)   # m[row, col]
    m[:3, 0] = np.array([1 - (yy + zz), xy + wz, xz - wy], f32) * f32(s[0])
    m[:3, 1] = np.array([xy - wz, 1 - (xx + zz), yz + wx], f32) * f32(s[1])
    m[:3, 2] = np.array([xz + wy, yz - wx, 1 - (xx + yy)], f32) * f32(s[2])
    m[:3, 3] = np.asarray(t, f32)
    m[3, 3] = 1
    return m


def sample_channel(inputs, outputs, interpolation, duration, time):
    """lib.rs:395-490 for one channel; returns (value_prev, value_next, factor)"""
    inputs = np.asarray(inputs, f32)
    time = f32(np.fmod(f32(time), f32(duration))) if duration > 0 else f32(time)
    count = len(inputs)
    nxt = 0
    while nxt < count and inputs[nxt] < time:
        nxt += 1
    if nxt == 0:
        nxt = 1
    if nxt >= count:
        nxt = 0
    prev = count - 1 if nxt == 0 else nxt - 1
    t0, t1 = inputs[prev], inputs[nxt]
    dt = f32(duration) - t0 + t1 if t1 < t0 else t1 - t0
    cur = (time - t0 if time >= t0 else (f32(duration) - t0) + time) if t1 < t0 else time - t0
    fac = min(max(cur / dt, f32(0)), f32(1)) if dt > 0.0001 else f32(0)
    stride, off = (3, 1) if interpolation == "CUBICSPLINE" else (1, 0)
    if interpolation == "STEP":
        fac = f32(0)
    return np.asarray(outputs[prev * stride + off], f32), np.asarray(outputs[nxt * stride + off], f32), f32(fac)


def node_globals(nodes):
    """nodes: list of dicts {t, r, s, children}; returns list of 4x4 global matrices (lib.rs:372-381)"""
    n = len(nodes)
    parent = [-1] * n
    for i, nd in enumerate(nodes):
        for c in nd.get("children", []):
            parent[c] = i
    out = [np.eye(4, dtype=f32) for _ in range(n)]

    def rec(i, pm):
        g = (pm @ mat_from_srt(nodes[i]["s"], nodes[i]["r"], nodes[i]["t"])).astype(f32)
        out[i] = g
        for c in nodes[i].get("children", []):
            rec(c, g)
    for i in range(n):
        if parent[i] < 0:
            rec(i, np.eye(4, dtype=f32))
    return out


def skin_vertices(pos, nrm, joints, weights, joint_mats):
    """rebuilder.rs:59-91"""
    outp, outn = np.zeros_like(pos, dtype=f32), np.zeros_like(nrm, dtype=f32)
    for i in range(len(pos)):
        m = np.zeros((4, 4), f32)
        for k in range(4):
            if weights[i, k] > 0:
                m = m + joint_mats[joints[i, k]] * f32(weights[i, k])
        if not m.any():
            m = np.eye(4, dtype=f32)
        outp[i] = (m @ np.append(pos[i], f32(1)).astype(f32))[:3]
        v = (m[:3, :3] @ nrm[i].astype(f32)).astype(f32)
        ln = np.sqrt(np.dot(v, v))
        outn[i] = v / ln if ln > 0 and np.isfinite(1 / ln) else 0
    return outp, outn
