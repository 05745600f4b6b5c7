"""ctypes binding of the CPU oracle (oracle/_build/librt_oracle.so) for tests, smoke() and the
cpu_baseline leg of bench.py.  The class mirrors WebGPURenderer's method surface so a parity
test drives both implementations with the same call sequence."""
import ctypes
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "librt_oracle.so")

_lib = None


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, "rt_oracle.cpp"),
            os.path.join(REPO, "include", "mi355rt_math.h"),
            os.path.join(REPO, "include", "mi355rt_layout.h")]
    if (not force and os.path.exists(ORACLE_LIB)
            and all(os.path.getmtime(s) <= os.path.getmtime(ORACLE_LIB) for s in srcs)):
        return ORACLE_LIB
    subprocess.run(["make", "-C", ORACLE_DIR, "-B"], check=True, stdout=subprocess.DEVNULL)
    return ORACLE_LIB


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = ctypes.CDLL(ORACLE_LIB)
        vp, u32, f32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float
        L.oracle_create.restype = vp
        for name, args in {
            "oracle_destroy": [vp], "oracle_set_threads": [vp, ctypes.c_int],
            "oracle_build_pipeline": [vp, u32, u32], "oracle_update_screen_size": [vp, u32, u32],
            "oracle_reset_accumulation": [vp], "oracle_upload_textures": [vp, vp, u32],
            "oracle_update_topology": [vp, vp, ctypes.c_size_t], "oracle_update_instances": [vp, vp, ctypes.c_size_t],
            "oracle_update_lights": [vp, vp, ctypes.c_size_t], "oracle_update_draw_commands": [vp, vp, ctypes.c_size_t],
            "oracle_update_geometry": [vp, vp, vp, vp, u32], "oracle_update_bvh": [vp, vp, u32, vp, u32],
            "oracle_update_scene_uniforms": [vp, vp, u32, u32], "oracle_set_stripes": [vp, u32, u32, u32],
            "oracle_compute": [vp, u32], "oracle_present": [vp], "oracle_capture_frame": [vp, vp],
            "oracle_read_accum": [vp, vp], "oracle_write_accum": [vp, vp], "oracle_read_gbuffer": [vp, vp, vp, vp],
            "oracle_read_history": [vp, vp], "oracle_read_uniforms": [vp, vp], "oracle_get_counters": [vp, vp],
            "oracle_reset_counters": [vp], "oracle_resize_texture": [vp, u32, u32, vp],
            "oracle_set_node_histogram": [vp, vp], "oracle_trace_vs_brute_force": [vp, vp, u32, vp, vp, vp],
            "oracle_trace_rays": [vp, vp, u32, ctypes.c_int, vp, vp],
        }.items():
            getattr(L, name).argtypes = args
            getattr(L, name).restype = None
        L.oracle_hardware_threads.restype = ctypes.c_int
        L.oracle_init_rng.restype = u32
        L.oracle_init_rng.argtypes = [u32, u32]
        L.oracle_rand_pcg.restype = f32
        L.oracle_rand_pcg.argtypes = [ctypes.POINTER(u32)]
        L.oracle_halton.restype = ctypes.c_double
        L.oracle_halton.argtypes = [u32, u32]
        L.oracle_sincos.argtypes = [f32, ctypes.POINTER(f32), ctypes.POINTER(f32)]
        for n in ("oracle_exp", "oracle_log"):
            getattr(L, n).restype = f32
            getattr(L, n).argtypes = [f32]
        for n in ("oracle_pow", "oracle_min", "oracle_max"):
            getattr(L, n).restype = f32
            getattr(L, n).argtypes = [f32, f32]
        L.oracle_f32_to_f16.restype = ctypes.c_uint16
        L.oracle_f32_to_f16.argtypes = [f32]
        L.oracle_f16_to_f32.restype = f32
        L.oracle_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.oracle_pack_normal.argtypes = [vp, vp]
        L.oracle_unpack_normal.argtypes = [vp, vp]
        L.oracle_hit_triangle.restype = f32
        L.oracle_hit_triangle.argtypes = [vp, vp, vp, vp, vp, f32, f32]
        L.oracle_intersect_aabb.restype = f32
        L.oracle_intersect_aabb.argtypes = [vp, vp, vp, vp, f32, f32]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


COUNTER_NAMES = ("primary_rays", "extension_rays", "shadow_rays", "nodes_visited", "tris_tested", "shaded_hits")


class OracleRenderer:
    """CPU oracle with the WebGPURenderer method names (WebGPURenderer.ts:7-138)."""

    def __init__(self, threads=0):
        self.L = lib()
        self.ctx = self.L.oracle_create()
        self.L.oracle_set_threads(self.ctx, threads)
        self.width = self.height = 0

    def __del__(self):
        if getattr(self, "ctx", None):
            self.L.oracle_destroy(self.ctx)
            self.ctx = None

    def buildPipeline(self, depth, spp):
        self.L.oracle_build_pipeline(self.ctx, depth, spp)

    def updateScreenSize(self, width, height):
        self.width, self.height = width, height
        self.L.oracle_update_screen_size(self.ctx, width, height)

    def resetAccumulation(self):
        self.L.oracle_reset_accumulation(self.ctx)

    def loadTexturesFromWorld(self, bridge):
        n = bridge.textureCount
        if n == 0:
            return
        layers = []
        for i in range(n):
            layer = bridge.getTextureRGBA(i)
            if layer is None:
                # encoded image (glTF input): the checker decodes with PIL (independent of the product's decoder) and
                # resizes with the oracle's restatement of the rule; missing / undecodable -> white fallback layer
                layer = np.empty((1024, 1024, 4), np.uint8)
                src = None
                blob = bridge.getTexture(i)
                if blob:
                    try:
                        import io
                        from PIL import Image
                        src = np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(blob)).convert("RGBA")))
                    except Exception:
                        src = None
                if src is None:
                    self.L.oracle_resize_texture(None, 0, 0, _ptr(layer))
                else:
                    self.L.oracle_resize_texture(_ptr(src), src.shape[1], src.shape[0], _ptr(layer))
            layers.append(layer)
        layers = np.ascontiguousarray(np.stack(layers), dtype=np.uint8)
        self.L.oracle_upload_textures(self.ctx, _ptr(layers), n)

    def updateBuffer(self, kind, data):
        a = np.ascontiguousarray(data)
        fn = {"topology": self.L.oracle_update_topology, "instance": self.L.oracle_update_instances,
              "lights": self.L.oracle_update_lights, "draw_commands": self.L.oracle_update_draw_commands}[kind]
        fn(self.ctx, _ptr(a), a.size)
        return False

    def updateCombinedGeometry(self, v, n, uv):
        v, n, uv = (np.ascontiguousarray(x, dtype=np.float32) for x in (v, n, uv))
        self.L.oracle_update_geometry(self.ctx, _ptr(v), _ptr(n), _ptr(uv), v.size // 4)
        return False

    def updateCombinedBVH(self, tlas, blas):
        tlas, blas = (np.ascontiguousarray(x, dtype=np.float32) for x in (tlas, blas))
        self.L.oracle_update_bvh(self.ctx, _ptr(tlas), tlas.size // 8, _ptr(blas), blas.size // 8)
        return False

    def updateSceneUniforms(self, cameraData, frameCount, lightCount):
        cam = np.ascontiguousarray(cameraData, dtype=np.float32)
        self.L.oracle_update_scene_uniforms(self.ctx, _ptr(cam), frameCount, lightCount)

    def recreateBindGroup(self):
        pass

    def setStripes(self, stripe_rows, rank, count):
        self.L.oracle_set_stripes(self.ctx, stripe_rows, rank, count)

    def compute(self, frameCount):
        self.L.oracle_compute(self.ctx, frameCount)

    def present(self):
        self.L.oracle_present(self.ctx)

    def captureFrame(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self.L.oracle_capture_frame(self.ctx, _ptr(out))
        return {"data": out, "width": self.width, "height": self.height}

    def sync(self):
        pass

    # --- parity artefacts ---
    def readAccum(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self.L.oracle_read_accum(self.ctx, _ptr(out))
        return out

    def writeAccum(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        self.L.oracle_write_accum(self.ctx, _ptr(a))

    def readGBuffer(self):
        alb = np.empty((self.height, self.width, 4), dtype=np.uint8)
        nid = np.empty((self.height, self.width, 4), dtype=np.float32)
        dep = np.empty((self.height, self.width), dtype=np.float32)
        self.L.oracle_read_gbuffer(self.ctx, _ptr(alb), _ptr(nid), _ptr(dep))
        return alb, nid, dep

    def readHistory(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint16)
        self.L.oracle_read_history(self.ctx, _ptr(out))
        return out

    def readUniforms(self):
        out = np.empty(256, dtype=np.uint8)
        self.L.oracle_read_uniforms(self.ctx, _ptr(out))
        return out

    def getCounters(self):
        out = np.zeros(6, dtype=np.uint64)
        self.L.oracle_get_counters(self.ctx, _ptr(out))
        return dict(zip(COUNTER_NAMES, (int(x) for x in out)))

    def resetCounters(self):
        self.L.oracle_reset_counters(self.ctx)

    def traceVsBruteForce(self, rays, skip_tri=None):
        """rays (n, 8) f32 {o, t_min, d, t_max} -> (bvh (n, 3) {t, tri, inst}, brute (n, 4) {t, tri, inst, ties})"""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        skip = None if skip_tri is None else np.ascontiguousarray(skip_tri, dtype=np.uint8)
        n = rays.shape[0]
        bvh = np.empty((n, 3), np.float32)
        brute = np.empty((n, 4), np.float32)
        self.L.oracle_trace_vs_brute_force(self.ctx, _ptr(rays), n, _ptr(bvh), _ptr(brute), None if skip is None else _ptr(skip))
        return bvh, brute


def _trace_rays(self, rays, any_hit=False):
    """rays (n, 8) f32 {o, t_min, d, t_max} through the restated traversal, ray by ray:
    -> (out (n, 4) f32 {t, tri, inst, occluded}, counts (n, 2) u64 {nodes_visited, tris_tested})"""
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    n = rays.shape[0]
    out = np.empty((n, 4), np.float32)
    counts = np.empty((n, 2), np.uint64)
    self.L.oracle_trace_rays(self.ctx, _ptr(rays), n, 1 if any_hit else 0, _ptr(out), _ptr(counts))
    return out, counts


OracleRenderer.traceRays = _trace_rays
