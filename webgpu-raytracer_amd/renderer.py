"""WebGPURenderer — host-side mirror of src/renderer/WebGPURenderer.ts over libmi355rt.so.

Same method names, argument meaning and return values as the reference class
(WebGPURenderer.ts:7-138).  Every call goes through the C ABI in include/mi355rt.h; there is no
CPU fallback: if the HIP library is missing or no device is present the constructor raises.
"""
import ctypes
import os

import numpy as np

from . import _build

RT_OK, RT_REALLOCATED, RT_SKIPPED = 0, 1, 2
_KINDS = {"topology": 0, "instance": 1, "lights": 2, "draw_commands": 3}
COUNTER_NAMES = ("primary_rays", "extension_rays", "shadow_rays", "nodes_visited", "tris_tested", "shaded_hits")

_lib = None


class RendererError(RuntimeError):
    pass


_hip_runtime = None
hip_runtime_note = None   # what the last load_library() decided about the HIP runtime, for diagnostics


def _elf_dynamic_strings(path, want_tags):
    """DT_NEEDED (1) / DT_SONAME (14) strings of a 64-bit little-endian ELF, read straight from the file (no tool, no dlopen)."""
    import struct
    out = {t: [] for t in want_tags}
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        return out
    e_shoff, = struct.unpack_from("<Q", data, 0x28)
    e_shentsize, e_shnum = struct.unpack_from("<HH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, e_shoff + i * e_shentsize) for i in range(e_shnum)]
    for sh in secs:
        if sh[1] != 6:          # SHT_DYNAMIC
            continue
        stroff = secs[sh[6]][4]  # sh_link -> .dynstr
        for off in range(sh[4], sh[4] + sh[5], 16):
            tag, val = struct.unpack_from("<qQ", data, off)
            if tag == 0:
                break
            if tag in out:
                end = data.index(b"\0", stroff + val)
                out[tag].append(data[stroff + val:end].decode())
    return out


def mapped_hip_runtimes():
    """Paths of every libamdhip64 mapped into this process (/proc/self/maps): more than one = two runtimes."""
    found = set()
    try:
        with open("/proc/self/maps") as maps:
            for line in maps:
                f = line.split()
                if len(f) >= 6 and os.path.basename(f[5]).startswith("libamdhip64.so"):
                    found.add(os.path.realpath(f[5]))
    except OSError:
        pass
    return sorted(found)


def _preload_hip_runtime(rt_lib_path):
    """ONE HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and asks for it
    as `libamdhip64.so`; libmi355rt.so asks for `libamdhip64.so.7` (RUNPATH /opt/rocm).  The loader matches by the
    requested name, so whichever came second used to get a second copy of the runtime: `import torch` after the renderer
    then found "No HIP GPUs", and stream handles or events of one copy meant nothing to the other.  When torch is installed
    its copy is therefore loaded first, globally — our NEEDED entry matches its SONAME, torch's later request resolves to
    the same file — and the renderer, torch and RCCL share one runtime whatever the import order.

    Guarded (a user who never touches torch must not be broken by it): the candidate is used only when its SONAME is what
    libmi355rt.so NEEDs (else the preload could not satisfy our request and the two-runtime bug would come back silently:
    warned instead); a candidate that fails to load (missing comgr / hsa dependencies, CPU-only torch layout) is skipped
    with a warning and the RUNPATH runtime is used.  MI355RT_HIP_RUNTIME=<path> names the runtime explicitly,
    MI355RT_HIP_RUNTIME=system never preloads.  No GPU is touched here."""
    global _hip_runtime, hip_runtime_note
    if _hip_runtime is not None:
        return
    import warnings
    _hip_runtime = False
    path = os.environ.get("MI355RT_HIP_RUNTIME")
    if path == "system":
        hip_runtime_note = "system runtime of the RUNPATH (MI355RT_HIP_RUNTIME=system)"
        return
    already = mapped_hip_runtimes()
    if already and not path:
        hip_runtime_note = "a HIP runtime is already mapped (%s): nothing preloaded" % ", ".join(already)
        return
    explicit = bool(path)
    if not path:
        import importlib.util
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(cand):
                path = cand
    if not path:
        hip_runtime_note = "system runtime of the RUNPATH (no torch-bundled libamdhip64.so found)"
        return
    try:
        needed = [n for n in _elf_dynamic_strings(rt_lib_path, (1,))[1] if n.startswith("libamdhip64.so")]
        soname = _elf_dynamic_strings(path, (14,))[14]
    except (OSError, ValueError, IndexError, KeyError) as e:
        needed, soname = [], []
        warnings.warn("mi355rt: cannot read the ELF dynamic section for the HIP runtime check (%s)" % e, RuntimeWarning)
    if needed and soname and soname[0] not in needed:
        hip_runtime_note = "%s has SONAME %s but libmi355rt.so needs %s: not preloaded" % (path, soname[0], needed[0])
        warnings.warn("mi355rt: " + hip_runtime_note + " — torch and the renderer will use DIFFERENT HIP runtimes in this "
                      "process (streams / events cannot be shared; set MI355RT_HIP_RUNTIME to a matching libamdhip64.so)",
                      RuntimeWarning)
        return
    try:
        _hip_runtime = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        hip_runtime_note = "preloaded %s%s" % (path, "" if explicit else " (torch's bundled runtime)")
    except OSError as e:
        _hip_runtime = False
        hip_runtime_note = "could not preload %s (%s): system runtime of the RUNPATH" % (path, e)
        warnings.warn("mi355rt: " + hip_runtime_note + (" — a later `import torch` would map a second HIP runtime"
                                                         if not explicit else ""), RuntimeWarning)


def load_library(path=None):
    """dlopen libmi355rt.so and declare every symbol of include/mi355rt.h.
    Loading does not touch the GPU; rt_create does."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or _build.RT_LIB
    if not os.path.exists(path):
        raise RendererError(
            "HIP renderer library not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    _preload_hip_runtime(path)
    L = ctypes.CDLL(path)
    maps = mapped_hip_runtimes()
    if len(maps) > 1:   # the two-runtime state the preload exists to prevent: say so instead of failing later in odd ways
        import warnings
        warnings.warn("mi355rt: %d HIP runtimes are mapped in this process (%s); handles of one mean nothing to the other"
                      % (len(maps), ", ".join(maps)), RuntimeWarning)
    vp, u32, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
    sigs = {
        "rt_create": (vp, [i32]), "rt_destroy": (None, [vp]), "rt_last_error": (ctypes.c_char_p, [vp]),
        "rt_set_pipeline": (i32, [vp, u32, u32]), "rt_resize": (i32, [vp, u32, u32]),
        "rt_reset_accum": (i32, [vp]), "rt_upload_textures": (i32, [vp, vp, u32]),
        "rt_alloc_texture_layers": (i32, [vp, u32]), "rt_upload_texture_image": (i32, [vp, u32, vp, u32, u32]),
        "rt_read_texture_layer": (i32, [vp, u32, vp, ctypes.c_size_t]),
        "rt_build_blas": (i32, [vp, vp, u32, vp, u32, vp, u32, ctypes.POINTER(u32), vp]),
        "rt_upload": (i32, [vp, i32, vp, ctypes.c_size_t]),
        "rt_upload_geometry": (i32, [vp, vp, vp, vp, u32]),
        "rt_upload_bvh": (i32, [vp, vp, u32, vp, u32]),
        "rt_set_scene": (i32, [vp, vp, u32, u32]), "rt_recreate_bind_group": (i32, [vp]),
        "rt_compute": (i32, [vp, u32]), "rt_present": (i32, [vp]),
        "rt_compute_batch": (i32, [vp, vp, u32]),
        "rt_capture": (i32, [vp, vp, ctypes.c_size_t]), "rt_sync": (i32, [vp]),
        "rt_read_accum": (i32, [vp, vp, ctypes.c_size_t]), "rt_write_accum": (i32, [vp, vp, ctypes.c_size_t]),
        "rt_read_gbuffer": (i32, [vp, vp, vp, vp]), "rt_read_history": (i32, [vp, vp, ctypes.c_size_t]),
        "rt_read_uniforms": (i32, [vp, vp]), "rt_get_counters": (i32, [vp, vp]),
        "rt_get_kernel_counters": (i32, [vp, i32, vp]), "rt_bind_accum": (i32, [vp, vp]),
        "rt_reset_counters": (i32, [vp]), "rt_set_counting": (i32, [vp, i32]),
        "rt_set_stripes": (i32, [vp, u32, u32, u32]), "rt_accum_device_ptr": (vp, [vp]),
        "rt_set_stream": (i32, [vp, vp]), "rt_bind_present_source": (i32, [vp, vp]),
        "rt_debug_clock_stamps": (i32, [vp, vp, u32]), "rt_debug_trace_sections": (i32, [vp, vp, i32]), "rt_debug_pt_sections": (i32, [vp, vp, i32]), "rt_debug_lane_stats": (i32, [vp, vp, i32]),
        "rt_debug_read_traversal_nodes": (i32, [vp, vp, vp, vp, u32]),
        "rt_debug_read_pairs": (i32, [vp, vp, vp, u32]),
        "rt_debug_ieee_check": (i32, [vp, i32, ctypes.c_uint64, ctypes.c_uint64, vp]),
        "rt_kernel_times": (i32, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u32), u32]),
        "rt_kernel_time_ms": (i32, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(u32)]),
        "rt_set_kernel_timing": (i32, [vp, i32]), "rt_device_count": (i32, []),
        "rt_set_kernel_variant": (i32, [vp, i32]), "rt_set_walk": (i32, [vp, i32]), "rt_set_lookahead": (i32, [vp, u32]),
        "rt_set_lookahead_limit": (i32, [vp, u32]),
        "rt_build_blas_levels": (i32, [vp]),
        "rt_world_update": (i32, [vp, vp]), "rt_world_set_static_cache": (i32, [vp, i32]), "rt_world_last_ms": (ctypes.c_double, [vp]), "rt_world_last_tlas_ms": (ctypes.c_double, [vp]),
        "rt_world_read": (i32, [vp, i32, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if path == _build.RT_LIB:
        _lib = L
    return L


EXPORTED_SYMBOLS = (
    "rt_create rt_destroy rt_last_error rt_set_pipeline rt_resize rt_reset_accum rt_upload_textures rt_upload "
    "rt_alloc_texture_layers rt_upload_texture_image rt_read_texture_layer rt_build_blas "
    "rt_upload_geometry rt_upload_bvh rt_set_scene rt_recreate_bind_group rt_compute rt_compute_batch rt_present rt_capture "
    "rt_sync rt_read_accum rt_write_accum rt_read_gbuffer rt_read_history rt_read_uniforms rt_get_counters "
    "rt_get_kernel_counters rt_bind_accum rt_bind_present_source rt_kernel_times rt_debug_clock_stamps rt_debug_trace_sections rt_debug_pt_sections rt_debug_lane_stats rt_debug_read_traversal_nodes rt_debug_read_pairs rt_debug_ieee_check "
    "rt_reset_counters rt_set_counting rt_set_stripes rt_accum_device_ptr rt_set_stream rt_kernel_time_ms "
    "rt_set_kernel_timing rt_device_count rt_set_kernel_variant rt_set_walk rt_set_lookahead "
    "rt_world_update rt_world_last_ms rt_world_last_tlas_ms rt_world_read rt_build_blas_levels rt_set_lookahead_limit rt_world_set_static_cache").split()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class WebGPURenderer:
    """`new WebGPURenderer(canvas)` + `await init()` (WebGPURenderer.ts:17-32).  The canvas
    argument of the reference becomes a device ordinal; width/height come from updateScreenSize."""

    def __init__(self, device=0):
        self.L = load_library()
        self.ctx = self.L.rt_create(int(device))
        if not self.ctx:
            msg = self.L.rt_last_error(None)
            raise RendererError("rt_create(%d) failed: %s" % (device, msg.decode() if msg else "unknown"))
        self.width = self.height = 0
        self._capture_buf = None  # reused between captures like readbackResultBuffer (WebGPUContext.ts:82-89)

    def __del__(self):
        self.destroy()

    def destroy(self):
        if getattr(self, "ctx", None):
            self.L.rt_destroy(self.ctx)
            self.ctx = None

    def _check(self, rc, what):
        if rc < 0:
            msg = self.L.rt_last_error(self.ctx)
            raise RendererError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
        return rc

    # ---- reference surface ----
    def init(self):
        return None

    def buildPipeline(self, depth, spp):
        self._check(self.L.rt_set_pipeline(self.ctx, int(depth), int(spp)), "buildPipeline")

    def updateScreenSize(self, width, height):
        self.width, self.height = int(width), int(height)
        self._check(self.L.rt_resize(self.ctx, self.width, self.height), "updateScreenSize")

    def resetAccumulation(self):
        self._check(self.L.rt_reset_accum(self.ctx), "resetAccumulation")

    def loadTexturesFromWorld(self, bridge):
        """ResourceManager.ts:153-198: no textures -> default white texture; otherwise one 1024x1024 layer per
        encoded image of the bridge — decoded on the host (mi355tex), resized on the GPU — and the white fallback
        bitmap for an image that is missing or does not decode (the reference warns and carries on, :169-175)."""
        n = bridge.textureCount
        if n == 0:
            self._check(self.L.rt_upload_textures(self.ctx, None, 0), "loadTexturesFromWorld")
            return
        from . import textures
        self._check(self.L.rt_alloc_texture_layers(self.ctx, n), "loadTexturesFromWorld")
        self.texture_warnings = []
        for i in range(n):
            data = bridge.getTexture(i)
            img = None
            if data is not None:
                try:
                    img = textures.decode_image(data)
                except textures.ImageDecodeError as e:
                    self.texture_warnings.append("Failed tex %d: %s" % (i, e))
            self.uploadTextureImage(i, img)

    def uploadTextureImage(self, layer, rgba):
        """One layer from an (h, w, 4) uint8 image of any size (GPU bilinear resize); None = white fallback."""
        if rgba is None:
            self._check(self.L.rt_upload_texture_image(self.ctx, layer, None, 0, 0), "uploadTextureImage")
            return
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 4:
            raise RendererError("uploadTextureImage expects an (h, w, 4) uint8 array")
        self._check(self.L.rt_upload_texture_image(self.ctx, layer, _ptr(a), a.shape[1], a.shape[0]), "uploadTextureImage")

    def loadTextureLayers(self, layers):
        """Already decoded and resized (n, 1024, 1024, 4) uint8 layers (rt_upload_textures)."""
        a = np.ascontiguousarray(layers, dtype=np.uint8)
        self._check(self.L.rt_upload_textures(self.ctx, _ptr(a), a.shape[0]), "loadTextureLayers")

    def buildBlas(self, verts4, indices):
        """GPU binned-SAH BLAS build (rt_build_blas): returns (nodes (n, 8) float32, order (n_tris,) uint32)."""
        v = np.ascontiguousarray(verts4, dtype=np.float32).reshape(-1, 4)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        n_tris = idx.size // 3
        nodes = np.empty((max(1, 2 * n_tris), 8), dtype=np.float32)
        order = np.empty(max(1, n_tris), dtype=np.uint32)
        n_nodes = ctypes.c_uint32()
        self._check(self.L.rt_build_blas(self.ctx, _ptr(v), v.shape[0], _ptr(idx), n_tris, _ptr(nodes), nodes.shape[0],
                                         ctypes.byref(n_nodes), _ptr(order)), "buildBlas")
        return nodes[:n_nodes.value].copy(), order[:n_tris].copy()

    # the bridge arrays of the device-resident world (rt_world_read): name -> (rt_world_array, dtype, row width)
    _WORLD_ARRAYS = {"vertices": (0, np.float32, 4), "normals": (1, np.float32, 4), "uvs": (2, np.float32, 2),
                     "mesh_topology": (3, np.uint32, 20), "tlas": (4, np.float32, 8), "blas": (5, np.float32, 8),
                     "instances": (6, np.float32, 36), "lights": (7, np.uint32, 2), "draw_commands": (8, np.uint32, 4)}

    def worldRead(self, name):
        """One bridge array as the last device-resident update(t) left it in HBM (flat, the bridge getter's dtype)."""
        which, dtype, _ = self._WORLD_ARRAYS[name]
        n = ctypes.c_size_t()
        self._check(self.L.rt_world_read(self.ctx, which, None, 0, ctypes.byref(n)), "worldRead(%s)" % name)
        out = np.empty(n.value // 4, dtype=dtype)
        if n.value:
            self._check(self.L.rt_world_read(self.ctx, which, _ptr(out), out.nbytes, ctypes.byref(n)), "worldRead(%s)" % name)
        return out

    def setWorldStaticCache(self, enabled):
        """device-resident update(t): keep the BLAS / rows of geometries without a skin between frames (default on)"""
        self._check(self.L.rt_world_set_static_cache(self.ctx, 1 if enabled else 0), "setWorldStaticCache")

    def worldLastMs(self):
        """GPU stream time of the last device-resident update(t) (ms)."""
        return float(self.L.rt_world_last_ms(self.ctx))

    def worldLastTlasMs(self):
        """... of its TLAS kernel alone (ms)."""
        return float(self.L.rt_world_last_tlas_ms(self.ctx))

    def readTextureLayer(self, layer):
        out = np.empty((1024, 1024, 4), dtype=np.uint8)
        self._check(self.L.rt_read_texture_layer(self.ctx, layer, _ptr(out), out.nbytes), "readTextureLayer")
        return out

    def updateBuffer(self, kind, data):
        a = np.ascontiguousarray(data)
        rc = self._check(self.L.rt_upload(self.ctx, _KINDS[kind], _ptr(a), a.nbytes), "updateBuffer(%s)" % kind)
        return rc == RT_REALLOCATED

    def updateCombinedGeometry(self, v, n, uv):
        v, n, uv = (np.ascontiguousarray(x, dtype=np.float32) for x in (v, n, uv))
        rc = self._check(self.L.rt_upload_geometry(self.ctx, _ptr(v), _ptr(n), _ptr(uv), v.size // 4),
                         "updateCombinedGeometry")
        return rc == RT_REALLOCATED

    def updateCombinedBVH(self, tlas, blas):
        tlas, blas = (np.ascontiguousarray(x, dtype=np.float32) for x in (tlas, blas))
        rc = self._check(self.L.rt_upload_bvh(self.ctx, _ptr(tlas), tlas.size // 8, _ptr(blas), blas.size // 8),
                         "updateCombinedBVH")
        return rc == RT_REALLOCATED

    def updateSceneUniforms(self, cameraData, frameCount, lightCount):
        cam = np.ascontiguousarray(cameraData, dtype=np.float32)
        if cam.size != 24:
            raise ValueError("cameraData must hold 24 floats")
        self._check(self.L.rt_set_scene(self.ctx, _ptr(cam), int(frameCount), int(lightCount)), "updateSceneUniforms")

    def recreateBindGroup(self):
        self.L.rt_recreate_bind_group(self.ctx)

    def compute(self, frameCount):
        return self._check(self.L.rt_compute(self.ctx, int(frameCount)), "compute")

    def computeBatch(self, frameCounts):
        """`for k in batch: compute(samplesDone + k)` (VideoRecorder.ts:278-280) as one dispatch per kernel."""
        fc = np.ascontiguousarray(list(frameCounts), dtype=np.uint32)
        return self._check(self.L.rt_compute_batch(self.ctx, _ptr(fc), fc.size), "computeBatch")

    def present(self):
        return self._check(self.L.rt_present(self.ctx), "present")

    def captureFrame(self):
        if not self.width:
            raise RendererError("No render target")  # WebGPUContext.ts:43
        n = self.width * self.height * 4
        if self._capture_buf is None or self._capture_buf.size != n:
            self._capture_buf = np.empty(n, dtype=np.uint8)
        self._check(self.L.rt_capture(self.ctx, _ptr(self._capture_buf), n), "captureFrame")
        return {"data": self._capture_buf.reshape(self.height, self.width, 4),
                "width": self.width, "height": self.height}

    def sync(self):
        """`await renderer.device.queue.onSubmittedWorkDone()`"""
        self._check(self.L.rt_sync(self.ctx), "sync")

    # ---- additions (parity, checkpoint/resume, sharding, measurement) ----
    def readAccum(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self.L.rt_read_accum(self.ctx, _ptr(out), out.nbytes), "readAccum")
        return out

    def writeAccum(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        self._check(self.L.rt_write_accum(self.ctx, _ptr(a), a.nbytes), "writeAccum")

    def readGBuffer(self):
        alb = np.empty((self.height, self.width, 4), dtype=np.uint8)
        nid = np.empty((self.height, self.width, 4), dtype=np.float32)
        dep = np.empty((self.height, self.width), dtype=np.float32)
        self._check(self.L.rt_read_gbuffer(self.ctx, _ptr(alb), _ptr(nid), _ptr(dep)), "readGBuffer")
        return alb, nid, dep

    def readHistory(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint16)
        self._check(self.L.rt_read_history(self.ctx, _ptr(out), out.nbytes), "readHistory")
        return out

    def readUniforms(self):
        out = np.empty(256, dtype=np.uint8)
        self._check(self.L.rt_read_uniforms(self.ctx, _ptr(out)), "readUniforms")
        return out

    def getCounters(self):
        out = np.zeros(6, dtype=np.uint64)
        self._check(self.L.rt_get_counters(self.ctx, _ptr(out)), "getCounters")
        return dict(zip(COUNTER_NAMES, (int(x) for x in out)))

    def getKernelCounters(self, kernel):
        """kernel: 0 = primary visibility, 1 = path trace"""
        out = np.zeros(6, dtype=np.uint64)
        self._check(self.L.rt_get_kernel_counters(self.ctx, int(kernel), _ptr(out)), "getKernelCounters")
        return dict(zip(COUNTER_NAMES, (int(x) for x in out)))

    def bindAccum(self, device_ptr):
        self._check(self.L.rt_bind_accum(self.ctx, ctypes.c_void_p(device_ptr or 0)), "bindAccum")

    def bindPresentSource(self, device_ptr):
        """present() reads this float4 device buffer instead of the accumulation buffer (0 / None = default)."""
        self._check(self.L.rt_bind_present_source(self.ctx, ctypes.c_void_p(device_ptr or 0)), "bindPresentSource")

    def resetCounters(self):
        self._check(self.L.rt_reset_counters(self.ctx), "resetCounters")

    def setCounting(self, detailed):
        self._check(self.L.rt_set_counting(self.ctx, 1 if detailed else 0), "setCounting")

    def setStripes(self, stripe_rows, rank, count):
        self._check(self.L.rt_set_stripes(self.ctx, int(stripe_rows), int(rank), int(count)), "setStripes")

    def accumDevicePtr(self):
        return self.L.rt_accum_device_ptr(self.ctx)

    def setStream(self, hip_stream_handle):
        self._check(self.L.rt_set_stream(self.ctx, ctypes.c_void_p(hip_stream_handle)), "setStream")

    def setKernelVariant(self, variant):
        """3 = auto (default: persistent kernel for LDS-resident scenes, wavefront form for larger ones), 2 = wavefront,
        1 = persistent waves + path regeneration, 0 = one pixel per lane megakernel; all bit-identical"""
        self._check(self.L.rt_set_kernel_variant(self.ctx, int(variant)), "setKernelVariant")

    def setLookaheadLimit(self, frames_left):
        """the run of consecutive frames ends in `frames_left` frames (this one included): trace no further ahead; 0 = unknown"""
        self._check(self.L.rt_set_lookahead_limit(self.ctx, int(frames_left)), "setLookaheadLimit")

    def setLookahead(self, max_frames):
        """speculative lookahead of the live loop (rt_set_lookahead): consecutive compute(f) calls are traced ahead as batches"""
        self._check(self.L.rt_set_lookahead(self.ctx, int(max_frames)), "setLookahead")

    def setWalk(self, walk):
        """traversal of the wavefront trace kernels: 1 = child-pair records (default), 0 = single nodes; bit-identical"""
        self._check(self.L.rt_set_walk(self.ctx, int(walk)), "setWalk")

    def setKernelTiming(self, enabled):
        self._check(self.L.rt_set_kernel_timing(self.ctx, 1 if enabled else 0), "setKernelTiming")

    def kernelTimeMs(self):
        pt, pv, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint32()
        self._check(self.L.rt_kernel_time_ms(self.ctx, ctypes.byref(pt), ctypes.byref(pv), ctypes.byref(n)),
                    "kernelTimeMs")
        return {"pathtrace_ms": pt.value, "primary_ms": pv.value, "launches": n.value}


TIMER_NAMES = ("primary", "pathtrace", "wf_shade", "wf_trace_shadow", "wf_trace_ext", "post")


def _kernel_times(self):
    """{timer: {"ms": sum of durations, "launches": n}} since the last read (rt_kernel_times)."""
    n = len(TIMER_NAMES)
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_uint32 * n)()
    self._check(self.L.rt_kernel_times(self.ctx, ms, cnt, n), "kernelTimes")
    return {TIMER_NAMES[k]: {"ms": ms[k], "launches": int(cnt[k])} for k in range(n)}


WebGPURenderer.kernelTimes = _kernel_times


def upload_scene(renderer, bridge, width, height):
    """The reference's scene-load sequence (src/main.ts:51-67,99-116): textures, geometry, BVH,
    topology, instances, lights, draw commands, uniforms, then resolution + reset.
    Works for any object with the WebGPURenderer method surface (the oracle binding too)."""
    renderer.loadTexturesFromWorld(bridge)
    renderer.updateCombinedGeometry(bridge.vertices, bridge.normals, bridge.uvs)
    renderer.updateCombinedBVH(bridge.tlas, bridge.blas)
    renderer.updateBuffer("topology", bridge.mesh_topology)
    renderer.updateBuffer("instance", bridge.instances)
    renderer.updateBuffer("lights", bridge.lights)
    renderer.updateBuffer("draw_commands", bridge.draw_commands)
    renderer.updateScreenSize(width, height)
    bridge.updateCamera(width, height)
    renderer.updateSceneUniforms(bridge.cameraData, 0, bridge.lightCount)
    renderer.recreateBindGroup()
    renderer.resetAccumulation()


def sync_world(renderer, bridge, width, height):
    """The per-frame scene sync of the live loop (src/main.ts:133-163): when the bridge has new data, re-upload BVH,
    instances, draw commands and — if the geometry changed — vertices, topology and lights; then camera uniforms,
    rebind if a buffer grew, reset the accumulation.  Returns True when something was uploaded."""
    if not bridge.hasNewData:
        return False
    rebind = False
    if getattr(bridge, "deviceResident", False):
        # update(t) ran inside the renderer (WorldBridge.setDeviceUpdater -> rt_world_update): every array is already
        # where the kernels read it; only the camera uniforms and the accumulation restart remain of main.ts:133-163
        bridge.hasNewGeometry = False
        bridge.updateCamera(width, height)
        renderer.updateSceneUniforms(bridge.cameraData, 0, bridge.lightCount)
        renderer.resetAccumulation()
        bridge.hasNewData = False
        return True
    rebind |= bool(renderer.updateCombinedBVH(bridge.tlas, bridge.blas))
    rebind |= bool(renderer.updateBuffer("instance", bridge.instances))
    rebind |= bool(renderer.updateBuffer("draw_commands", bridge.draw_commands))
    if bridge.hasNewGeometry:
        rebind |= bool(renderer.updateCombinedGeometry(bridge.vertices, bridge.normals, bridge.uvs))
        rebind |= bool(renderer.updateBuffer("topology", bridge.mesh_topology))
        rebind |= bool(renderer.updateBuffer("lights", bridge.lights))
        bridge.hasNewGeometry = False
    bridge.updateCamera(width, height)
    renderer.updateSceneUniforms(bridge.cameraData, 0, bridge.lightCount)
    if rebind:
        renderer.recreateBindGroup()
    renderer.resetAccumulation()
    bridge.hasNewData = False
    return True


class LiveLoop:
    """`renderFrame` of src/main.ts:119-181 without the browser: every `update_interval` frames the world advances to
    t = totalFrameCount / update_interval / 60 (animation + rebuild), the scene is re-synced, accumulation restarts;
    every call traces one frame and presents."""

    def __init__(self, renderer, bridge, width, height, update_interval=0, lookahead=32):
        self.renderer, self.bridge = renderer, bridge
        if lookahead > 1 and hasattr(renderer, "setLookahead"):
            renderer.setLookahead(lookahead)    # a still scene accumulates frame after frame: trace them ahead in batches
        self.width, self.height = width, height
        self.update_interval = int(update_interval)      # <= 0: the world is never advanced (main.ts:127)
        self.frameCount = 0
        self.totalFrameCount = 0

    def render_frame(self):
        if self.update_interval > 0 and self.frameCount >= self.update_interval:
            self.bridge.update(self.totalFrameCount / (self.update_interval or 1) / 60)
        if sync_world(self.renderer, self.bridge, self.width, self.height):
            self.frameCount = 0
        self.frameCount += 1
        self.totalFrameCount += 1
        if self.update_interval > 0 and hasattr(self.renderer, "setLookaheadLimit"):
            # the world moves again in update_interval - frameCount + 1 frames: nothing is traced ahead past that
            self.renderer.setLookaheadLimit(max(1, self.update_interval - self.frameCount + 1))
        self.renderer.compute(self.frameCount)
        self.renderer.present()
