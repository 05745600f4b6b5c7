"""WorldBridge — host-side mirror of src/world-bridge.ts over the native scene compiler.

The reference's WorldBridge is an async proxy to a Web Worker that owns the WASM `World`
(src/world-bridge.ts:4-216, src/worker/wasm-worker.ts).  Here the `World` is the C++ scene
compiler behind include/mi355scene.h and calls are synchronous; the getter names, array
dtypes and element strides are the reference's (src/worker/protocol.ts:14-44).
"""
import ctypes
import os

import numpy as np

from . import _build

_F32P = ctypes.POINTER(ctypes.c_float)
_U32P = ctypes.POINTER(ctypes.c_uint32)
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.SCENE_LIB
    if not os.path.exists(path):
        _build.build_scene()
    lib = ctypes.CDLL(path)
    lib.ms_world_create.restype = ctypes.c_void_p
    lib.ms_world_create.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.ms_world_destroy.argtypes = [ctypes.c_void_p]
    lib.ms_last_error.restype = ctypes.c_char_p
    lib.ms_world_update.argtypes = [ctypes.c_void_p, ctypes.c_float]
    lib.ms_world_update_camera.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
    for name in ("vertices", "normals", "uvs", "tlas", "blas", "instances", "camera"):
        f = getattr(lib, "ms_world_" + name)
        f.restype = _F32P
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    for name in ("mesh_topology", "lights", "draw_commands"):
        f = getattr(lib, "ms_world_" + name)
        f.restype = _U32P
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    lib.ms_world_create_glb.restype = ctypes.c_void_p
    lib.ms_world_create_glb.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    for name in ("animation_count", "node_count", "encoded_texture_count"):
        f = getattr(lib, "ms_world_" + name)
        f.restype = ctypes.c_size_t
        f.argtypes = [ctypes.c_void_p]
    lib.ms_world_animation_name.restype = ctypes.c_char_p
    lib.ms_world_animation_name.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.ms_world_set_animation.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.ms_world_load_animation_glb.restype = ctypes.c_int
    lib.ms_world_load_animation_glb.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.ms_world_encoded_texture.restype = ctypes.POINTER(ctypes.c_uint8)
    lib.ms_world_encoded_texture.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    lib.ms_world_set_blas_builder.restype = None
    lib.ms_world_set_blas_builder.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.ms_world_set_device_updater.restype = None
    lib.ms_world_set_device_updater.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.ms_world_device_resident.restype = ctypes.c_int
    lib.ms_world_device_resident.argtypes = [ctypes.c_void_p]
    lib.ms_world_texture_count.restype = ctypes.c_size_t
    lib.ms_world_texture_count.argtypes = [ctypes.c_void_p]
    lib.ms_world_texture_rgba.restype = ctypes.POINTER(ctypes.c_uint8)
    lib.ms_world_texture_rgba.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    _lib = lib
    return lib


class WorldBridge:
    """Same data accessors as the reference class (world-bridge.ts:172-205)."""

    def __init__(self, zero_copy=False):
        """zero_copy=True hands out views into the scene compiler's arrays instead of copies — what the reference's
        getters do with wasm memory (world-bridge.ts:172-205): valid until the next update() / loadScene() / close()."""
        self._zero_copy = bool(zero_copy)
        self._lib = _load()
        self._world = None
        self._cache = {}
        self._tex_blobs = {}
        self._blas_renderer = None
        self._device_renderer = None
        self._device_fn = None
        self.deviceResident = False      # the last update(t) ran on the device: the arrays below were NOT refreshed
        self.hasNewData = False
        self.hasNewGeometry = False
        self._last_wh = (-1, -1)

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_world", None):
            self._lib.ms_world_destroy(self._world)
            self._world = None

    # world-bridge.ts:109-130
    def loadScene(self, sceneName, objSource=None, glbData=None):
        self.close()
        obj = objSource.encode() if isinstance(objSource, str) else objSource
        if glbData is not None:
            glb = bytes(glbData)
            w = self._lib.ms_world_create_glb(sceneName.encode(), obj, glb, len(glb))
        else:
            w = self._lib.ms_world_create(sceneName.encode(), obj)
        if not w:
            raise ValueError(self._lib.ms_last_error().decode())
        # a GLB that does not parse leaves the procedural scene alone (lib.rs:57-67 ignores the error); keep the reason
        self.loadWarning = self._lib.ms_last_error().decode() if glbData is not None else ""
        self._world = w
        self.deviceResident = False
        self._apply_blas_builder()
        self._apply_device_updater()
        self._tex_blobs = {}
        self._last_wh = (-1, -1)
        self._refresh()
        self.hasNewData = True
        self.hasNewGeometry = True

    def setBlasBuilder(self, renderer):
        """Run the per-geometry BLAS build of update(t) on the GPU: `renderer` is a WebGPURenderer whose rt_build_blas
        returns the CPU builder's tree byte for byte; None restores the CPU builder.  The hook is a
        C function pointer handed to the scene compiler — no Python in the loop."""
        self._blas_renderer = renderer
        self._apply_blas_builder()

    def _apply_blas_builder(self):
        if not self._world:
            return
        r = self._blas_renderer
        if r is None or not r.ctx:
            self._lib.ms_world_set_blas_builder(self._world, None, None)
        else:
            fn = ctypes.cast(r.L.rt_build_blas, ctypes.c_void_p)
            self._lib.ms_world_set_blas_builder(self._world, fn, r.ctx)

    def setDeviceUpdater(self, renderer, fn=None):
        """Run the whole per-frame half of update(t) on the GPU, inside `renderer`'s scene buffers (rt_world_update,
        SURVEY.md §8f N1): skinning, BLAS builds, topology / light / draw-command packing, TLAS, instances.  update(t)
        then only samples the animation and hands the joint matrices over; the host arrays behind the getters are not
        refreshed (`deviceResident` is True, sync_world() uploads nothing) — renderer.worldRead(name) reads one back.
        None restores the host path.  `fn` (tests): a ctypes callback with ms_device_updater's signature instead of
        the renderer's rt_world_update."""
        self._device_renderer = renderer
        self._device_fn = fn
        self._apply_device_updater()

    def _apply_device_updater(self):
        if not self._world:
            return
        r = self._device_renderer
        if self._device_fn is not None:
            self._lib.ms_world_set_device_updater(self._world, ctypes.cast(self._device_fn, ctypes.c_void_p), None)
        elif r is None or not r.ctx:
            self._lib.ms_world_set_device_updater(self._world, None, None)
        else:
            self._lib.ms_world_set_device_updater(self._world, ctypes.cast(r.L.rt_world_update, ctypes.c_void_p), r.ctx)

    # world-bridge.ts:141-145
    def update(self, time):
        if self._blas_renderer is not None and not self._blas_renderer.ctx:
            raise RuntimeError("the renderer set with setBlasBuilder() has been destroyed")
        if self._device_renderer is not None and self._device_fn is None and not self._device_renderer.ctx:
            raise RuntimeError("the renderer set with setDeviceUpdater() has been destroyed")
        self._lib.ms_world_update(self._world, float(time))
        self.deviceResident = bool(self._lib.ms_world_device_resident(self._world))
        self.deviceWarning = ""
        if self._device_renderer is not None or self._device_fn is not None:
            if not self.deviceResident:      # the device path refused this scene / frame: the host path ran, say why
                self.deviceWarning = self._lib.ms_last_error().decode()
                if self._device_renderer is not None and self._device_renderer.ctx:
                    self.deviceWarning += ": " + self._device_renderer.L.rt_last_error(self._device_renderer.ctx).decode()
        elif self._blas_renderer is not None:
            err = self._lib.ms_last_error().decode()
            if err:
                raise RuntimeError(err)      # the GPU builder failed: no silent CPU result
        if not self.deviceResident:
            self._refresh()
        self.hasNewData = True
        self.hasNewGeometry = True

    # world-bridge.ts:150-161
    def updateCamera(self, width, height):
        if self._last_wh == (width, height):
            return
        self._last_wh = (width, height)
        self._lib.ms_world_update_camera(self._world, float(width), float(height))
        self._cache["camera"] = self._get("camera", np.float32)

    def _get(self, name, dtype):
        n = ctypes.c_size_t()
        p = getattr(self._lib, "ms_world_" + name)(self._world, ctypes.byref(n))
        if n.value == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(p, shape=(n.value,)).astype(dtype, copy=not self._zero_copy)

    def _refresh(self):
        for name in ("vertices", "normals", "uvs", "tlas", "blas", "instances", "camera"):
            self._cache[name] = self._get(name, np.float32)
        for name in ("mesh_topology", "lights", "draw_commands"):
            self._cache[name] = self._get(name, np.uint32)

    vertices = property(lambda s: s._cache["vertices"])
    normals = property(lambda s: s._cache["normals"])
    uvs = property(lambda s: s._cache["uvs"])
    mesh_topology = property(lambda s: s._cache["mesh_topology"])
    tlas = property(lambda s: s._cache["tlas"])
    blas = property(lambda s: s._cache["blas"])
    instances = property(lambda s: s._cache["instances"])
    lights = property(lambda s: s._cache["lights"])
    draw_commands = property(lambda s: s._cache["draw_commands"])
    cameraData = property(lambda s: s._cache["camera"])

    @property
    def lightCount(self):
        return len(self._cache["lights"]) // 2

    @property
    def hasWorld(self):
        return self._world is not None and len(self._cache.get("vertices", ())) > 0

    # world-bridge.ts:98-99, 161-170
    def getAnimationList(self):
        n = int(self._lib.ms_world_animation_count(self._world)) if self._world else 0
        return [self._lib.ms_world_animation_name(self._world, i).decode() for i in range(n)]

    def loadAnimation(self, data):
        b = bytes(data)
        return int(self._lib.ms_world_load_animation_glb(self._world, b, len(b)))

    def setAnimation(self, index):
        self._lib.ms_world_set_animation(self._world, int(index))

    @property
    def nodeCount(self):
        return int(self._lib.ms_world_node_count(self._world)) if self._world else 0

    @property
    def textureCount(self):
        if not self._world:
            return 0
        n = int(self._lib.ms_world_encoded_texture_count(self._world))   # glTF input: encoded images
        return n if n else int(self._lib.ms_world_texture_count(self._world))

    def getTexture(self, index):
        """Encoded image bytes of texture `index` (world-bridge.ts:101-106, `World::get_texture_ptr/_len`
        lib.rs:359-381).  The synthetic scenes only hold raw texels, so the blob is their PNG encoding (cached)."""
        if index in self._tex_blobs:
            return self._tex_blobs[index]
        if self._world and int(self._lib.ms_world_encoded_texture_count(self._world)):
            n = ctypes.c_size_t()
            p = self._lib.ms_world_encoded_texture(self._world, index, ctypes.byref(n))
            if not p or n.value == 0:
                return None                    # external / missing image: the renderer writes the white fallback layer
            blob = bytes(np.ctypeslib.as_array(p, shape=(n.value,)))
            self._tex_blobs[index] = blob
            return blob
        rgba = self.getTextureRGBA(index)
        if rgba is None:
            return None
        from .textures import encode_png
        blob = encode_png(rgba)
        self._tex_blobs[index] = blob
        return blob

    def getTextureRGBA(self, index):
        """Decoded 1024x1024 RGBA8 layer (synthetic scenes): what decoding + resizing getTexture(index) yields."""
        p = self._lib.ms_world_texture_rgba(self._world, index)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(1024, 1024, 4)).copy()

    def printStats(self):  # world-bridge.ts:207-215 (with the real strides)
        return "V=%d, Topo=%d, I=%d, TLAS=%d, BLAS=%d, Lights=%d" % (
            len(self.vertices) // 4, len(self.mesh_topology) // 20, len(self.instances) // 36,
            len(self.tlas) // 8, len(self.blas) // 8, self.lightCount)
