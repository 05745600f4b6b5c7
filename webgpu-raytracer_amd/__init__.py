"""webgpu-raytracer_amd — MI355X-native path-tracing hot path of kokutoupan/webgpu-raytracer.

Host-side mirror of the reference's renderer interface over a C-ABI HIP library:
  WebGPURenderer  (src/renderer/WebGPURenderer.ts)  -> renderer.WebGPURenderer
  WorldBridge     (src/world-bridge.ts)             -> world_bridge.WorldBridge
The directory name has a hyphen; import it as `webgpu_raytracer_amd` (shim module at the
repo root) or through importlib.
"""
from . import _build
from .world_bridge import WorldBridge
from .renderer import WebGPURenderer, RendererError, upload_scene, sync_world, LiveLoop
from .recorder import FrameLoop, FrameJobRunner, job_list, write_png
from . import textures

__all__ = ["WebGPURenderer", "WorldBridge", "RendererError", "upload_scene", "sync_world", "LiveLoop", "FrameLoop", "FrameJobRunner", "job_list", "write_png", "textures", "build"]


def build(force=False):
    """Compile every native piece in-tree (scene compiler, HIP renderer, N-API addon)."""
    return _build.build_all(force=force)
