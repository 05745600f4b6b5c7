"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/mi355rt.h declares; the scene-compiler ABI likewise. No compute calls."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header, prefix):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


@pytest.fixture(scope="module")
def rt_lib(W):
    W._build.build_rt()
    return ctypes.CDLL(W._build.RT_LIB)


def test_rt_library_exports_every_declared_symbol(W, rt_lib):
    names = declared_symbols("mi355rt.h", "rt_")
    assert len(names) >= 30
    for n in names:
        assert hasattr(rt_lib, n), "libmi355rt.so does not export %s" % n
    # the python binding declares exactly the same set
    from webgpu_raytracer_amd import renderer
    assert sorted(renderer.EXPORTED_SYMBOLS) == names
    renderer.load_library()


def test_scene_library_exports_every_declared_symbol(W):
    lib = ctypes.CDLL(W._build.build_scene())
    for n in declared_symbols("mi355scene.h", "ms_"):
        assert hasattr(lib, n), "libmi355scene.so does not export %s" % n


def test_texture_library_exports_every_declared_symbol(W):
    lib = ctypes.CDLL(W._build.build_tex())
    names = declared_symbols("mi355tex.h", "mt_")
    assert len(names) >= 5
    for n in names:
        assert hasattr(lib, n), "libmi355tex.so does not export %s" % n
    assert sorted(W.textures.TEX_EXPORTED_SYMBOLS) == names
    W.textures.load_tex_library()


def test_library_is_gfx950_code_object(W, rt_lib):
    blob = open(W._build.RT_LIB, "rb").read()
    assert b"gfx950" in blob
    for kernel in (b"k_pathtrace", b"k_primary_visibility", b"k_postprocess", b"k_prepare_tris", b"k_resize_texture"):
        assert kernel in blob


def test_renderer_fails_loudly_without_gpu(W):
    """No CPU fallback: without a HIP device the constructor raises instead of degrading."""
    from webgpu_raytracer_amd import renderer
    L = renderer.load_library()
    if L.rt_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(W.RendererError):
        W.WebGPURenderer(0)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, link or name it."""
    pkg = os.path.join(REPO, "webgpu-raytracer_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c", ".js")):
                text = open(os.path.join(d, f), errors="ignore").read()
                assert "rt_oracle" not in text and "oracle_lib" not in text and "oracle/" not in text, \
                    "%s references the oracle" % os.path.join(d, f)


def test_hip_runtime_preload_is_guarded(W, tmp_path):
    """the preload of torch's bundled HIP runtime must not break users who never touch torch.  The ELF reader
    finds what libmi355rt.so NEEDs and a candidate's SONAME without dlopen; a candidate that cannot be loaded, or whose
    SONAME is not the one we need, is skipped with a warning instead of raising or silently leaving two runtimes."""
    import subprocess, sys, os
    from webgpu_raytracer_amd import renderer as R
    W._build.build_rt()
    needed = R._elf_dynamic_strings(W._build.RT_LIB, (1,))[1]
    assert any(n.startswith("libamdhip64.so") for n in needed)
    assert R._elf_dynamic_strings(W._build.SCENE_LIB, (14,))[14] in ([], ["libmi355scene.so"])
    bad = tmp_path / "libamdhip64.so"
    bad.write_bytes(b"not an ELF file")
    code = ("import warnings, sys; sys.path.insert(0, %r)\n"
            "import webgpu_raytracer_amd as W\nfrom webgpu_raytracer_amd import renderer as R\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always')\n"
            "    R.load_library()\n"
            "print('NOTE', R.hip_runtime_note)\nprint('WARN', len([x for x in w if issubclass(x.category, RuntimeWarning)]))\n" % REPO)
    env = dict(os.environ, MI355RT_HIP_RUNTIME=str(bad))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "could not preload" in out.stdout and "WARN 1" in out.stdout
    env = dict(os.environ, MI355RT_HIP_RUNTIME="system")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "MI355RT_HIP_RUNTIME=system" in out.stdout and "WARN 0" in out.stdout
