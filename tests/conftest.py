import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def W():
    import webgpu_raytracer_amd as pkg
    pkg._build.build_scene()
    return pkg


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_lib as ol
    ol.build_oracle()
    return ol


@pytest.fixture()
def gpu_renderer(W):
    """A fresh context per test: totalFrames (the Halton jitter index) is renderer-lifetime state
    (WebGPURenderer.ts:15,89), so parity with a fresh oracle needs a fresh renderer."""
    W._build.build_rt()
    r = W.WebGPURenderer(0)
    yield r
    r.destroy()
