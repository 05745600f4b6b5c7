"""Known answers for the post pass (PostProcess.wgsl) that do not come from a restatement of its filters: on a UNIFORM
accumulation buffer every stage before the tone map is the identity — firefly clamp (c <= 3c + 0.1), bilinear un-jitter,
bilateral filter (all range weights 1), neighbourhood clamp of the history (mean = c, stddev = 0 pulls the zero history
to c), alpha blend — so the output must be gamma(ACES(c)) and the f16 history c, whatever the frame count.  The expected
bytes are computed here in float64 from the two closed-form curves (PostProcess.wgsl:36-39, 174)."""
import numpy as np
import pytest

import parity_util as pu


def _expected_rgba8(c):
    c = np.asarray(c, dtype=np.float64)
    aces = np.clip((c * (2.51 * c + 0.03)) / (c * (2.43 * c + 0.59) + 0.14), 0.0, 1.0)
    return np.clip(aces, 0.0, 1.0) ** (1.0 / 2.2) * 255.0


@pytest.mark.parametrize("colour", [(0.18, 0.18, 0.18), (0.02, 0.5, 1.0), (4.0, 2.0, 0.25), (0.0, 0.001, 12.0)])
@pytest.mark.parametrize("frame_count", [1, 2, 16, 17, 64])
def test_uniform_image_is_tone_mapped_and_nothing_else(W, oracle_lib, colour, frame_count):
    b = pu.bridge_for(W, "cornell")
    w, h = 40, 24
    r = oracle_lib.OracleRenderer()
    r.buildPipeline(2, 1)
    W.upload_scene(r, b, w, h)
    for f in range(1, frame_count + 1):          # host state (frame_count, average jitter) as after frame_count dispatches
        if f in (1, frame_count):
            r.compute(f)
    acc = np.empty((h, w, 4), dtype=np.float32)
    acc[..., :3] = np.asarray(colour, dtype=np.float32) * np.float32(frame_count)
    acc[..., 3] = frame_count
    r.writeAccum(acc)
    r.present()
    out = r.captureFrame()["data"].reshape(h, w, 4)
    want = _expected_rgba8(np.asarray(colour, dtype=np.float32).astype(np.float64))
    # the accumulated colour is (c * n) / n in f32, the curves run in f32: allow one code value
    assert np.all(np.abs(out[..., :3].astype(np.float64) - np.round(want)[None, None, :]) <= 1.0), (out[0, 0], want)
    assert (out[..., 3] == 255).all()
    # exactly uniform: no stage may introduce spatial structure
    assert (out == out[0, 0]).all()
    # the rgba16f history the next frame blends with is the colour itself (the zero history was clamped to the neighbourhood)
    hist = r.readHistory().reshape(h, w, 4).view(np.float16).astype(np.float64)
    c16 = np.asarray(colour, dtype=np.float64)
    # (beyond 16 frames the clamp window is 60 standard deviations of a neighbourhood whose f32 variance is rounding noise:
    # the zero history lands a few per cent below c instead of on it)
    tol = 2e-3 if frame_count <= 16 else 5e-2
    assert np.all(np.abs(hist[..., :3] - c16[None, None, :]) <= np.maximum(tol * c16, 1e-6)[None, None, :]), (hist[0, 0], c16)
    assert (hist[..., 3] == 1.0).all()
