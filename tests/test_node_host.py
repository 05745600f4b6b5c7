"""The Node host (N-API addon + index.js) — north_star asks for the host side in the reference's own
language.  CPU: the addon loads under the image's Node and its WorldBridge yields byte-identical bridge
arrays.  GPU: a headless render driven from JavaScript is bit-identical to the CPU oracle."""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE_DIR = os.path.join(REPO, "webgpu-raytracer_amd", "node")

node = shutil.which("node")
needs_node = pytest.mark.skipif(node is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node / node_api.h not present")


@pytest.fixture(scope="module")
def addon(W):
    W._build.build_rt()
    path = W._build.build_node_addon()
    assert path and os.path.exists(path)
    return path


@needs_node
def test_addon_loads_and_world_arrays_match_python(W, addon):
    js = ("const {WorldBridge}=require('%s/index.js');const c=require('crypto');(async()=>{const b=new WorldBridge();"
          "await b.loadScene('cornell');b.updateCamera(512,512);const o={};"
          "for(const k of ['vertices','normals','uvs','mesh_topology','tlas','blas','instances','lights','draw_commands','cameraData'])"
          "o[k]=c.createHash('sha256').update(Buffer.from(b[k].buffer)).digest('hex');console.log(JSON.stringify(o));})()"
          % NODE_DIR)
    out = subprocess.run([node, "-e", js], check=True, capture_output=True, text=True).stdout
    got = json.loads(out.strip().splitlines()[-1])
    b = W.WorldBridge()
    b.loadScene("cornell")
    b.updateCamera(512, 512)
    for k, digest in got.items():
        assert hashlib.sha256(np.ascontiguousarray(getattr(b, k)).tobytes()).hexdigest() == digest, k


@needs_node
@pytest.mark.gpu
def test_javascript_driven_render_matches_the_oracle(W, oracle_lib, addon):
    env = dict(os.environ, RT_NODE_BATCH="4")   # 3 live-loop frames, then frames 4..7 through computeBatch + present
    out = subprocess.run([node, os.path.join(NODE_DIR, "render_cornell.js"), "cornell", "96", "80", "3", "4"],
                         check=True, capture_output=True, text=True, timeout=300, env=env).stdout
    got = json.loads(out.strip().splitlines()[-1])
    b = W.WorldBridge()
    b.loadScene("cornell")
    cpu = oracle_lib.OracleRenderer()
    cpu.buildPipeline(4, 1)
    W.upload_scene(cpu, b, 96, 80)
    for f in (1, 2, 3):
        cpu.compute(f)
        cpu.present()
    for f in (4, 5, 6, 7):
        cpu.compute(f)
    cpu.present()
    assert hashlib.sha256(cpu.readAccum().tobytes()).hexdigest() == got["accum_sha256"]
    assert hashlib.sha256(cpu.captureFrame()["data"].tobytes()).hexdigest() == got["rgba_sha256"]
    c = cpu.getCounters()
    assert got["counters"]["primary_rays"] == c["primary_rays"]
    assert got["counters"]["extension_rays"] == c["extension_rays"]
    assert got["counters"]["shadow_rays"] == c["shadow_rays"]


@needs_node
def test_javascript_bridge_hands_out_png_that_decodes_to_the_layer(W, addon):
    """world-bridge.ts getTexture(): encoded bytes. The JS bridge's PNG goes through mtDecode (the addon's binding of
    mt_decode) and must give back the scene's raw layer."""
    js = ("const {WorldBridge}=require('%s/index.js');const n=require('%s/mi355rt.node');const c=require('crypto');"
          "(async()=>{const b=new WorldBridge();await b.loadScene('sponza_like');const png=b.getTexture(2);"
          "const img=n.mtDecode(png);let bad=null;try{n.mtDecode(new Uint8Array([1,2,3]))}catch(e){bad=e.message}"
          "console.log(JSON.stringify({count:b.textureCount,w:img.width,h:img.height,bad,"
          "sha:c.createHash('sha256').update(Buffer.from(img.data.buffer)).digest('hex'),"
          "magic:Buffer.from(png.buffer,png.byteOffset,4).toString('latin1')}));})()" % (NODE_DIR, NODE_DIR))
    out = subprocess.run([node, "-e", js], check=True, capture_output=True, text=True, timeout=300).stdout
    got = json.loads(out.strip().splitlines()[-1])
    b = W.WorldBridge()
    b.loadScene("sponza_like")
    assert got["count"] == 8 and got["w"] == 1024 and got["h"] == 1024 and got["magic"] == "\x89PNG"
    assert got["bad"] == "not a PNG or JPEG image"
    assert hashlib.sha256(b.getTextureRGBA(2).tobytes()).hexdigest() == got["sha"]


@needs_node
@pytest.mark.gpu
def test_javascript_textured_render_matches_the_oracle(W, oracle_lib, addon):
    """sponza_like from JS: loadTexturesFromWorld = PNG -> mtDecode -> rtUploadTextureImage (GPU resize)."""
    env = dict(os.environ, RT_NODE_GPU_BLAS="1")       # and the 263k-triangle BLAS rebuilt by rt_build_blas through the bridge hook
    out = subprocess.run([node, os.path.join(NODE_DIR, "render_cornell.js"), "sponza_like", "64", "40", "2", "5"],
                         check=True, capture_output=True, text=True, timeout=600, env=env).stdout
    got = json.loads(out.strip().splitlines()[-1])
    b = W.WorldBridge()
    b.loadScene("sponza_like")
    cpu = oracle_lib.OracleRenderer()
    cpu.buildPipeline(5, 1)
    W.upload_scene(cpu, b, 64, 40)
    for f in (1, 2):
        cpu.compute(f)
        cpu.present()
    assert hashlib.sha256(cpu.readAccum().tobytes()).hexdigest() == got["accum_sha256"]
    assert hashlib.sha256(cpu.captureFrame()["data"].tobytes()).hexdigest() == got["rgba_sha256"]


@needs_node
def test_javascript_bridge_loads_a_glb_like_the_python_bridge(W, addon, tmp_path):
    """loadScene(name, undefined, glbData) + update(t) + animation API from JS give the arrays the Python bridge gives."""
    import test_gltf
    b, _ = test_gltf.build_skinned(W)
    b.image_texture(W.textures.encode_png(np.full((4, 4, 4), 200, np.uint8)))
    glb = b.glb()
    path = tmp_path / "strip.glb"
    path.write_bytes(glb)
    js = ("const {WorldBridge}=require('%s/index.js');const c=require('crypto');const fs=require('fs');(async()=>{"
          "const b=new WorldBridge();await b.loadScene('viewer',undefined,new Uint8Array(fs.readFileSync('%s')));"
          "b.setAnimation(0);b.update(0.3);b.updateCamera(64,48);const o={anims:b.getAnimationList(),warn:b.loadWarning,tex:b.textureCount,"
          "texlen:b.getTexture(0).length};"
          "for(const k of ['vertices','normals','mesh_topology','tlas','blas','instances','lights'])"
          "o[k]=c.createHash('sha256').update(Buffer.from(b[k].buffer)).digest('hex');"
          "const bad=new WorldBridge();await bad.loadScene('viewer',undefined,new Uint8Array([1,2,3,4]));o.badwarn=bad.loadWarning;"
          "console.log(JSON.stringify(o));})()" % (NODE_DIR, path))
    out = subprocess.run([node, "-e", js], check=True, capture_output=True, text=True, timeout=300).stdout
    got = json.loads(out.strip().splitlines()[-1])
    br = W.WorldBridge()
    br.loadScene("viewer", glbData=glb)
    br.update(0.3)
    assert got["anims"] == ["bend", "anim"] and got["warn"] == "" and got["badwarn"] != "" and got["tex"] == 1
    assert got["texlen"] == len(br.getTexture(0))
    for k in ("vertices", "normals", "mesh_topology", "tlas", "blas", "instances", "lights"):
        assert hashlib.sha256(np.ascontiguousarray(getattr(br, k)).tobytes()).hexdigest() == got[k], k


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["RT_NODE_GPU_BLAS", "RT_NODE_DEVICE_UPDATE"])
def test_javascript_live_loop_on_an_animated_glb_matches_the_oracle(W, oracle_lib, addon, tmp_path, mode):
    """node/animate_glb.js: LiveLoop (main.ts renderFrame) over a skinned, animated, textured GLB with the GPU BLAS builder
    or with the whole update(t) on the device (rt_world_update), against the oracle driven by the Python LiveLoop on the
    same frames."""
    import test_gltf
    b, _ = test_gltf.build_skinned(W)
    b.image_texture(W.textures.encode_png(np.full((4, 4, 4), 180, np.uint8)))
    b.doc["materials"] = [{"pbrMetallicRoughness": {"metallicFactor": 0.0, "baseColorTexture": {"index": 0}}}]
    b.doc["meshes"][0]["primitives"][0]["material"] = 0
    path = tmp_path / "strip.glb"
    path.write_bytes(b.glb())
    env = dict(os.environ, **{mode: "1"})
    out = subprocess.run([node, os.path.join(NODE_DIR, "animate_glb.js"), str(path), "96", "64", "7", "2", "5"],
                         check=True, capture_output=True, text=True, timeout=600, env=env).stdout
    got = json.loads(out.strip().splitlines()[-1])
    assert got["deviceResident"] == (mode == "RT_NODE_DEVICE_UPDATE")
    br = W.WorldBridge()
    br.loadScene("viewer", glbData=b.glb())
    cpu = oracle_lib.OracleRenderer()
    cpu.buildPipeline(5, 1)
    cpu.loadTexturesFromWorld(br)
    cpu.updateScreenSize(96, 64)
    loop = W.LiveLoop(cpu, br, 96, 64, update_interval=2)
    for _ in range(7):
        loop.render_frame()
    assert got["animations"] == ["bend", "anim"] and got["frameCount"] == loop.frameCount
    assert hashlib.sha256(cpu.readAccum().tobytes()).hexdigest() == got["accum_sha256"]
    assert hashlib.sha256(cpu.captureFrame()["data"].tobytes()).hexdigest() == got["rgba_sha256"]
