'use strict';
// index.js — Node host: the reference's WebGPURenderer / WorldBridge class surface
// (src/renderer/WebGPURenderer.ts:7-138, src/world-bridge.ts:4-216) over the N-API addon.
// Plain CommonJS so that the Node 12 in this image runs it; index.d.ts carries the TypeScript
// signatures of the reference classes.
const path = require('path');
const native = require(path.join(__dirname, 'mi355rt.node'));

const KIND = { topology: 0, instance: 1, lights: 2, draw_commands: 3 };
const RT_REALLOCATED = 1;

// PNG container (colour type 6, filter 0) around zlib — stands in for the encoded images a glTF would carry
const zlib = require('zlib');
const CRC_TABLE = (() => {
  const t = new Uint32Array(256);
  for (let i = 0; i < 256; i++) {
    let c = i;
    for (let k = 0; k < 8; k++) c = (c & 1) ? (0xedb88320 ^ (c >>> 1)) : (c >>> 1);
    t[i] = c >>> 0;
  }
  return t;
})();
function crc32(buf) {
  let c = 0xffffffff;
  for (let i = 0; i < buf.length; i++) c = CRC_TABLE[(c ^ buf[i]) & 255] ^ (c >>> 8);
  return (c ^ 0xffffffff) >>> 0;
}
function pngChunk(tag, data) {
  const out = Buffer.alloc(12 + data.length);
  out.writeUInt32BE(data.length, 0);
  out.write(tag, 4, 'latin1');
  Buffer.from(data.buffer, data.byteOffset, data.length).copy(out, 8);
  out.writeUInt32BE(crc32(out.subarray(4, 8 + data.length)), 8 + data.length);
  return out;
}
function encodePng(rgba, width, height) {
  const rows = Buffer.alloc(height * (1 + width * 4));
  for (let y = 0; y < height; y++)
    Buffer.from(rgba.buffer, rgba.byteOffset + y * width * 4, width * 4).copy(rows, y * (1 + width * 4) + 1);
  const ihdr = Buffer.alloc(13);
  ihdr.writeUInt32BE(width, 0);
  ihdr.writeUInt32BE(height, 4);
  ihdr[8] = 8; ihdr[9] = 6;
  return new Uint8Array(Buffer.concat([Buffer.from([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]), pngChunk('IHDR', ihdr),
    pngChunk('IDAT', zlib.deflateSync(rows, { level: 1 })), pngChunk('IEND', Buffer.alloc(0))]));
}

class WebGPURenderer {
  // `new WebGPURenderer(canvas)`: the canvas becomes a device ordinal (there is no swap chain).
  constructor(device = 0) {
    this._device = device;
    this._ctx = null;
    this.width = 0;
    this.height = 0;
    this._capture = null;
  }
  // only init() throws in the reference (WebGPUContext.ts:15,19)
  async init() {
    this._ctx = native.rtCreate(this._device);
  }
  get device() {
    const self = this;
    return { queue: { onSubmittedWorkDone: async () => { self._check(native.rtSync(self._ctx), 'sync'); } } };
  }
  _check(rc, what) {
    if (rc < 0) throw new Error(`${what} failed (${rc}): ${native.rtLastError(this._ctx)}`);
    return rc;
  }
  buildPipeline(depth, spp) { this._check(native.rtSetPipeline(this._ctx, depth, spp), 'buildPipeline'); }
  updateScreenSize(width, height) {
    this.width = width;
    this.height = height;
    this._check(native.rtResize(this._ctx, width, height), 'updateScreenSize');
  }
  resetAccumulation() { this._check(native.rtResetAccum(this._ctx), 'resetAccumulation'); }
  // ResourceManager.ts:153-198: encoded image per texture -> decode (host) -> 1024x1024 layer (GPU resize);
  // an image that is missing or does not decode becomes the white fallback bitmap and a warning (:169-175)
  async loadTexturesFromWorld(bridge) {
    const n = bridge.textureCount;
    if (n === 0) { this._check(native.rtUploadTextures(this._ctx, null, 0), 'loadTexturesFromWorld'); return; }
    this._check(native.rtAllocTextureLayers(this._ctx, n), 'loadTexturesFromWorld');
    this.textureWarnings = [];
    for (let i = 0; i < n; i++) {
      const data = bridge.getTexture(i);
      let img = null;
      if (data) {
        try { img = native.mtDecode(data); } catch (e) { this.textureWarnings.push(`Failed tex ${i}: ${e.message}`); }
      }
      this._check(native.rtUploadTextureImage(this._ctx, i, img ? img.data : null, img ? img.width : 0, img ? img.height : 0),
        'loadTexturesFromWorld');
    }
  }
  updateBuffer(type, data) {
    return this._check(native.rtUpload(this._ctx, KIND[type], data), `updateBuffer(${type})`) === RT_REALLOCATED;
  }
  updateCombinedGeometry(v, n, uv) {
    return this._check(native.rtUploadGeometry(this._ctx, v, n, uv), 'updateCombinedGeometry') === RT_REALLOCATED;
  }
  updateCombinedBVH(tlas, blas) {
    return this._check(native.rtUploadBVH(this._ctx, tlas, blas), 'updateCombinedBVH') === RT_REALLOCATED;
  }
  updateSceneUniforms(cameraData, frameCount, lightCount) {
    this._check(native.rtSetScene(this._ctx, cameraData, frameCount, lightCount), 'updateSceneUniforms');
  }
  recreateBindGroup() {}
  compute(frameCount) { this._check(native.rtCompute(this._ctx, frameCount), 'compute'); }
  // the recorder's `for (k < batch) compute(samplesDone + k)` as one dispatch per kernel (bit-identical)
  computeBatch(frameCounts) { this._check(native.rtComputeBatch(this._ctx, Uint32Array.from(frameCounts)), 'computeBatch'); }
  present() { this._check(native.rtPresent(this._ctx), 'present'); }
  async captureFrame() {
    if (!this.width) throw new Error('No render target');
    const n = this.width * this.height * 4;
    if (!this._capture || this._capture.length !== n) this._capture = new Uint8Array(n);  // reused between calls
    this._check(native.rtCapture(this._ctx, this._capture), 'captureFrame');
    return { data: this._capture.buffer, width: this.width, height: this.height };
  }
  // additions
  readAccum() {
    const out = new Float32Array(this.width * this.height * 4);
    this._check(native.rtReadAccum(this._ctx, out), 'readAccum');
    return out;
  }
  getCounters() {
    const c = native.rtGetCounters(this._ctx);
    return { primary_rays: c[0], extension_rays: c[1], shadow_rays: c[2], nodes_visited: c[3], tris_tested: c[4], shaded_hits: c[5] };
  }
  destroy() { if (this._ctx) { native.rtDestroy(this._ctx); this._ctx = null; } }
}

class WorldBridge {
  constructor() { this._w = null; this._cache = {}; this.hasNewData = false; this.hasNewGeometry = false; this._wh = [-1, -1]; }
  async initWasm() {}
  // world-bridge.ts:109-130; glbData: Uint8Array of a .glb (or .gltf JSON with data URIs)
  async loadScene(sceneName, objSource, glbData) {
    if (this._w) native.msDestroy(this._w);
    this._w = native.msCreate(sceneName, objSource === undefined ? null : objSource, glbData || null);
    // a GLB that does not parse leaves the procedural scene alone (lib.rs:57-67 ignores the error); keep the reason
    this.loadWarning = glbData ? native.msLastError() : '';
    if (this._blasRenderer) native.msSetBlasBuilder(this._w, this._blasRenderer._ctx);
    if (this._deviceRenderer) native.msSetDeviceUpdater(this._w, this._deviceRenderer._ctx);
    this.deviceResident = false;
    this._wh = [-1, -1];
    this._refresh();
    this.hasNewData = true;
    this.hasNewGeometry = true;
  }
  update(time) {
    if (this._blasRenderer && !this._blasRenderer._ctx) throw new Error('the renderer set with setBlasBuilder() has been destroyed');
    if (this._deviceRenderer && !this._deviceRenderer._ctx) throw new Error('the renderer set with setDeviceUpdater() has been destroyed');
    native.msUpdate(this._w, time);
    this.deviceResident = !!native.msDeviceResident(this._w);
    this.deviceWarning = (this._deviceRenderer && !this.deviceResident) ? native.msLastError() : '';
    if (!this._deviceRenderer && this._blasRenderer) {
      const err = native.msLastError();
      if (err) throw new Error(err);   // the GPU builder failed: no silent CPU result
    }
    if (!this.deviceResident) this._refresh();   // a device-resident update leaves the host arrays alone
    this.hasNewData = true; this.hasNewGeometry = true;
  }
  // run the whole per-frame half of update(t) on the GPU, inside `renderer`'s buffers (rt_world_update): skinning, BLAS
  // builds, topology / light / draw-command packing, TLAS, instances; syncWorld() then uploads nothing.  null = host path
  setDeviceUpdater(renderer) {
    this._deviceRenderer = renderer || null;
    if (this._w) native.msSetDeviceUpdater(this._w, renderer ? renderer._ctx : null);
    if (!renderer) this.deviceResident = false;
  }
  // build the BLASes of update(t) on the GPU (rt_build_blas: the CPU builder's tree, byte for byte); null = CPU builder
  setBlasBuilder(renderer) {
    this._blasRenderer = renderer || null;
    if (this._w) native.msSetBlasBuilder(this._w, renderer ? renderer._ctx : null);
  }
  updateCamera(width, height) {
    if (this._wh[0] === width && this._wh[1] === height) return;
    this._wh = [width, height];
    native.msUpdateCamera(this._w, width, height);
    this._cache.camera = native.msGet(this._w, 'camera');
  }
  _refresh() {
    for (const k of ['vertices', 'normals', 'uvs', 'mesh_topology', 'tlas', 'blas', 'instances', 'lights', 'draw_commands', 'camera'])
      this._cache[k] = native.msGet(this._w, k);
  }
  get vertices() { return this._cache.vertices; }
  get normals() { return this._cache.normals; }
  get uvs() { return this._cache.uvs; }
  get mesh_topology() { return this._cache.mesh_topology; }
  get tlas() { return this._cache.tlas; }
  get blas() { return this._cache.blas; }
  get instances() { return this._cache.instances; }
  get lights() { return this._cache.lights; }
  get lightCount() { return this._cache.lights.length / 2; }
  get draw_commands() { return this._cache.draw_commands; }
  get cameraData() { return this._cache.camera; }
  get textureCount() {
    if (!this._w) return 0;
    return native.msEncodedTextureCount(this._w) || native.msTextureCount(this._w);
  }
  // world-bridge.ts:98-99, 161-170
  getAnimationList() { return this._w ? native.msAnimationNames(this._w) : []; }
  loadAnimation(data) { return native.msLoadAnimation(this._w, data); }
  setAnimation(index) { native.msSetAnimation(this._w, index); }
  get hasWorld() { return !!this._w && this._cache.vertices.length > 0; }
  getTextureRGBA(i) { return native.msTexture(this._w, i); }
  // world-bridge.ts:101-106 hands out ENCODED images; the synthetic scenes hold raw texels, so encode them as PNG
  getTexture(i) {
    if (native.msEncodedTextureCount(this._w)) return native.msEncodedTexture(this._w, i);   // glTF input
    const rgba = this.getTextureRGBA(i);
    return rgba ? encodePng(rgba, 1024, 1024) : undefined;
  }
}

// src/main.ts:133-163: the per-frame scene sync of the live loop. Returns true when something was uploaded.
function syncWorld(renderer, bridge, width, height) {
  if (!bridge.hasNewData) return false;
  let rebind = false;
  if (bridge.deviceResident) {   // update(t) ran inside the renderer: only the camera uniforms and the restart remain
    bridge.hasNewGeometry = false;
    bridge.updateCamera(width, height);
    renderer.updateSceneUniforms(bridge.cameraData, 0, bridge.lightCount);
    renderer.resetAccumulation();
    bridge.hasNewData = false;
    return true;
  }
  rebind = renderer.updateCombinedBVH(bridge.tlas, bridge.blas) || rebind;
  rebind = renderer.updateBuffer('instance', bridge.instances) || rebind;
  rebind = renderer.updateBuffer('draw_commands', bridge.draw_commands) || rebind;
  if (bridge.hasNewGeometry) {
    rebind = renderer.updateCombinedGeometry(bridge.vertices, bridge.normals, bridge.uvs) || rebind;
    rebind = renderer.updateBuffer('topology', bridge.mesh_topology) || rebind;
    rebind = renderer.updateBuffer('lights', bridge.lights) || rebind;
    bridge.hasNewGeometry = false;
  }
  bridge.updateCamera(width, height);
  renderer.updateSceneUniforms(bridge.cameraData, 0, bridge.lightCount);
  if (rebind) renderer.recreateBindGroup();
  renderer.resetAccumulation();
  bridge.hasNewData = false;
  return true;
}

// `renderFrame` of src/main.ts:119-181 without requestAnimationFrame: every updateInterval frames the world advances to
// t = totalFrameCount / updateInterval / 60, the scene is re-synced and the accumulation restarts; each call traces and presents
class LiveLoop {
  constructor(renderer, bridge, width, height, updateInterval = 0) {
    Object.assign(this, { renderer, bridge, width, height, updateInterval, frameCount: 0, totalFrameCount: 0 });
  }
  renderFrame() {
    if (this.updateInterval > 0 && this.frameCount >= this.updateInterval)
      this.bridge.update(this.totalFrameCount / (this.updateInterval || 1) / 60);
    if (syncWorld(this.renderer, this.bridge, this.width, this.height)) this.frameCount = 0;
    this.frameCount++;
    this.totalFrameCount++;
    this.renderer.compute(this.frameCount);
    this.renderer.present();
  }
}

module.exports = { WebGPURenderer, WorldBridge, LiveLoop, syncWorld, native };
