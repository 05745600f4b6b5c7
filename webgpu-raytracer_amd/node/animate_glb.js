'use strict';
// Headless live loop over an animated glTF (src/main.ts:119-181): load GLB -> upload -> N x renderFrame with the world
// advanced every `interval` frames (GPU BLAS builder behind update(t)) -> hashes of the accumulation buffer and the frame.
// usage: node animate_glb.js model.glb [width] [height] [frames] [interval] [depth]   -> prints one JSON line
const crypto = require('crypto');
const fs = require('fs');
const { WebGPURenderer, WorldBridge, LiveLoop } = require('./index.js');

(async () => {
  const [file, w = '96', h = '64', frames = '6', interval = '2', depth = '5'] = process.argv.slice(2);
  const width = parseInt(w, 10), height = parseInt(h, 10);
  const renderer = new WebGPURenderer(0);
  await renderer.init();
  renderer.buildPipeline(parseInt(depth, 10), 1);
  const bridge = new WorldBridge();
  await bridge.initWasm();
  if (process.env.RT_NODE_GPU_BLAS) bridge.setBlasBuilder(renderer);
  if (process.env.RT_NODE_DEVICE_UPDATE) bridge.setDeviceUpdater(renderer);   // update(t) inside the renderer (rt_world_update)
  await bridge.loadScene('viewer', undefined, new Uint8Array(fs.readFileSync(file)));
  await renderer.loadTexturesFromWorld(bridge);
  renderer.updateScreenSize(width, height);
  const loop = new LiveLoop(renderer, bridge, width, height, parseInt(interval, 10));
  for (let f = 0; f < parseInt(frames, 10); f++) loop.renderFrame();
  await renderer.device.queue.onSubmittedWorkDone();
  const acc = renderer.readAccum();
  const frame = await renderer.captureFrame();
  const sha = (buf) => crypto.createHash('sha256').update(Buffer.from(buf)).digest('hex');
  console.log(JSON.stringify({ frames: parseInt(frames, 10), animations: bridge.getAnimationList(), frameCount: loop.frameCount,
    accum_sha256: sha(acc.buffer), rgba_sha256: sha(frame.data), deviceResident: !!bridge.deviceResident }));
  renderer.destroy();
})().catch((e) => { console.error(e); process.exit(1); });
