'use strict';
// Headless driver reproducing the reference's call sequence (src/main.ts:51-116,119-181):
// load scene -> upload buffers -> resolution -> N x { compute(frame); present() } -> captureFrame.
// usage: node render_cornell.js [scene] [width] [height] [frames] [depth]   -> prints one JSON line
const crypto = require('crypto');
const { WebGPURenderer, WorldBridge } = require('./index.js');

(async () => {
  const [scene = 'cornell', w = '96', h = '96', frames = '3', depth = '4'] = process.argv.slice(2);
  const width = parseInt(w, 10), height = parseInt(h, 10);
  const bridge = new WorldBridge();
  await bridge.initWasm();
  await bridge.loadScene(scene);
  const renderer = new WebGPURenderer(0);
  await renderer.init();
  renderer.buildPipeline(parseInt(depth, 10), 1);
  if (process.env.RT_NODE_GPU_BLAS) {  // rebuild the scene's BLASes with the GPU builder (same arrays, byte for byte)
    bridge.setBlasBuilder(renderer);
    bridge.update(0);
  }
  await renderer.loadTexturesFromWorld(bridge);
  renderer.updateCombinedGeometry(bridge.vertices, bridge.normals, bridge.uvs);
  renderer.updateCombinedBVH(bridge.tlas, bridge.blas);
  renderer.updateBuffer('topology', bridge.mesh_topology);
  renderer.updateBuffer('instance', bridge.instances);
  renderer.updateBuffer('lights', bridge.lights);
  renderer.updateBuffer('draw_commands', bridge.draw_commands);
  renderer.updateScreenSize(width, height);
  bridge.updateCamera(width, height);
  renderer.updateSceneUniforms(bridge.cameraData, 0, bridge.lightCount);
  renderer.recreateBindGroup();
  renderer.resetAccumulation();
  for (let f = 1; f <= parseInt(frames, 10); f++) { renderer.compute(f); renderer.present(); }
  if (process.env.RT_NODE_BATCH) {  // extra frames through the batched entry point
    const n = parseInt(process.env.RT_NODE_BATCH, 10), first = parseInt(frames, 10) + 1;
    renderer.computeBatch(Array.from({ length: n }, (_, i) => first + i));
    renderer.present();
  }
  await renderer.device.queue.onSubmittedWorkDone();
  const frame = await renderer.captureFrame();
  const acc = renderer.readAccum();
  const sha = (buf) => crypto.createHash('sha256').update(Buffer.from(buf)).digest('hex');
  console.log(JSON.stringify({ scene, width, height, frames: parseInt(frames, 10),
    rgba_sha256: sha(frame.data), accum_sha256: sha(acc.buffer), counters: renderer.getCounters() }));
  renderer.destroy();
})().catch((e) => { console.error(e); process.exit(1); });
