// TypeScript view of index.js: the method surface of the reference classes
// (src/renderer/WebGPURenderer.ts:7-138, src/world-bridge.ts:4-216).
export class WebGPURenderer {
  constructor(device?: number);
  readonly device: { queue: { onSubmittedWorkDone(): Promise<void> } };
  init(): Promise<void>;
  buildPipeline(depth: number, spp: number): void;
  updateScreenSize(width: number, height: number): void;
  resetAccumulation(): void;
  loadTexturesFromWorld(bridge: WorldBridge): Promise<void>;
  updateBuffer(type: "topology" | "instance" | "lights" | "draw_commands", data: Uint32Array | Float32Array): boolean;
  updateCombinedGeometry(v: Float32Array, n: Float32Array, uv: Float32Array): boolean;
  updateCombinedBVH(tlas: Float32Array, blas: Float32Array): boolean;
  updateSceneUniforms(cameraData: Float32Array, frameCount: number, lightCount: number): void;
  recreateBindGroup(): void;
  compute(frameCount: number): void;
  computeBatch(frameCounts: ArrayLike<number>): void;
  present(): void;
  captureFrame(): Promise<{ data: ArrayBufferLike; width: number; height: number }>;
  readAccum(): Float32Array;
  getCounters(): Record<string, number>;
  destroy(): void;
}
export class WorldBridge {
  hasNewData: boolean;
  hasNewGeometry: boolean;
  initWasm(): Promise<void>;
  loadScene(sceneName: string, objSource?: string, glbData?: Uint8Array): Promise<void>;
  update(time: number): void;
  updateCamera(width: number, height: number): void;
  readonly vertices: Float32Array;
  readonly normals: Float32Array;
  readonly uvs: Float32Array;
  readonly mesh_topology: Uint32Array;
  readonly tlas: Float32Array;
  readonly blas: Float32Array;
  readonly instances: Float32Array;
  readonly lights: Uint32Array;
  readonly lightCount: number;
  readonly draw_commands: Uint32Array;
  readonly cameraData: Float32Array;
  readonly textureCount: number;
  readonly hasWorld: boolean;
  getTextureRGBA(index: number): Uint8Array | undefined;
  /** build the BLASes of update(t) on the GPU with this renderer (same tree as the CPU builder); null restores the CPU builder */
  setBlasBuilder(renderer: WebGPURenderer | null): void;
  /** the whole per-frame half of update(t) on the GPU (rt_world_update); the host arrays are then not refreshed */
  setDeviceUpdater(renderer: WebGPURenderer | null): void;
  deviceResident: boolean;
  deviceWarning: string;
  getAnimationList(): string[];
  loadAnimation(data: Uint8Array): number;
  setAnimation(index: number): void;
  /** why a GLB passed to loadScene was ignored ('' when it loaded) */
  loadWarning: string;
  /** encoded image bytes (PNG / JPEG), as world-bridge.ts:101-106 hands them out */
  getTexture(index: number): Uint8Array | undefined;
}

/** src/main.ts:133-163 — re-upload what the bridge marks as new, reset the accumulation; true when something was uploaded */
export function syncWorld(renderer: WebGPURenderer, bridge: WorldBridge, width: number, height: number): boolean;
/** `renderFrame` of src/main.ts:119-181 */
export class LiveLoop {
  constructor(renderer: WebGPURenderer, bridge: WorldBridge, width: number, height: number, updateInterval?: number);
  frameCount: number;
  totalFrameCount: number;
  renderFrame(): void;
}
