'use strict';
// Node-host restatement of the reference's Renderer (renderer.ts:35-594) on the MI355X C ABI.
// Same constructor and methods: new Renderer(canvas, interactiveCamera, device, gaussians, tileSize),
// animate(): Promise<void>, destroy(): Promise<void>.  `canvas` is any {width, height} with an
// optional onFrame(Uint8Array rgba, width, height) sink (the blit target of render.wgsl);
// `device` is the HIP device ordinal (or {ordinal, flags}) where the reference takes a GPUDevice.
const path = require('path');

let native = null;
function loadNative() {
  if (!native) {
    const file = path.join(__dirname, '..', 'lib', 'gsplat_napi.node');
    try {
      native = require(file);
    } catch (e) {
      throw new Error(`gsplat: native addon not built (${file}): run python gaussian-splatting-wgpu_amd/csrc/build.py; there is no fallback renderer. ${e.message}`);
    }
  }
  return native;
}

class Renderer {
  constructor(canvas, interactiveCamera, device, gaussians, tileSize) {
    this.tileSize = tileSize;
    this.canvas = canvas;
    this.interactiveCamera = interactiveCamera;
    this.device = device;
    this.numFrames = 0;
    this.numIntersections = 0;
    this.numGaussians = gaussians.numGaussians;
    this.destroyCallback = null;
    this.destroyed = false;
    this.lastDraw = Date.now();
    this.frameTimes = null;
    const n = loadNative();
    const ordinal = typeof device === 'number' ? device : ((device && device.ordinal) || 0);
    const flags = (device && device.flags) || 0;
    // device.shareWith: another Renderer on the same device whose resident splats this one renders (gs_share_splats) -
    // several renderers driven round-robin keep several frames in flight (the reference has one: every stage is awaited)
    const owner = (device && device.shareWith) || null;
    this.owner = owner; // keeps the owner alive
    if (!canvas || !(canvas.width > 0) || !(canvas.height > 0)) throw new Error('WebGPU context not found!'); // renderer.ts:108-111
    this.handle = n.create({ width: canvas.width, height: canvas.height, tileSize, device: ordinal, flags });
    if (owner) n.shareSplats(this.handle, owner.handle);
    else n.uploadSplats(this.handle, gaussians.gaussiansBuffer, this.numGaussians); // renderer.ts:130-137
    this.uniforms = new Float32Array(40);
    this.autoSchedule = !(canvas.manual === true);
    if (this.autoSchedule) setImmediate(() => this.animate()); // requestAnimationFrame(() => this.animate()), renderer.ts:323
  }

  // resolves when the renderer has been torn down on the next animate() tick (renderer.ts:90-94)
  destroy() {
    return new Promise((resolve) => {
      this.destroyCallback = resolve;
      if (!this.autoSchedule) this.animate();
    });
  }

  destroyImpl() {
    if (this.destroyCallback === null) throw new Error('destroyImpl called without destroyCallback set!');
    if (!this.destroyed) {
      loadNative().destroy(this.handle);
      this.destroyed = true;
    }
    this.destroyCallback();
  }

  async animate() {
    if (this.destroyCallback !== null) { this.destroyImpl(); return; }
    if (this.destroyed) return;
    const rearm = () => { if (this.autoSchedule) setImmediate(() => this.animate()); };
    if (!this.interactiveCamera.isDirty()) { rearm(); return; }
    const camera = this.interactiveCamera.getCamera();
    camera.packUniforms(this.canvas.width, this.canvas.height, this.uniforms); // renderer.ts:362-392
    const n = loadNative();
    await n.renderAsync(this.handle, this.uniforms); // the whole frame, renderer.ts:394-574
    if (this.destroyed) return;
    const st = n.stats(this.handle);
    this.numIntersections = st.numIntersections;
    this.frameTimes = st.stageUs;
    this.numFrames++;
    if (typeof this.canvas.onFrame === 'function') {
      this.canvas.onFrame(new Uint8Array(n.readRgba8(this.handle)), this.canvas.width, this.canvas.height);
    }
    rearm();
  }

  // synchronous helpers for tools and tests
  renderUniforms(uniforms, debug) { loadNative().renderSync(this.handle, uniforms, !!debug); this.numFrames++; }
  readPixels() { return new Uint8Array(loadNative().readRgba8(this.handle)); }
  readBuffer(which) { return loadNative().readBuffer(this.handle, which); }
  stats() { return loadNative().stats(this.handle); }
}

module.exports = { Renderer, loadNative };
