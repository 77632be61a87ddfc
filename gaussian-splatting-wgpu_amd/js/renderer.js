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
    else if (gaussians.plyPath) this.numGaussians = n.uploadPly(this.handle, gaussians.plyPath); // streaming loader: no packed buffer on the host
    else n.uploadSplats(this.handle, gaussians.gaussiansBuffer, this.numGaussians); // renderer.ts:130-137
    this.uniforms = new Float32Array(40);
    // canvas.pipeline = K >= 2 (extension; the reference awaits every stage of every frame, renderer.ts:394-587): animate() still
    // resolves once per frame and hands onFrame that frame's pixels, but it enqueues frame k+1 before frame k's pixels have
    // arrived -- up to K frames are on the device at once -- and the pixels land in K page-locked sinks that are reused
    // (a view handed to onFrame is valid until K more frames have been enqueued) instead of a fresh 8 MB ArrayBuffer per frame.
    this.pipeline = Math.max(1, Math.min(4, (canvas.pipeline | 0) || 1));
    this.sinks = null;
    this.pending = []; // promises of the frames in flight, oldest first
    this.slot = 0;
    if (this.pipeline > 1) {
      const bytes = canvas.width * canvas.height * 4;
      this.sinks = [];
      for (let k = 0; k < this.pipeline; ++k) this.sinks.push(new Uint8Array(n.hostAlloc(bytes)));
    }
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
      loadNative().destroy(this.handle); // (frames still in flight: the native side defers the teardown to their completion)
      this.destroyed = true;
    }
    this.destroyCallback();
  }

  // One frame of the pipelined mode: enqueue now, resolve when THIS frame's pixels are in its sink.
  animatePipelined() {
    const n = loadNative();
    const k = this.slot;
    this.slot = (k + 1) % this.pipeline;
    const sink = this.sinks[k];
    const u = new Float32Array(this.uniforms); // this frame's block: `uniforms` is repacked by the next animate() before the enqueue below may run
    const wait = this.pending.length >= this.pipeline ? this.pending.shift() : Promise.resolve();
    // the sink of slot k is free once the frame that used it K frames ago has been delivered
    const p = wait.then(() => {
      if (this.destroyed) return undefined;
      return n.renderToSink(this.handle, u, sink).then(() => {
        this.numFrames++;
        if (typeof this.canvas.onFrame === 'function') this.canvas.onFrame(sink, this.canvas.width, this.canvas.height);
      });
    });
    this.pending.push(p.catch(() => {}));
    return p;
  }

  async animate() {
    if (this.destroyCallback !== null) {
      if (this.pending.length) { await Promise.all(this.pending); this.pending = []; }
      this.destroyImpl();
      return;
    }
    if (this.destroyed) return;
    const rearm = () => { if (this.autoSchedule) setImmediate(() => this.animate()); };
    if (!this.interactiveCamera.isDirty()) { rearm(); return; }
    const camera = this.interactiveCamera.getCamera();
    camera.packUniforms(this.canvas.width, this.canvas.height, this.uniforms); // renderer.ts:362-392
    if (this.pipeline > 1) {
      const p = this.animatePipelined(); // (the uniforms are copied by the native call before it returns)
      rearm(); // the next frame may be enqueued at once
      await p;
      return;
    }
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
