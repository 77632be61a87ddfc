'use strict';
// Driven by tests/test_js_host.py: node host_check.js <command> <args...>; prints JSON on stdout.
const fs = require('fs');
const path = require('path');
const g = require(path.join(__dirname, '..', '..', 'gaussian-splatting-wgpu_amd', 'js'));

function camJSON(cam, w, h) {
  return { view: Array.from(cam.viewMatrix), proj: Array.from(cam.getProjMatrix()), pos: Array.from(cam.getPosition()), uniforms: Array.from(cam.packUniforms(w, h)) };
}

async function main() {
  const cmd = process.argv[2];
  if (cmd === 'camera') {
    const out = {};
    const c = g.Camera.default();
    out.default = camJSON(c, 800, 800);
    const ic = new g.InteractiveCamera(g.Camera.default(), { width: 640, height: 480 });
    ic.key('w'); ic.key('d'); ic.key('q'); ic.key('j'); ic.key('i'); ic.key('u');
    out.moved = camJSON(ic.getCamera(), 640, 480);
    out.dirtyAfter = ic.isDirty();
    const raw = JSON.parse(process.argv[3]);
    out.fromJSON = camJSON(g.cameraFromJSON(raw, 800, 800), 800, 800);
    console.log(JSON.stringify(out));
  } else if (cmd === 'ply') {
    const buf = await g.loadFileAsArrayBuffer(process.argv[3]);
    const pg = new g.PackedGaussians(buf);
    fs.writeFileSync(process.argv[4], Buffer.from(pg.gaussiansBuffer));
    console.log(JSON.stringify({ n: pg.numGaussians, degree: pg.sphericalHarmonicsDegree, size: pg.gaussianArrayLayout.size, nSh: pg.nShCoeffs }));
  } else if (cmd === 'plynative') {
    const pg = g.PackedGaussians.fromFile(process.argv[3]);
    fs.writeFileSync(process.argv[4], Buffer.from(pg.gaussiansBuffer));
    console.log(JSON.stringify({ n: pg.numGaussians, degree: pg.sphericalHarmonicsDegree, size: pg.gaussianArrayLayout.size }));
  } else if (cmd === 'render') {
    // render <records.bin> <n> <W> <H> <tile> <uniforms.bin> <out.rgba>
    const rec = fs.readFileSync(process.argv[3]);
    const n = parseInt(process.argv[4], 10), W = parseInt(process.argv[5], 10), H = parseInt(process.argv[6], 10), ts = parseInt(process.argv[7], 10);
    const ub = fs.readFileSync(process.argv[8]);
    const u = new Float32Array(ub.buffer.slice(ub.byteOffset, ub.byteOffset + 160));
    const pg = g.PackedGaussians.fromRecords(rec.buffer.slice(rec.byteOffset, rec.byteOffset + rec.byteLength), n);
    // a camera whose packUniforms returns exactly the given block
    const cam = { packUniforms: (w, h, out) => { out.set(u); return out; } };
    const ic = { dirty: true, isDirty() { return this.dirty; }, getCamera() { this.dirty = false; return cam; } };
    let frames = 0;
    let last = null;
    const canvas = { width: W, height: H, onFrame: (rgba) => { frames++; last = rgba; } };
    const r = new g.Renderer(canvas, ic, { ordinal: 0, flags: g.loadNative().FLAG_EXACT_BLEND }, pg, ts);
    // the constructor armed animate() like requestAnimationFrame does; wait for the first frame
    while (frames === 0) await new Promise((res) => setImmediate(res));
    fs.writeFileSync(process.argv[9], Buffer.from(last));
    const st = r.stats();
    const keys = new Uint32Array(r.readBuffer(g.BUF.KEYS));
    await r.destroy();
    console.log(JSON.stringify({ frames, numIntersections: r.numIntersections, stats: st, nkeys: keys.length, key0: keys.length ? keys[0] : 0 }));
  } else if (cmd === 'pipeline') {
    // pipeline <records.bin> <n> <W> <H> <tile> <uniforms.bin (K x 160 B)> <K>: K cameras through the pipelined animate() (canvas.pipeline = 3,
    // pinned sinks, no await between the calls) must deliver, in order, the frames a synchronous render of each camera reads back
    const rec = fs.readFileSync(process.argv[3]);
    const n = parseInt(process.argv[4], 10), W = parseInt(process.argv[5], 10), H = parseInt(process.argv[6], 10), ts = parseInt(process.argv[7], 10);
    const ub = fs.readFileSync(process.argv[8]);
    const K = parseInt(process.argv[9], 10);
    const us = [];
    for (let k = 0; k < K; ++k) us.push(new Float32Array(ub.buffer.slice(ub.byteOffset + 160 * k, ub.byteOffset + 160 * (k + 1))));
    const pg = g.PackedGaussians.fromRecords(rec.buffer.slice(rec.byteOffset, rec.byteOffset + rec.byteLength), n);
    const idle = { isDirty() { return false; }, getCamera() { return null; } };
    const a = new g.Renderer({ width: W, height: H, manual: true }, idle, { ordinal: 0 }, pg, ts);
    const want = [];
    for (let k = 0; k < K; ++k) { a.renderUniforms(us[k]); want.push(a.readPixels()); }
    let step = 0;
    const cam = { packUniforms: (w, h, o) => { o.set(us[step]); return o; } };
    const ic = { isDirty() { return true; }, getCamera() { return cam; } };
    const got = [];
    const canvas = { width: W, height: H, manual: true, pipeline: 3, onFrame: (rgba) => { got.push(Uint8Array.from(rgba)); } };
    const b = new g.Renderer(canvas, ic, { ordinal: 0, shareWith: a }, pg, ts);
    for (let rep = 0; rep < 2; ++rep) for (let k = 0; k < K; ++k) b.renderUniforms(us[k]); // capacities
    const ps = [];
    for (let k = 0; k < K; ++k) { step = k; ps.push(b.animate()); }
    await Promise.all(ps);
    let same = got.length === K;
    for (let k = 0; same && k < K; ++k) {
      same = got[k].length === want[k].length;
      for (let i = 0; same && i < want[k].length; ++i) same = got[k][i] === want[k][i];
    }
    const frames = b.numFrames;
    await b.destroy(); await a.destroy();
    console.log(JSON.stringify({ same, delivered: got.length, frames }));
  } else if (cmd === 'bench') {
    // bench <scene.ply> <W> <H> <tile> <orbit.bin (64 x 160 B)> <frames> : frames/s of Renderer.animate() with a dirty camera every frame,
    //   (a) as the reference drives it: await animate() per frame, no frame sink;  (b) the same with an onFrame sink (8 MB read-back + a
    //   fresh Uint8Array per frame);  (c) canvas.pipeline = 3: pinned sinks, frame k+1 enqueued before frame k's pixels have arrived
    const W = parseInt(process.argv[4], 10), H = parseInt(process.argv[5], 10), ts = parseInt(process.argv[6], 10);
    const ob = fs.readFileSync(process.argv[7]);
    const frames = parseInt(process.argv[8], 10);
    const orbit = [];
    for (let k = 0; k < 64; ++k) orbit.push(new Float32Array(ob.buffer.slice(ob.byteOffset + 160 * k, ob.byteOffset + 160 * (k + 1))));
    const out = {};
    let owner = null;
    for (const mode of ['await_no_sink', 'await_with_sink', 'pipeline3_pinned_sinks']) {
      let step = 0, delivered = 0, sum = 0;
      const cam = { packUniforms: (w, h, o) => { o.set(orbit[step % 64]); return o; } };
      const ic = { isDirty() { return true; }, getCamera() { return cam; } };
      const canvas = { width: W, height: H, manual: true };
      if (mode !== 'await_no_sink') canvas.onFrame = (rgba) => { delivered++; sum += rgba[(delivered * 7919) % rgba.length]; };
      if (mode === 'pipeline3_pinned_sinks') canvas.pipeline = 3;
      const dev = owner ? { ordinal: 0, shareWith: owner } : { ordinal: 0 };
      const r = new g.Renderer(canvas, ic, dev, { numGaussians: 0, plyPath: process.argv[3] }, ts);
      if (!owner) owner = r;
      for (let rep = 0; rep < 2; ++rep) // capacities of every member of the ring for the whole orbit (outside the timed loop)
        for (let k = 0; k < 64; ++k) r.renderUniforms(orbit[k]);
      const run = async (n) => {
        const inflight = [];
        for (let i = 0; i < n; ++i) {
          step = i;
          const p = r.animate();
          if (r.pipeline > 1) { inflight.push(p); if (inflight.length >= r.pipeline) await inflight.shift(); } else await p;
        }
        await Promise.all(inflight);
      };
      await run(20);
      const t0 = process.hrtime.bigint();
      await run(frames);
      const dt = Number(process.hrtime.bigint() - t0) / 1e9;
      out[mode] = { frames, fps: frames / dt, ms_per_frame: dt / frames * 1e3, delivered, checksum: sum };
      if (r !== owner) await r.destroy();
    }
    out.numGaussians = owner.numGaussians;
    await owner.destroy();
    console.log(JSON.stringify(out));
  } else if (cmd === 'shared') {
    // shared <records.bin> <n> <W> <H> <tile> <uniforms.bin>: a second renderer borrows the first one's splats
    const rec = fs.readFileSync(process.argv[3]);
    const n = parseInt(process.argv[4], 10), W = parseInt(process.argv[5], 10), H = parseInt(process.argv[6], 10), ts = parseInt(process.argv[7], 10);
    const ub = fs.readFileSync(process.argv[8]);
    const u = new Float32Array(ub.buffer.slice(ub.byteOffset, ub.byteOffset + 160));
    const pg = g.PackedGaussians.fromRecords(rec.buffer.slice(rec.byteOffset, rec.byteOffset + rec.byteLength), n);
    const ic = { isDirty() { return false; }, getCamera() { return null; } };
    const a = new g.Renderer({ width: W, height: H, manual: true }, ic, { ordinal: 0 }, pg, ts);
    const b = new g.Renderer({ width: W, height: H, manual: true }, ic, { ordinal: 0, shareWith: a }, pg, ts);
    a.renderUniforms(u); b.renderUniforms(u);
    const pa = a.readPixels(), pb = b.readPixels();
    let same = pa.length === pb.length && pa.length === W * H * 4;
    for (let i = 0; same && i < pa.length; ++i) same = pa[i] === pb[i];
    await b.destroy(); await a.destroy();
    console.log(JSON.stringify({ same, bytes: pa.length }));
  } else if (cmd === 'ppm') {
    // ppm <out.ppm> <W> <H>: a known rgba pattern through the presentation sink
    const W = parseInt(process.argv[4], 10), H = parseInt(process.argv[5], 10);
    const rgba = new Uint8Array(W * H * 4);
    for (let i = 0; i < W * H; ++i) { rgba[4 * i] = i & 255; rgba[4 * i + 1] = (i >> 3) & 255; rgba[4 * i + 2] = (7 * i) & 255; rgba[4 * i + 3] = 255; }
    g.writePPM(process.argv[3], rgba, W, H);
    console.log(JSON.stringify({ ok: true }));
  } else if (cmd === 'camfile') {
    // camfile <cameras.json>: the reference's CameraFileParser list (camera.ts:344-400) without the DOM
    const list = g.loadCameraFile(process.argv[3], { width: 800, height: 800 });
    console.log(JSON.stringify(list.map((e) => ({ name: e.name, cam: camJSON(e.camera, 800, 800) }))));
  } else if (cmd === 'busy') {
    // busy <records.bin> <n> <W> <H> <uniforms.bin>: the native handle refuses calls while a renderAsync frame is in flight,
    // and destroy() during the frame is deferred to its completion
    const rec = fs.readFileSync(process.argv[3]);
    const n = parseInt(process.argv[4], 10), W = parseInt(process.argv[5], 10), H = parseInt(process.argv[6], 10);
    const ub = fs.readFileSync(process.argv[7]);
    const u = new Float32Array(ub.buffer.slice(ub.byteOffset, ub.byteOffset + 160));
    const nat = g.loadNative();
    const h = nat.create({ width: W, height: H, tileSize: 16, device: 0 });
    nat.uploadSplats(h, rec.buffer.slice(rec.byteOffset, rec.byteOffset + rec.byteLength), n);
    let badN = false;
    try { nat.uploadSplats(h, rec.buffer, -1); } catch (e) { badN = true; }
    const p = nat.renderAsync(h, u);
    let refused = 0;
    try { nat.renderSync(h, u); } catch (e) { refused += /in flight/.test(String(e)) ? 1 : 0; }
    try { nat.stats(h); } catch (e) { refused += /in flight/.test(String(e)) ? 1 : 0; }
    try { nat.renderAsync(h, u); } catch (e) { refused += /in flight/.test(String(e)) ? 1 : 0; }
    nat.destroy(h); // deferred: the worker still owns the context
    await p;
    let gone = false;
    try { nat.stats(h); } catch (e) { gone = /destroyed/.test(String(e)); }
    console.log(JSON.stringify({ refused, gone, badN }));
  } else {
    throw new Error('unknown command ' + cmd);
  }
}
main().catch((e) => { console.error(e.stack || String(e)); process.exit(1); });
