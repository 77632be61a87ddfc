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
  } else {
    throw new Error('unknown command ' + cmd);
  }
}
main().catch((e) => { console.error(e.stack || String(e)); process.exit(1); });
