'use strict';
// Node-host restatement of the reference's ply.ts: PackedGaussians (ply.ts:32-228) and
// loadFileAsArrayBuffer (ply.ts:3-30, FileReader replaced by fs).  Header rules are the
// reference's: `element vertex N` (:77-81), properties in file order (:82-88), data starts right
// after `end_header\n` (:94), ONLY `float` and `uchar` properties consume bytes (:113-119), SH read
// order f_dc_{0..2} then f_rest_{rgb*K + i} (:179-187), degree from the f_rest count (:168-176).
// The record layout is the packed 320-byte struct of ply.ts:190-198 (packing.ts:142-174,233-245);
// the hot path hard-codes 16 SH coefficients (process_gaussians.wgsl:6), so lower degrees are
// padded with zeros to the same stride (SURVEY A.1 canonical decision).
const fs = require('fs');

const RECORD_FLOATS = 80; // 320 bytes

function loadFileAsArrayBuffer(path) {
  return new Promise((resolve, reject) => {
    fs.readFile(path, (err, buf) => {
      if (err) { reject(err); return; }
      resolve(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength));
    });
  });
}

class PackedGaussians {
  static decodeHeader(plyArrayBuffer) {
    const bytes = new Uint8Array(plyArrayBuffer);
    let headerText = '';
    let headerOffset = 0;
    while (true) {
      if (headerOffset >= bytes.length) throw new Error('PLY header has no end_header');
      const end = Math.min(headerOffset + 50, bytes.length);
      headerText += Buffer.from(plyArrayBuffer, headerOffset, end - headerOffset).toString('latin1');
      headerOffset += 50;
      if (headerText.includes('end_header')) break;
    }
    let vertexCount = 0;
    const propertyTypes = [];
    for (const raw of headerText.split('\n')) {
      const line = raw.trim();
      if (line.startsWith('element vertex')) {
        const m = line.match(/\d+/);
        if (m) vertexCount = parseInt(m[0], 10);
      } else if (line.startsWith('property')) {
        const m = line.match(/(\w+)\s+(\w+)\s+(\w+)/);
        if (m) {
          const i = propertyTypes.findIndex((p) => p[0] === m[3]);
          if (i >= 0) propertyTypes[i][1] = m[2]; else propertyTypes.push([m[3], m[2]]);
        }
      } else if (line === 'end_header') {
        break;
      }
    }
    const vertexByteOffset = headerText.indexOf('end_header') + 'end_header'.length + 1;
    return [vertexCount, propertyTypes, new DataView(plyArrayBuffer, vertexByteOffset)];
  }

  get nShCoeffs() {
    const d = this.sphericalHarmonicsDegree;
    if (d === 0) return 1;
    if (d === 1) return 4;
    if (d === 2) return 9;
    if (d === 3) return 16;
    throw new Error(`Unsupported SH degree: ${d}`);
  }

  constructor(arrayBuffer) {
    const [vertexCount, propertyTypes, vertexData] = PackedGaussians.decodeHeader(arrayBuffer);
    this.numGaussians = vertexCount;
    let nRestCoeffs = 0;
    for (const [name] of propertyTypes) if (name.startsWith('f_rest_')) nRestCoeffs += 1;
    const nCoeffsPerColor = nRestCoeffs / 3;
    this.sphericalHarmonicsDegree = Math.sqrt(nCoeffsPerColor + 1) - 1;
    const nSh = this.nShCoeffs; // throws on an unsupported degree

    // byte offset of every property inside a vertex (only float / uchar advance, ply.ts:113-119)
    const offsets = {};
    let stride = 0;
    for (const [name, type] of propertyTypes) {
      offsets[name] = [stride, type];
      if (type === 'float') stride += 4; else if (type === 'uchar') stride += 1;
    }
    const need = ['x', 'y', 'z', 'scale_0', 'scale_1', 'scale_2', 'rot_0', 'rot_1', 'rot_2', 'rot_3', 'opacity', 'f_dc_0', 'f_dc_1', 'f_dc_2'];
    const shOrder = ['f_dc_0', 'f_dc_1', 'f_dc_2'];
    for (let i = 0; i < nCoeffsPerColor; ++i) for (let rgb = 0; rgb < 3; ++rgb) shOrder.push(`f_rest_${rgb * nCoeffsPerColor + i}`);
    for (const n of need.concat(shOrder)) if (!offsets[n]) throw new Error(`PLY is missing property ${n}`);
    if (vertexData.byteLength < vertexCount * stride) throw new Error('PLY vertex data is truncated');

    const read = (base, name) => {
      const [off, type] = offsets[name];
      if (type === 'float') return vertexData.getFloat32(base + off, true);
      if (type === 'uchar') return vertexData.getUint8(base + off) / 255.0;
      return undefined;
    };

    this.gaussianLayout = { size: RECORD_FLOATS * 4 };
    this.gaussianArrayLayout = { size: vertexCount * RECORD_FLOATS * 4 };
    this.gaussiansBuffer = new ArrayBuffer(this.gaussianArrayLayout.size);
    const out = new Float32Array(this.gaussiansBuffer);
    const fields = [['x', 0], ['y', 1], ['z', 2], ['scale_0', 4], ['scale_1', 5], ['scale_2', 6], ['rot_0', 8], ['rot_1', 9],
      ['rot_2', 10], ['rot_3', 11], ['opacity', 12]];
    for (let i = 0; i < vertexCount; ++i) {
      const base = i * stride, o = i * RECORD_FLOATS;
      for (const [name, slot] of fields) out[o + slot] = read(base, name);
      for (let k = 0; k < nSh; ++k) for (let c = 0; c < 3; ++c) out[o + 16 + 4 * k + c] = read(base, shOrder[3 * k + c]);
    }
  }

  // Native loader (C ABI gs_ply_load through N-API): same result as `new PackedGaussians(arrayBuffer)`,
  // without the per-vertex JS object churn the reference warns about (index.html:16).
  static fromFile(path) {
    const { loadNative } = require('./renderer');
    const r = loadNative().loadPly(path);
    const pg = PackedGaussians.fromRecords(r.records, r.n);
    pg.sphericalHarmonicsDegree = r.degree;
    return pg;
  }

  // Builds a PackedGaussians straight from packed 320-byte records (synthetic scenes, tests).
  static fromRecords(arrayBuffer, numGaussians) {
    const pg = Object.create(PackedGaussians.prototype);
    pg.numGaussians = numGaussians;
    pg.sphericalHarmonicsDegree = 3;
    pg.gaussianLayout = { size: RECORD_FLOATS * 4 };
    pg.gaussianArrayLayout = { size: numGaussians * RECORD_FLOATS * 4 };
    pg.gaussiansBuffer = arrayBuffer;
    return pg;
  }
}

module.exports = { PackedGaussians, loadFileAsArrayBuffer };
