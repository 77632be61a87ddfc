'use strict';
// The subset of wgpu-matrix 2.9.0 that the reference's camera.ts uses (camera.ts:1,25,38,43,90,
// 146-147,154,159-161,166-170,316-318,332), restated: column-major Float32Array(16) `Mat4`,
// 12-float padded `Mat3`, `multiply(a,b) = a*b`, translate/rotate post-multiply.  wgpu-matrix is
// not vendored (no network): parity with it is unpinned and checked algebraically in tests.

const mat4 = {
  create(...v) {
    const m = new Float32Array(16);
    for (let i = 0; i < v.length && i < 16; ++i) m[i] = v[i];
    return m;
  },
  identity() { return mat4.create(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1); },
  clone(m) { return new Float32Array(m); },
  multiply(a, b, dst) {
    dst = dst || new Float32Array(16);
    const o = new Array(16);
    for (let c = 0; c < 4; ++c) {
      for (let r = 0; r < 4; ++r) {
        o[c * 4 + r] = a[r] * b[c * 4] + a[4 + r] * b[c * 4 + 1] + a[8 + r] * b[c * 4 + 2] + a[12 + r] * b[c * 4 + 3];
      }
    }
    for (let i = 0; i < 16; ++i) dst[i] = o[i];
    return dst;
  },
  transpose(m, dst) {
    dst = dst || new Float32Array(16);
    const o = new Array(16);
    for (let c = 0; c < 4; ++c) for (let r = 0; r < 4; ++r) o[c * 4 + r] = m[r * 4 + c];
    for (let i = 0; i < 16; ++i) dst[i] = o[i];
    return dst;
  },
  inverse(m, dst) {
    dst = dst || new Float32Array(16);
    // cofactor expansion in doubles, stored as f32
    const m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3], m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    const m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11], m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
    const t0 = m22 * m33, t1 = m32 * m23, t2 = m12 * m33, t3 = m32 * m13, t4 = m12 * m23, t5 = m22 * m13;
    const t6 = m02 * m33, t7 = m32 * m03, t8 = m02 * m23, t9 = m22 * m03, t10 = m02 * m13, t11 = m12 * m03;
    const t12 = m20 * m31, t13 = m30 * m21, t14 = m10 * m31, t15 = m30 * m11, t16 = m10 * m21, t17 = m20 * m11;
    const t18 = m00 * m31, t19 = m30 * m01, t20 = m00 * m21, t21 = m20 * m01, t22 = m00 * m11, t23 = m10 * m01;
    const c0 = (t0 * m11 + t3 * m21 + t4 * m31) - (t1 * m11 + t2 * m21 + t5 * m31);
    const c1 = (t1 * m01 + t6 * m21 + t9 * m31) - (t0 * m01 + t7 * m21 + t8 * m31);
    const c2 = (t2 * m01 + t7 * m11 + t10 * m31) - (t3 * m01 + t6 * m11 + t11 * m31);
    const c3 = (t5 * m01 + t8 * m11 + t11 * m21) - (t4 * m01 + t9 * m11 + t10 * m21);
    const d = 1.0 / (m00 * c0 + m10 * c1 + m20 * c2 + m30 * c3);
    const o = [
      d * c0, d * c1, d * c2, d * c3,
      d * ((t1 * m10 + t2 * m20 + t5 * m30) - (t0 * m10 + t3 * m20 + t4 * m30)),
      d * ((t0 * m00 + t7 * m20 + t8 * m30) - (t1 * m00 + t6 * m20 + t9 * m30)),
      d * ((t3 * m00 + t6 * m10 + t11 * m30) - (t2 * m00 + t7 * m10 + t10 * m30)),
      d * ((t4 * m00 + t9 * m10 + t10 * m20) - (t5 * m00 + t8 * m10 + t11 * m20)),
      d * ((t12 * m13 + t15 * m23 + t16 * m33) - (t13 * m13 + t14 * m23 + t17 * m33)),
      d * ((t13 * m03 + t18 * m23 + t21 * m33) - (t12 * m03 + t19 * m23 + t20 * m33)),
      d * ((t14 * m03 + t19 * m13 + t22 * m33) - (t15 * m03 + t18 * m13 + t23 * m33)),
      d * ((t17 * m03 + t20 * m13 + t23 * m23) - (t16 * m03 + t21 * m13 + t22 * m23)),
      d * ((t14 * m22 + t17 * m32 + t13 * m12) - (t16 * m32 + t12 * m12 + t15 * m22)),
      d * ((t20 * m32 + t12 * m02 + t19 * m22) - (t18 * m22 + t21 * m32 + t13 * m02)),
      d * ((t18 * m12 + t23 * m32 + t15 * m02) - (t22 * m32 + t14 * m02 + t19 * m12)),
      d * ((t22 * m22 + t16 * m02 + t21 * m12) - (t20 * m12 + t23 * m22 + t17 * m02)),
    ];
    for (let i = 0; i < 16; ++i) dst[i] = o[i];
    return dst;
  },
  translation(v) { const m = mat4.identity(); m[12] = v[0]; m[13] = v[1]; m[14] = v[2]; return m; },
  translate(m, v, dst) { return mat4.multiply(m, mat4.translation(v), dst); },
  rotationX(a) { const c = Math.cos(a), s = Math.sin(a); return mat4.create(1, 0, 0, 0, 0, c, s, 0, 0, -s, c, 0, 0, 0, 0, 1); },
  rotationY(a) { const c = Math.cos(a), s = Math.sin(a); return mat4.create(c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0, 0, 0, 0, 1); },
  rotationZ(a) { const c = Math.cos(a), s = Math.sin(a); return mat4.create(c, s, 0, 0, -s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1); },
  rotateX(m, a, dst) { return mat4.multiply(m, mat4.rotationX(a), dst); },
  rotateY(m, a, dst) { return mat4.multiply(m, mat4.rotationY(a), dst); },
  rotateZ(m, a, dst) { return mat4.multiply(m, mat4.rotationZ(a), dst); },
  getTranslation(m) { return new Float32Array([m[12], m[13], m[14]]); },
  fromMat3(m3) {
    return mat4.create(m3[0], m3[1], m3[2], 0, m3[4], m3[5], m3[6], 0, m3[8], m3[9], m3[10], 0, 0, 0, 0, 1);
  },
};

const mat3 = {
  // 9 values fill the three columns of a 12-float padded matrix
  create(...v) {
    const m = new Float32Array(12);
    if (v.length >= 9) { m[0] = v[0]; m[1] = v[1]; m[2] = v[2]; m[4] = v[3]; m[5] = v[4]; m[6] = v[5]; m[8] = v[6]; m[9] = v[7]; m[10] = v[8]; }
    return m;
  },
};

const vec3 = {
  mulScalar(v, k) { return new Float32Array([v[0] * k, v[1] * k, v[2] * k]); },
  dot(a, b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; },
};

module.exports = { mat4, mat3, vec3 };
