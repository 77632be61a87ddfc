'use strict';
// Node-host restatement of the reference's camera.ts: Camera (camera.ts:52-190), InteractiveCamera
// (:193-308, DOM listeners replaced by an explicit input API), cameraFromJSON (:323-340) and
// getProjectionMatrix / focal2fov (:16-39, :310-312).  Same class names, fields and methods.
const fs = require('fs');
const { mat4, mat3, vec3 } = require('./mat4');

// Perspective matrix of camera.ts:16-39, written directly in the column-major form the reference obtains by
// filling a row-major matrix and transposing it: x' = 2n/(r-l) x, y' = 2n/(t-b) y, z' = f/(f-n) z - fn/(f-n), w' = z
// (symmetric frustum, so the (r+l) and (t+b) terms vanish).
function getProjectionMatrix(znear, zfar, fovX, fovY) {
  const halfW = Math.tan(0.5 * fovX) * znear;
  const halfH = Math.tan(0.5 * fovY) * znear;
  const depth = zfar - znear;
  const m = new Float32Array(16);
  m[0] = (2.0 * znear) / (halfW + halfW);
  m[5] = (2.0 * znear) / (halfH + halfH);
  m[10] = zfar / depth;
  m[14] = -(zfar * znear) / depth;
  m[11] = 1.0;
  m[15] = 0.0;
  return m;
}

function focal2fov(focal, pixels) { return 2 * Math.atan(pixels / (2 * focal)); }

// world -> camera matrix of the reference's start-up view (the literal at camera.ts:90-107), column-major
const DEFAULT_VIEW = [
  0.582345724105835, -0.3235852122306824, 0.7372694611549377, 0,
  0.23868794739246368, 0.9381394982337952, 0.22253619134426117, 0,
  -0.7680802941322327, 0.04477229341864586, 0.6242981553077698, 0,
  0.13517332077026367, -1.1848870515823364, 3.3873789310455322, 1,
];

// Same fields and methods as the reference's Camera (camera.ts:52-190); the pose edits go through the
// inverse view matrix (camera-to-world) exactly as there.
class Camera {
  constructor(height, width, viewMatrix, perspective, focalX, focalY, scaleModifier) {
    Object.assign(this, { height, width, viewMatrix, perspective, focalX, focalY, scaleModifier });
  }

  // 800x800 canvas, focal 800, near 0.2, far 10 (camera.ts:79-122)
  static default(_canvas) {
    const size = 800;
    const fov = focal2fov(size, size);
    return new Camera(size, size, mat4.create(...DEFAULT_VIEW), getProjectionMatrix(0.2, 10, fov, fov), size, size, 1);
  }

  setScale(scale) { this.scaleModifier = scale; }
  setFocalX(focalX) { this.focalX = focalX; }
  setFocalY(focalY) { this.focalY = focalY; }

  // camera position in world space = translation of the inverse view matrix (camera.ts:145-148)
  getPosition() { return mat4.getTranslation(mat4.inverse(this.viewMatrix)); }
  // perspective * view (camera.ts:150-155)
  getProjMatrix() { return mat4.multiply(this.perspective, this.viewMatrix); }

  // post-multiplies the camera-to-world matrix by `edit` and inverts back (camera.ts:158-171)
  _editPose(edit) {
    const camToWorld = mat4.inverse(this.viewMatrix);
    edit(camToWorld);
    mat4.inverse(camToWorld, this.viewMatrix);
  }

  translate(x, y, z) { this._editPose((m) => mat4.translate(m, [x, y, z], m)); }

  // note the reference's argument order: `y` turns about X, `x` about Y (camera.ts:167-168)
  rotate(x, y, z) {
    this._editPose((m) => {
      mat4.rotateX(m, y, m);
      mat4.rotateY(m, x, m);
      mat4.rotateZ(m, z, m);
    });
  }

  // The 160-byte uniform block Renderer.animate packs (renderer.ts:15-24,362-392).
  packUniforms(canvasWidth, canvasHeight, out) {
    const u = out || new Float32Array(40);
    u.set(this.viewMatrix, 0);
    u.set(this.getProjMatrix(), 16);
    u.set(this.getPosition(), 32);
    u[35] = 0.5 * canvasWidth / this.focalX;
    u[36] = 0.5 * canvasHeight / this.focalY;
    u[37] = this.focalX;
    u[38] = this.focalY;
    u[39] = this.scaleModifier;
    return u;
  }
}

// The reference wires DOM mouse/keyboard listeners (camera.ts:220-279); a Node host feeds the same
// deltas through key()/drag()/wheel().
class InteractiveCamera {
  constructor(camera, canvas) {
    this.camera = camera;
    this.canvas = canvas;
    this.dRX = this.dRY = this.dRZ = this.dTX = this.dTY = this.dTZ = 0;
    this.dirty = true;
  }

  static default(canvas) { return new InteractiveCamera(Camera.default(canvas), canvas); }

  // keyboard deltas of camera.ts:251-278 as a table: key -> [field, step]
  key(k) {
    const step = InteractiveCamera.KEY_STEPS[k];
    if (!step) return false;
    this[step[0]] += step[1];
    this.dirty = true;
    return true;
  }

  drag(movementX, movementY) {
    this.dRX = (movementX * 2 * Math.PI) / this.canvas.width;
    this.dRY = (-movementY * 2 * Math.PI) / this.canvas.height;
    this.dirty = true;
  }

  wheel(deltaY) { this.dTZ = deltaY * 0.1; this.dirty = true; }

  setNewCamera(newCamera) { this.camera = newCamera; this.dirty = true; }
  isDirty() { return this.dirty; }

  getCamera() {
    if (this.isDirty()) {
      this.camera.translate(this.dTX, this.dTY, this.dTZ);
      this.camera.rotate(this.dRX, this.dRY, this.dRZ);
      this.dTX = this.dTY = this.dTZ = this.dRX = this.dRY = this.dRZ = 0;
      this.dirty = false;
    }
    return this.camera;
  }
}

InteractiveCamera.KEY_STEPS = {
  w: ['dTY', -0.1], s: ['dTY', 0.1], a: ['dTX', -0.1], d: ['dTX', 0.1], q: ['dTZ', 0.1], e: ['dTZ', -0.1],
  j: ['dRX', 0.1], l: ['dRX', -0.1], i: ['dRY', 0.1], k: ['dRY', -0.1], u: ['dRZ', 0.1], o: ['dRZ', -0.1],
};

// 3DGS cameras.json entry -> Camera (camera.ts:314-340): the rows of `rotation` become the columns of the 3x3 block
// (mat3.create(...rotation.flat())), followed by a translation by -position: view = [R^T | 0] * T(-position).
function cameraFromJSON(rawCamera, _canvasW, _canvasH) {
  const size = 800;
  const fov = focal2fov(size, size);
  const flat = [].concat(...rawCamera.rotation);
  const view = mat4.fromMat3(mat3.create(...flat));
  mat4.translate(view, vec3.mulScalar(rawCamera.position, -1), view);
  return new Camera(size, size, view, getProjectionMatrix(0.2, 100, fov, fov), size, size, 1);
}

// CameraFileParser (camera.ts:344-400) without the <ul> UI: returns [{name, camera}].
function loadCameraFile(path, canvas) {
  const list = JSON.parse(fs.readFileSync(path, 'utf8'));
  return list.map((c) => ({ name: c.img_name, camera: cameraFromJSON(c, canvas ? canvas.width : 800, canvas ? canvas.height : 800) }));
}

module.exports = { Camera, InteractiveCamera, cameraFromJSON, loadCameraFile, getProjectionMatrix, focal2fov };
