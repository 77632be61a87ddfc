'use strict';
// Node-host restatement of the reference's camera.ts: Camera (camera.ts:52-190), InteractiveCamera
// (:193-308, DOM listeners replaced by an explicit input API), cameraFromJSON (:323-340) and
// getProjectionMatrix / focal2fov (:16-39, :310-312).  Same class names, fields and methods.
const fs = require('fs');
const { mat4, mat3, vec3 } = require('./mat4');

function getProjectionMatrix(znear, zfar, fovX, fovY) {
  const tanHalfFovY = Math.tan(fovY / 2), tanHalfFovX = Math.tan(fovX / 2);
  const top = tanHalfFovY * znear, bottom = -top, right = tanHalfFovX * znear, left = -right;
  const P = mat4.create();
  const zSign = 1.0;
  P[0] = (2.0 * znear) / (right - left);
  P[5] = (2.0 * znear) / (top - bottom);
  P[8] = (right + left) / (right - left);
  P[9] = (top + bottom) / (top - bottom);
  P[10] = zSign * zfar / (zfar - znear);
  P[11] = -(zfar * znear) / (zfar - znear);
  P[14] = zSign;
  P[15] = 0.0;
  return mat4.transpose(P);
}

function focal2fov(focal, pixels) { return 2 * Math.atan(pixels / (2 * focal)); }

class Camera {
  constructor(height, width, viewMatrix, perspective, focalX, focalY, scaleModifier) {
    this.height = height;
    this.width = width;
    this.viewMatrix = viewMatrix;
    this.perspective = perspective;
    this.focalX = focalX;
    this.focalY = focalY;
    this.scaleModifier = scaleModifier;
  }

  static default(_canvas) {
    const canvasW = 800, canvasH = 800, fovFactor = 1;
    const fovX = focal2fov(canvasW, canvasW) / fovFactor, fovY = focal2fov(canvasH, canvasH) / fovFactor;
    const projectionMatrix = getProjectionMatrix(0.2, 10, fovX, fovY);
    const viewMatrix = mat4.create(
      0.582345724105835, -0.3235852122306824, 0.7372694611549377, 0,
      0.23868794739246368, 0.9381394982337952, 0.22253619134426117, 0,
      -0.7680802941322327, 0.04477229341864586, 0.6242981553077698, 0,
      0.13517332077026367, -1.1848870515823364, 3.3873789310455322, 1);
    return new Camera(canvasW, canvasH, viewMatrix, projectionMatrix, canvasW, canvasH, 1 * fovFactor);
  }

  setScale(scale) { this.scaleModifier = scale; }
  setFocalX(focalX) { this.focalX = focalX; }
  setFocalY(focalY) { this.focalY = focalY; }

  getPosition() { return mat4.getTranslation(mat4.inverse(this.viewMatrix)); }
  getProjMatrix() { return mat4.multiply(this.perspective, this.viewMatrix); }

  translate(x, y, z) {
    const viewInv = mat4.inverse(this.viewMatrix);
    mat4.translate(viewInv, [x, y, z], viewInv);
    mat4.inverse(viewInv, this.viewMatrix);
  }

  rotate(x, y, z) {
    const viewInv = mat4.inverse(this.viewMatrix);
    mat4.rotateX(viewInv, y, viewInv);
    mat4.rotateY(viewInv, x, viewInv);
    mat4.rotateZ(viewInv, z, viewInv);
    mat4.inverse(viewInv, this.viewMatrix);
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

  key(k) {
    const keyMap = {
      w: () => { this.dTY -= 0.1; }, s: () => { this.dTY += 0.1; }, a: () => { this.dTX -= 0.1; }, d: () => { this.dTX += 0.1; },
      q: () => { this.dTZ += 0.1; }, e: () => { this.dTZ -= 0.1; }, j: () => { this.dRX += 0.1; }, l: () => { this.dRX -= 0.1; },
      i: () => { this.dRY += 0.1; }, k: () => { this.dRY -= 0.1; }, u: () => { this.dRZ += 0.1; }, o: () => { this.dRZ -= 0.1; },
    };
    if (!keyMap[k]) return false;
    keyMap[k]();
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

function worldToCamFromRT(R, t) {
  const camToWorld = mat4.fromMat3(R);
  const minusT = vec3.mulScalar(t, -1);
  mat4.translate(camToWorld, minusT, camToWorld);
  return camToWorld;
}

function cameraFromJSON(rawCamera, _canvasW, _canvasH) {
  const canvW = 800, canvH = 800;
  const fovX = focal2fov(canvW, canvW), fovY = focal2fov(canvH, canvH);
  const projectionMatrix = getProjectionMatrix(0.2, 100, fovX, fovY);
  const flat = [].concat(...rawCamera.rotation);
  const R = mat3.create(...flat);
  const viewMatrix = worldToCamFromRT(R, rawCamera.position);
  return new Camera(canvH, canvW, viewMatrix, projectionMatrix, canvW, canvH, 1);
}

// CameraFileParser (camera.ts:344-400) without the <ul> UI: returns [{name, camera}].
function loadCameraFile(path, canvas) {
  const list = JSON.parse(fs.readFileSync(path, 'utf8'));
  return list.map((c) => ({ name: c.img_name, camera: cameraFromJSON(c, canvas ? canvas.width : 800, canvas ? canvas.height : 800) }));
}

module.exports = { Camera, InteractiveCamera, cameraFromJSON, loadCameraFile, getProjectionMatrix, focal2fov };
