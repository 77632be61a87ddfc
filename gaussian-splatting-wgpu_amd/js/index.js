'use strict';
// gsplat (Node host): the reference's class surface -- Renderer, Camera, InteractiveCamera,
// PackedGaussians, loadFileAsArrayBuffer -- backed by the MI355X-native C ABI through N-API.
const { Renderer, loadNative } = require('./renderer');
const { Camera, InteractiveCamera, cameraFromJSON, loadCameraFile, getProjectionMatrix, focal2fov } = require('./camera');
const { PackedGaussians, loadFileAsArrayBuffer } = require('./ply');
const { mat4, mat3, vec3 } = require('./mat4');

// Writes an rgba8 frame as a binary PPM (presentation sink for hosts without a canvas).
function writePPM(file, rgba, width, height) {
  const fs = require('fs');
  const header = Buffer.from(`P6\n${width} ${height}\n255\n`, 'ascii');
  const rgb = Buffer.alloc(width * height * 3);
  for (let i = 0, j = 0; i < width * height; ++i) { rgb[j++] = rgba[4 * i]; rgb[j++] = rgba[4 * i + 1]; rgb[j++] = rgba[4 * i + 2]; }
  fs.writeFileSync(file, Buffer.concat([header, rgb]));
}

module.exports = {
  Renderer, Camera, InteractiveCamera, PackedGaussians, loadFileAsArrayBuffer, cameraFromJSON, loadCameraFile,
  getProjectionMatrix, focal2fov, mat4, mat3, vec3, writePPM, loadNative,
  BUF: { TILE_COUNTS: 0, TILE_OFFSETS: 1, GAUSSIAN_DATA: 2, KEYS_UNSORTED: 3, VALUES_UNSORTED: 4, KEYS: 5, VALUES: 6, RANGES: 7, RGBA8: 8, RGB_F32: 9 },
};
