"""Python mirror of the reference's ``Camera`` (camera.ts:16-39,52-190,310-340) and of the uniform
block ``Renderer.animate`` derives from it (renderer.ts:362-392).

Matrices are column-major float32[16] like wgpu-matrix's ``Mat4`` (a Float32Array): every helper
computes in float64 from float32 inputs and rounds the result to float32, which is what storing a
JS number into a Float32Array does.  wgpu-matrix 2.9.0 itself is not in the container, so the
library conventions below are restated from its documented behaviour: parity unpinned.
"""
import math

import numpy as np


def _f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32)


def mat4_multiply(a, b):
    """mat4.multiply(a, b) = a * b (column-major)."""
    A = np.asarray(a, dtype=np.float64).reshape(4, 4).T
    B = np.asarray(b, dtype=np.float64).reshape(4, 4).T
    return _f32((A @ B).T.reshape(16))


def mat4_inverse(m):
    M = np.asarray(m, dtype=np.float64).reshape(4, 4).T
    return _f32(np.linalg.inv(M).T.reshape(16))


def mat4_transpose(m):
    return _f32(np.asarray(m, dtype=np.float64).reshape(4, 4).T.reshape(16))


def mat4_translation(v):
    m = np.eye(4)
    m[0:3, 3] = v
    return _f32(m.T.reshape(16))


def mat4_rotation(axis, angle):
    c, s = math.cos(angle), math.sin(angle)
    m = np.eye(4)
    if axis == 0:
        m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    elif axis == 1:
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    else:
        m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return _f32(m.T.reshape(16))


def focal2fov(focal, pixels):
    """camera.ts:310-312"""
    return 2.0 * math.atan(pixels / (2.0 * focal))


def get_projection_matrix(znear, zfar, fov_x, fov_y):
    """camera.ts:16-39, written directly in column-major form (the reference fills a row-major matrix and
    transposes it): x' = 2n/(r-l) x, y' = 2n/(t-b) y, z' = f/(f-n) z - fn/(f-n), w' = z; symmetric frustum."""
    half_w, half_h = math.tan(0.5 * fov_x) * znear, math.tan(0.5 * fov_y) * znear
    depth = zfar - znear
    m = np.zeros(16, dtype=np.float32)
    m[0] = (2.0 * znear) / (half_w + half_w)
    m[5] = (2.0 * znear) / (half_h + half_h)
    m[10] = zfar / depth
    m[14] = -(zfar * znear) / depth
    m[11] = 1.0
    return m


class Camera:
    """camera.ts:52-190.  Constructor argument order is the reference's (height, width, ...)."""

    def __init__(self, height, width, view_matrix, perspective, focal_x, focal_y, scale_modifier):
        self.height = height
        self.width = width
        self.viewMatrix = np.asarray(view_matrix, dtype=np.float32).copy()
        self.perspective = np.asarray(perspective, dtype=np.float32).copy()
        self.focalX = focal_x
        self.focalY = focal_y
        self.scaleModifier = scale_modifier

    @staticmethod
    def default():
        """camera.ts:79-122 (800x800, f = 800, the literal view matrix at :90-107)."""
        fov = focal2fov(800, 800)
        view = np.array([0.582345724105835, -0.3235852122306824, 0.7372694611549377, 0,
                         0.23868794739246368, 0.9381394982337952, 0.22253619134426117, 0,
                         -0.7680802941322327, 0.04477229341864586, 0.6242981553077698, 0,
                         0.13517332077026367, -1.1848870515823364, 3.3873789310455322, 1], dtype=np.float32)
        return Camera(800, 800, view, get_projection_matrix(0.2, 10, fov, fov), 800, 800, 1)

    @staticmethod
    def from_json(raw):
        """cameraFromJSON (camera.ts:323-340): 3DGS cameras.json entry -> Camera (800x800, f=800)."""
        fov = focal2fov(800, 800)
        R = np.asarray(raw["rotation"], dtype=np.float64)  # rows; mat3.create(...flat()) makes them columns
        m = np.eye(4)
        m[0:3, 0:3] = R.T
        cam = _f32(m.T.reshape(16))
        view = mat4_multiply(cam, mat4_translation(-np.asarray(raw["position"], dtype=np.float64)))
        return Camera(800, 800, view, get_projection_matrix(0.2, 100, fov, fov), 800, 800, 1)

    def getPosition(self):
        """camera.ts:145-148"""
        return mat4_inverse(self.viewMatrix)[12:15].copy()

    def getProjMatrix(self):
        """camera.ts:150-155"""
        return mat4_multiply(self.perspective, self.viewMatrix)

    def translate(self, x, y, z):
        """camera.ts:158-162"""
        inv = mat4_multiply(mat4_inverse(self.viewMatrix), mat4_translation([x, y, z]))
        self.viewMatrix = mat4_inverse(inv)

    def rotate(self, x, y, z):
        """camera.ts:165-171 (note the reference applies y about X and x about Y)"""
        inv = mat4_inverse(self.viewMatrix)
        inv = mat4_multiply(inv, mat4_rotation(0, y))
        inv = mat4_multiply(inv, mat4_rotation(1, x))
        inv = mat4_multiply(inv, mat4_rotation(2, z))
        self.viewMatrix = mat4_inverse(inv)

    def uniforms(self, canvas_width, canvas_height):
        """The 160-byte uniform block (renderer.ts:15-24,362-392) as float32[40]."""
        u = np.zeros(40, dtype=np.float32)
        u[0:16] = self.viewMatrix
        u[16:32] = self.getProjMatrix()
        u[32:35] = self.getPosition()
        u[35] = 0.5 * canvas_width / self.focalX
        u[36] = 0.5 * canvas_height / self.focalY
        u[37] = self.focalX
        u[38] = self.focalY
        u[39] = self.scaleModifier
        return u
