"""Camera mirror (camera.ts): algebraic pins, since wgpu-matrix itself is not available offline."""
import numpy as np

from gsplat.camera import Camera, focal2fov, get_projection_matrix, mat4_inverse, mat4_multiply
from gsplat import synth


def _M(m):
    return np.asarray(m, dtype=np.float64).reshape(4, 4).T


def test_projection_matrix_entries():
    """camera.ts:16-39 with fov from focal2fov: P[0][0] = 2f/W, P[1][1] = 2f/H, w_clip = z_view."""
    W, H, f = 1920, 1080, 1920.0
    P = _M(get_projection_matrix(0.2, 100.0, focal2fov(f, W), focal2fov(f, H)))
    np.testing.assert_allclose(P[0, 0], 2 * f / W, rtol=1e-6)
    np.testing.assert_allclose(P[1, 1], 2 * f / H, rtol=1e-6)
    np.testing.assert_allclose(P[3], [0, 0, 1, 0], atol=1e-7)
    np.testing.assert_allclose(P[2, 2], 100.0 / 99.8, rtol=1e-6)
    np.testing.assert_allclose(P[2, 3], -100.0 * 0.2 / 99.8, rtol=1e-6)


def test_position_is_minus_Rt_t():
    cam = synth.orbit_camera(13, 640, 360)
    V = _M(cam.viewMatrix)
    np.testing.assert_allclose(cam.getPosition(), -V[:3, :3].T @ V[:3, 3], atol=1e-5)
    np.testing.assert_allclose(_M(mat4_multiply(cam.viewMatrix, mat4_inverse(cam.viewMatrix))), np.eye(4), atol=1e-5)
    np.testing.assert_allclose(_M(cam.getProjMatrix()), _M(cam.perspective) @ V, rtol=1e-5, atol=1e-6)


def test_uniform_block_layout():
    """renderer.ts:15-24,362-392: 160 bytes, tanHalfFov = 0.5*canvas/focal."""
    cam = synth.orbit_camera(0, 1920, 1080)
    u = cam.uniforms(1920, 1080)
    assert u.nbytes == 160
    np.testing.assert_array_equal(u[0:16], cam.viewMatrix)
    assert u[35] == np.float32(0.5) and u[36] == np.float32(0.5 * 1080 / 1920) and u[37] == 1920 and u[39] == 1


def test_translate_rotate_roundtrip():
    cam = Camera.default()
    v0 = cam.viewMatrix.copy()
    cam.translate(0.1, -0.2, 0.3)
    cam.translate(-0.1, 0.2, -0.3)
    np.testing.assert_allclose(cam.viewMatrix, v0, atol=2e-6)
    cam.rotate(0.0, 0.0, 0.25)
    cam.rotate(0.0, 0.0, -0.25)
    np.testing.assert_allclose(cam.viewMatrix, v0, atol=2e-6)


def test_from_json_is_world_to_camera():
    """cameraFromJSON (camera.ts:323-340): rotation rows become columns; view = R^T (x - position)."""
    th = 0.3
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    pos = np.array([1.0, 2.0, 3.0])
    cam = Camera.from_json({"rotation": R.tolist(), "position": pos.tolist()})
    V = _M(cam.viewMatrix)
    x = np.array([0.5, -0.25, 4.0])
    np.testing.assert_allclose(V[:3, :3] @ x + V[:3, 3], R.T @ (x - pos), atol=1e-5)
