"""Seeded synthetic scenes and camera orbits for tests and bench.py.

The scene is produced directly in the reference's packed layout: the 320-byte AoS record that
``PackedGaussians`` builds (ply.ts:190-198): pos@0, log_scale@16, rot@32 (r,x,y,z), opacity@48,
sh@64 = 16 x vec3 (stride 16 B).  numpy's counter-based Philox generator keeps the bits identical
on every machine; chunks are seeded independently so a scene of N splats is a prefix of a larger one.
"""
import numpy as np

from .camera import Camera, get_projection_matrix, focal2fov

SPLAT_FLOATS = 80  # 320 B
CHUNK = 1 << 16
BASE_SEED = 0x5EED0001


def bicycle_like(n, seed=BASE_SEED, out=None):
    """"bicycle-like" distribution: 70 % foreground N(0,1.5^2) clipped to radius 4, 30 % background
    shell radius 4..25; anisotropic log-scales growing with distance; random rotations, opacities, SH."""
    n = int(n)
    if out is None:
        out = np.zeros((n, SPLAT_FLOATS), dtype=np.float32)
    for c0 in range(0, n, CHUNK):
        m = min(CHUNK, n - c0)
        rng = np.random.Generator(np.random.Philox(key=[seed, c0 // CHUNK]))
        full = CHUNK  # always draw a full chunk so that prefixes agree
        sel = rng.random(full, dtype=np.float32)
        fg = rng.standard_normal((full, 3), dtype=np.float32) * np.float32(1.5)
        r = np.linalg.norm(fg, axis=1)
        fg *= np.minimum(1.0, 4.0 / np.maximum(r, 1e-12)).astype(np.float32)[:, None]
        d = rng.standard_normal((full, 3), dtype=np.float32)
        d /= np.maximum(np.linalg.norm(d, axis=1), 1e-12)[:, None]
        rad = (4.0 + 21.0 * rng.random(full, dtype=np.float32)).astype(np.float32)
        bg = d * rad[:, None]
        pos = np.where((sel < 0.7)[:, None], fg, bg).astype(np.float32)
        dist = np.maximum(1.0, np.linalg.norm(pos, axis=1)).astype(np.float32)
        ls = (np.log(np.float32(0.004) * dist)[:, None] +
              np.float32(0.8) * rng.standard_normal((full, 3), dtype=np.float32)).astype(np.float32)
        rot = rng.standard_normal((full, 4), dtype=np.float32)
        op = (np.float32(1.0) + np.float32(2.5) * rng.standard_normal(full, dtype=np.float32)).astype(np.float32)
        dc = (np.float32(0.5) + rng.standard_normal((full, 3), dtype=np.float32)).astype(np.float32)
        rest = (np.float32(0.08) * rng.standard_normal((full, 15, 3), dtype=np.float32)).astype(np.float32)
        o = out[c0:c0 + m]
        o[:, 0:3] = pos[:m]
        o[:, 4:7] = ls[:m]
        o[:, 8:12] = rot[:m]
        o[:, 12] = op[:m]
        o[:, 16:19] = dc[:m]
        sh = o[:, 20:80].reshape(m, 15, 4)
        sh[:, :, 0:3] = rest[:m]
    return out


def look_at_view(eye, target=(0.0, 0.0, 0.0), world_up=(0.0, 1.0, 0.0)):
    """World->camera matrix (column-major float32[16]) in the 3DGS convention the reference's
    shaders assume: +x right, +y down, +z forward (depth = view.z, process_gaussians.wgsl:118-120)."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    down_w = -np.asarray(world_up, dtype=np.float64)
    r = np.cross(down_w, f)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f])  # rows
    t = -R @ eye
    m = np.zeros(16, dtype=np.float64)
    for c in range(3):
        for rr in range(3):
            m[c * 4 + rr] = R[rr, c]
    m[12:15] = t
    m[15] = 1.0
    return m.astype(np.float32)


def orbit_camera(step, W, H, steps=64, radius=4.5, height=1.0, znear=0.2, zfar=100.0):
    """Camera on the benchmark orbit: focalX = focalY = W so tan_fovx = 0.5 like Camera.default
    (camera.ts:80-88,118-119)."""
    th = 2.0 * np.pi * (step % steps) / steps
    eye = (radius * np.cos(th), height, radius * np.sin(th))
    focal = float(W)
    fovx, fovy = focal2fov(focal, W), focal2fov(focal, H)
    return Camera(H, W, look_at_view(eye), get_projection_matrix(znear, zfar, fovx, fovy), focal, focal, 1.0)


def bicycle_like_torch(n, seed=BASE_SEED, device="cuda"):
    """Same distribution as bicycle_like, generated straight into device memory with torch's
    generator (bench.py: avoids a minute of host RNG and a 2 GB PCIe copy at 6.1 M splats).
    Deterministic for a given seed on a given torch build; NOT bit-identical to the numpy scene."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    n = int(n)
    out = torch.zeros((n, SPLAT_FLOATS), dtype=torch.float32, device=device)
    sel = torch.rand(n, generator=g, device=device)
    fg = torch.randn((n, 3), generator=g, device=device) * 1.5
    r = fg.norm(dim=1).clamp_min(1e-12)
    fg = fg * torch.clamp(4.0 / r, max=1.0)[:, None]
    d = torch.randn((n, 3), generator=g, device=device)
    d = d / d.norm(dim=1).clamp_min(1e-12)[:, None]
    rad = 4.0 + 21.0 * torch.rand(n, generator=g, device=device)
    pos = torch.where((sel < 0.7)[:, None], fg, d * rad[:, None])
    dist = pos.norm(dim=1).clamp_min(1.0)
    out[:, 0:3] = pos
    out[:, 4:7] = torch.log(0.004 * dist)[:, None] + 0.8 * torch.randn((n, 3), generator=g, device=device)
    out[:, 8:12] = torch.randn((n, 4), generator=g, device=device)
    out[:, 12] = 1.0 + 2.5 * torch.randn(n, generator=g, device=device)
    out[:, 16:19] = 0.5 + torch.randn((n, 3), generator=g, device=device)
    out[:, 20:80].view(n, 15, 4)[:, :, 0:3] = 0.08 * torch.randn((n, 15, 3), generator=g, device=device)
    return out
