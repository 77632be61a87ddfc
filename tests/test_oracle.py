"""CPU tests of the oracle (no GPU): pins it against the reference's own known-answer material,
against an independently written numpy restatement, against analytic single-splat cases and against
the committed golden vectors."""
import glob
import hashlib
import os

import numpy as np
import pytest

from conftest import scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---- the reference's own KATs ---------------------------------------------------------------------
def test_scan_matches_reference_serial_definition(oracle):
    """exclusive_scan.ts:105-112 (commented-out serialExclusiveScan): out[0]=0;
    out[i]=in[i-1]+out[i-1]; returns out[n-1]+in[n-1]."""
    rng = np.random.default_rng(7)
    for n in (1, 2, 511, 512, 513, 262144 + 5):
        a = rng.integers(0, 100, size=n, dtype=np.uint32)
        out = np.zeros(n, dtype=np.uint64)
        for i in range(1, min(n, 2000)):
            out[i] = a[i - 1] + out[i - 1]
        offs, total = oracle.scan(a)
        np.testing.assert_array_equal(offs[: min(n, 2000)], out[: min(n, 2000)].astype(np.uint32))
        assert total == int(a.sum())
        np.testing.assert_array_equal(offs, np.concatenate([[0], np.cumsum(a[:-1], dtype=np.uint64)]).astype(np.uint32))


def test_sort_matches_reference_testsort_vector(oracle):
    """radix_sort/utils.ts:55-81: n = 8192 keys n-1-i must come out as 0..n-1."""
    n = 8192
    keys = np.arange(n - 1, -1, -1, dtype=np.uint32)
    k, v = oracle.sort(keys, np.arange(n, dtype=np.uint32))
    np.testing.assert_array_equal(k, np.arange(n, dtype=np.uint32))
    np.testing.assert_array_equal(v, np.arange(n - 1, -1, -1, dtype=np.uint32))


def test_sort_is_stable(oracle):
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 50, size=200000, dtype=np.uint32) * np.uint32(1000) + rng.integers(0, 3, size=200000, dtype=np.uint32)
    vals = np.arange(keys.size, dtype=np.uint32)
    k, v = oracle.sort(keys, vals)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(k, keys[order])
    np.testing.assert_array_equal(v, vals[order])


def test_camera_default_matrix_is_the_reference_literal():
    """camera.ts:90-107."""
    from gsplat.camera import Camera
    v = Camera.default().viewMatrix
    assert v[0] == np.float32(0.582345724105835) and v[14] == np.float32(3.3873789310455322) and v[15] == 1


# ---- canonical exp ----------------------------------------------------------------------------------
def test_canonical_exp_accuracy_and_agreement(oracle):
    from oracle import np_oracle
    x = np.concatenate([np.linspace(-104.5, 89.0, 40001), np.linspace(-6, 0.5, 20001), [0.0, -0.0, np.inf, -np.inf]]).astype(np.float32)
    e = np_oracle.expf(x)
    ref = np.exp(x.astype(np.float64))
    ok = np.isfinite(ref) & (ref > 1.2e-38) & (ref < 3.4e38)
    rel = np.abs(e[ok].astype(np.float64) - ref[ok]) / ref[ok]
    assert rel.max() < 1.5 * 2.0 ** -23  # <= ~1.5 ulp; WGSL allows 3 + 2|x| ulp
    c = np.array([oracle.expf(v) for v in x[::61]], dtype=np.float32)
    np.testing.assert_array_equal(c.view(np.uint32), e[::61].view(np.uint32))  # C and numpy agree bit for bit
    assert oracle.expf(100.0) == np.inf and oracle.expf(-200.0) == 0.0 and np.isnan(oracle.expf(np.nan))


# ---- C oracle vs the independent numpy restatement -------------------------------------------------------
@pytest.mark.parametrize("n,W,H,ts,step", [(3000, 128, 128, 16, 5), (1500, 100, 60, 8, 17), (4000, 160, 96, 32, 40)])
def test_c_oracle_equals_numpy_restatement(oracle, n, W, H, ts, step):
    from gsplat import synth
    from oracle import np_oracle as NP
    s = scene(n)
    u = synth.orbit_camera(step, W, H).uniforms(W, H)
    r = oracle.render(s, u, W, H, ts)
    pre = NP.preprocess(s, u, W, H, ts)
    gd = r["gdata"]
    np.testing.assert_array_equal(gd[:, 0:2], pre["uv"].view(np.uint32))
    np.testing.assert_array_equal(gd[:, 4:7], pre["conic"].view(np.uint32))
    np.testing.assert_array_equal(gd[:, 7], pre["depth"].view(np.uint32))
    np.testing.assert_array_equal(gd[:, 8:11], pre["color"].view(np.uint32))
    np.testing.assert_array_equal(gd[:, 11], pre["opacity"].view(np.uint32))
    np.testing.assert_array_equal(gd[:, 12:16], pre["rect"])
    np.testing.assert_array_equal(r["tile_counts"], pre["count"])
    k, v = NP.keys_values(pre, W, ts)
    np.testing.assert_array_equal(k, r["keys"])
    np.testing.assert_array_equal(v, r["values"])
    sk, sv = NP.sort_kv(k, v)
    np.testing.assert_array_equal(sk, r["sorted_keys"])
    np.testing.assert_array_equal(sv, r["sorted_values"])
    ntx, nty = oracle.num_tiles(W, H, ts)
    rg = NP.ranges(sk, ntx * nty)
    np.testing.assert_array_equal(rg, r["ranges"])
    img = NP.blend(pre, sv, rg, W, H, ts)
    np.testing.assert_array_equal(img.view(np.uint32), r["rgbf"].view(np.uint32))
    np.testing.assert_array_equal(NP.to_rgba8(img), r["rgba8"])


# ---- analytic single-splat cases ---------------------------------------------------------------------------
def _one_splat(pos, log_scale, rot=(1, 0, 0, 0), opacity=10.0, dc=(1.0, 0.5, 0.25)):
    s = np.zeros((1, 80), dtype=np.float32)
    s[0, 0:3] = pos
    s[0, 4:7] = log_scale
    s[0, 8:12] = rot
    s[0, 12] = opacity
    s[0, 16:19] = dc
    return s


def _front_camera(W, H, focal):
    """Camera at the origin looking down +z, identity rotation."""
    from gsplat.camera import Camera, focal2fov, get_projection_matrix
    view = np.eye(4, dtype=np.float32).T.reshape(16)
    return Camera(H, W, view, get_projection_matrix(0.2, 100.0, focal2fov(focal, W), focal2fov(focal, H)), focal, focal, 1.0)


def test_isotropic_splat_matches_closed_form(oracle):
    """Sigma = s^2 I, J = diag(f/z): cov2d = (f s / z)^2 + 0.3 on the diagonal, conic = 1/that,
    uv = 0.5 (on the axis), colour = 0.5 + C0*dc, opacity = sigmoid(o)."""
    W = H = 128
    f, z, sc = 128.0, 4.0, 0.25
    cam = _front_camera(W, H, f)
    s = _one_splat((0, 0, z), np.log([sc] * 3))
    gd, cnt = oracle.preprocess(s, cam.uniforms(W, H), W, H)
    g = gd[0].view(np.float32)
    var = (f * sc / z) ** 2 + 0.3
    assert abs(g[0] - 0.5) < 1e-6 and abs(g[1] - 0.5) < 1e-6
    np.testing.assert_allclose([g[4], g[6]], [1 / var, 1 / var], rtol=1e-5)
    assert abs(g[5]) < 1e-6 and abs(g[7] - z) < 1e-6
    np.testing.assert_allclose(g[8:11], 0.5 + 0.28209479177387814 * np.array([1.0, 0.5, 0.25]), rtol=1e-6)
    assert abs(g[11] - 1 / (1 + np.exp(-10.0))) < 1e-6
    radius = np.ceil(3 * np.sqrt(var))
    lo, hi = int(64 - radius) // 16, int(64 + radius) // 16 + 1
    np.testing.assert_array_equal(gd[0, 12:16], [lo, lo, hi, hi])
    assert cnt[0] == (hi - lo) ** 2
    # centre pixel: alpha = min(0.99, opacity * exp(0)) -> colour * 0.99 (process is front-to-back with T0 = 1)
    r = oracle.render(s, cam.uniforms(W, H), W, H)
    np.testing.assert_allclose(r["rgbf"][64, 64], 0.99 * g[8:11], rtol=1e-6)
    # a pixel d px away on the axis: alpha = opacity * exp(-0.5 d^2 / var)
    d = 9
    a = g[11] * np.exp(-0.5 * d * d / var)
    np.testing.assert_allclose(r["rgbf"][64, 64 + d], a * g[8:11], rtol=2e-5)


def test_culls_and_thresholds(oracle):
    W = H = 64
    cam = _front_camera(W, H, 64.0)
    u = cam.uniforms(W, H)
    # behind the near limit view.z <= 0.2 (process_gaussians.wgsl:120)
    assert oracle.preprocess(_one_splat((0, 0, 0.2), np.log([0.01] * 3)), u, W, H)[1][0] == 0
    assert oracle.preprocess(_one_splat((0, 0, 0.21), np.log([0.01] * 3)), u, W, H)[1][0] > 0
    # NDC cull |x| >= 1.1: x/z * 2f/W >= 1.1  <=>  x >= 0.55 z at f = W
    assert oracle.preprocess(_one_splat((0.56 * 2, 0, 2.0), np.log([0.01] * 3)), u, W, H)[1][0] == 0
    assert oracle.preprocess(_one_splat((0.54 * 2, 0, 2.0), np.log([0.01] * 3)), u, W, H)[1][0] > 0
    # alpha floor 1/255: a splat with opacity below it contributes nothing anywhere
    faint = _one_splat((0, 0, 2.0), np.log([0.2] * 3), opacity=float(np.log((1 / 300) / (1 - 1 / 300))))
    assert not oracle.render(faint, u, W, H)["rgbf"].any()


def test_column_aliasing_quirk_is_reproduced(oracle):
    """rect.max.x may be ntx+1; tile id y*ntx+ntx lands in column 0 of the next row, so a
    wide splat appears twice in those lists (write_tile_ids.wgsl:26-31)."""
    W, H = 64, 48
    cam = _front_camera(W, H, 64.0)
    s = _one_splat((0, 0, 1.0), np.log([1.0] * 3))  # covers the whole screen
    r = oracle.render(s, cam.uniforms(W, H), W, H)
    ntx, nty = 4, 3
    np.testing.assert_array_equal(r["gdata"][0, 12:16], [0, 0, ntx + 1, nty + 1])
    assert r["num_intersections"] == (ntx + 1) * (nty + 1)
    tiles = r["sorted_keys"] // 1000
    assert (tiles == ntx).sum() == 2  # tile (1,0): once as itself, once as the alias of (0, ntx)
    assert (tiles >= ntx * nty).sum() == ntx + 1 + 1  # row nty (ntx+1 instances) + the alias of (nty-1, ntx)
    lens = np.diff(np.concatenate([[0], r["ranges"]]))
    assert lens[0] == 1 and lens[ntx] == 2 and lens.sum() == r["ranges"][-1] <= r["num_intersections"]


def test_slab_counts_partition_the_full_frame(oracle):
    from gsplat import synth
    n, W, H = 5000, 320, 160
    s = scene(n)
    u = synth.orbit_camera(2, W, H).uniforms(W, H)
    full = oracle.render(s, u, W, H)
    tot = 0
    parts = []
    for c0, c1 in [(0, 3), (3, 4), (4, 11), (11, 20)]:
        r = oracle.render(s, u, W, H, cols=(c0, c1))
        tot += r["num_intersections"]
        parts.append(r["rgba8"][:, c0 * 16:min(W, c1 * 16)])
    assert tot == full["num_intersections"]
    np.testing.assert_array_equal(np.concatenate(parts, axis=1), full["rgba8"])


# ---- golden vectors --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))))
def test_golden_vectors(oracle, path):
    from gsplat import synth
    g = np.load(path, allow_pickle=False)
    n, W, H, ts, step = (int(v) for v in g["params"])
    s = synth.bicycle_like(n)
    assert _sha(s) == str(g["input_sha256"]), "synthetic scene generator drifted"
    u = synth.orbit_camera(step, W, H).uniforms(W, H)
    np.testing.assert_array_equal(u, g["uniforms"])
    r = oracle.render(s, u, W, H, ts)
    assert r["num_intersections"] == int(g["num_intersections"])
    np.testing.assert_array_equal(r["tile_counts"], g["tile_counts"])
    assert _sha(r["gdata"]) == str(g["gdata_sha256"])
    np.testing.assert_array_equal(r["sorted_keys"], g["sorted_keys"])
    np.testing.assert_array_equal(r["sorted_values"], g["sorted_values"])
    np.testing.assert_array_equal(r["ranges"], g["ranges"])
    np.testing.assert_array_equal(r["rgba8"], g["rgba8"])
    assert _sha(r["rgbf"]) == str(g["rgbf_sha256"])


def test_offscreen_splats_clamp_into_edge_columns(oracle):
    """getRect (process_gaussians.wgsl:305-313) clamps tile columns to [0, ntx]: a splat that passes the NDC cull
    (|x| < 1.1) but lies entirely left of the canvas still gets an instance in column 0, one entirely right of it
    lands in column ntx, which aliases to column 0 of the next row -- both belong to the slab that owns column 0."""
    W, H = 320, 160
    cam = _front_camera(W, H, 320.0)
    u = cam.uniforms(W, H)
    z = 4.0
    left = _one_splat((-1.05 * 0.5 * z, 0, z), np.log([0.002] * 3))   # ndc.x = -1.05 -> px = -8
    right = _one_splat((1.05 * 0.5 * z, 0, z), np.log([0.002] * 3))   # ndc.x = +1.05 -> px = 328
    gl, cl = oracle.preprocess(left, u, W, H)
    gr, cr = oracle.preprocess(right, u, W, H)
    rows_l, rows_r = int(gl[0, 15] - gl[0, 13]), int(gr[0, 15] - gr[0, 13])
    assert list(gl[0, 12:16:2]) == [0, 1] and cl[0] == rows_l >= 1      # one column: 0
    assert list(gr[0, 12:16:2]) == [20, 21] and cr[0] == rows_r >= 1    # one column: ntx = 20 (the alias column)
    for s, full in ((left, cl[0]), (right, cr[0])):
        assert oracle.preprocess(s, u, W, H, cols=(0, 5))[1][0] == full    # the owner of column 0 keeps it
        assert oracle.preprocess(s, u, W, H, cols=(5, 20))[1][0] == 0      # every other slab drops it
