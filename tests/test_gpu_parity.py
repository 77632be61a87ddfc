"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Integer outputs (tile counts, offsets, rects, keys, sorted values, ranges) must be bit-exact; the
GaussianData floats are bit-exact too (canonical semantics, oracle/gs_oracle.c header); the image is
bit-exact in GS_FLAG_EXACT_BLEND mode and within 1e-4 per channel in the default fused mode.
"""
import numpy as np
import pytest

from conftest import scene

pytestmark = pytest.mark.gpu


from gpu_checks import check_product_lists, check_stages as _check_stages, make_renderer as _mk, orbit_uniforms as _uniforms


@pytest.mark.parametrize("n,W,H,ts", [(10240, 256, 256, 16), (10000, 256, 256, 16), (3001, 200, 120, 16),
                                       (20000, 320, 192, 8), (20000, 320, 200, 32)])
def test_frame_exact_mode(oracle, n, W, H, ts):
    from gsplat import _abi
    s, u = scene(n), _uniforms(W, H)
    ref = oracle.render(s, u, W, H, ts)
    r = _mk(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(u, debug=True)
    r.wait()
    _check_stages(r, ref, exact_image=True)
    r.render_uniforms(u)  # the product path (no debug copies, no gdata clear)
    r.wait()
    _check_stages(r, ref, exact_image=True, debug=False)
    r.set_option(_abi.GS_OPT_TILE_CULL, 0)  # product frames with the reference's binning, in both emission orders
    for order in (1, 0):  # gaussian-index order + full-key sort; depth-ordered emission + tile-only instance sort
        r.set_option(_abi.GS_OPT_EMIT_ORDER, order)
        r.render_uniforms(u)
        r.wait()
        assert r.stats()["tight_binning"] == 0 and r.stats()["depth_ordered"] == (1 - order)
        _check_stages(r, ref, exact_image=True, debug=False)
    r.set_option(_abi.GS_OPT_TILE_CULL, 1)  # back to the tight row pipeline: same image, subset lists
    r.render_uniforms(u)
    r.wait()
    assert r.stats()["tight_binning"] == 1
    _check_stages(r, ref, exact_image=True, debug=False)
    r.destroy()


@pytest.mark.parametrize("n,W,H,ts", [(10240, 256, 256, 16), (60000, 640, 360, 16)])
def test_frame_fused_mode(oracle, n, W, H, ts):
    s, u = scene(n), _uniforms(W, H, step=11)
    ref = oracle.render(s, u, W, H, ts, want_illcond=True)
    r = _mk(s, W, H, ts)
    r.render_uniforms(u, debug=True)
    r.wait()
    _check_stages(r, ref, exact_image=False)
    r.render_uniforms(u)
    r.wait()
    _check_stages(r, ref, exact_image=False, debug=False)
    r.destroy()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 191, 192, 193, 255, 256, 257, 449, 1300])
def test_blend_id_stream_list_lengths(oracle, n):
    """The blend reads a tile's list 192 entries ahead and queues the entries of its 8x8 block in a 256-entry ring (k_blend.hip):
    lists whose lengths sit on every boundary of that machinery (one step of the stream, the lookahead, the ring, a ring that wraps
    several times).  One tile and a half: n faint splats over the whole 16x16 tile (every block's bit set: dense survivors) plus
    n / 3 small ones on single blocks of the neighbouring tile (sparse bits: the stream needs several steps per batch).  Low
    opacities keep every pixel alive to the end of its list.  EXACT: bit-equal to the oracle; fused: within 1e-4."""
    from gsplat import _abi
    from gpu_checks import check_image
    W, H, ts = 32, 16, 16
    rng = np.random.Generator(np.random.Philox(key=[91, n]))
    m = max(n // 3, 1)
    s = np.zeros((n + m, 80), dtype=np.float32)
    # big, faint: centre somewhere in tile 0, sigma ~ 10 pixels
    px = np.concatenate([rng.uniform(2.0, 14.0, n), 16.0 + rng.uniform(1.0, 15.0, m)]).astype(np.float32)
    py = np.concatenate([rng.uniform(2.0, 14.0, n), rng.uniform(1.0, 15.0, m)]).astype(np.float32)
    s[:, 0] = 2.0 * px / W - 1.0
    s[:, 1] = 2.0 * py / H - 1.0
    s[:, 2] = 1.0 + rng.uniform(0.0, 3.0, n + m).astype(np.float32)  # depth buckets spread: the list order is exercised too
    s[:n, 4:7] = np.log(rng.uniform(0.4, 0.9, (n, 3))).astype(np.float32)
    s[n:, 4:7] = np.log(rng.uniform(0.02, 0.05, (m, 3))).astype(np.float32)
    s[:, 8] = 1.0
    s[:, 8:12] += 0.2 * rng.standard_normal((n + m, 4)).astype(np.float32)
    s[:n, 12] = rng.uniform(-5.0, -3.5, n).astype(np.float32)   # opacity 0.007 .. 0.03: above 1/255, far from saturating a pixel
    s[n:, 12] = rng.uniform(-3.0, -1.0, m).astype(np.float32)
    s[:, 16:19] = rng.uniform(0.2, 1.5, (n + m, 3)).astype(np.float32)
    u = np.zeros(40, dtype=np.float32)
    u[0] = u[5] = u[10] = u[15] = 1.0
    u[16] = u[21] = u[26] = u[31] = 1.0
    u[35] = u[36] = 0.5
    u[37], u[38] = 16.0, 8.0   # focal: W / 2, H / 2 pixels per unit at depth 1
    u[39] = 1.0
    ref = oracle.render(s, u, W, H, ts, want_illcond=True)
    assert ref["ranges"][0] >= min(n, 1)  # tile 0 really holds the long list
    for flags, exact in ((_abi.GS_FLAG_EXACT_BLEND, True), (0, False)):
        r = _mk(s, W, H, ts, flags=flags)
        r.render_uniforms(u)
        r.wait()
        check_product_lists(r, ref, oracle, W, H, ts)
        check_image(r, ref, exact, max_ill=0.5)
        r.destroy()


def test_row_item_arena_overflow_regrows_and_rerenders(oracle):
    """The tight projection hands out row-item slots from an arena of max(2 N, 1 Mi) slots (16 sharded bump cursors); a frame that
    needs more is flagged, gs_wait grows the arena (and here the value capacity too) and renders the frame again.  100 000 large, faint
    splats on a 512x512 canvas of 8-pixel tiles take about fourteen tile rows each: 1.1+ M slots, 15+ M instances."""
    from gsplat import _abi
    from gpu_checks import check_image
    W = H = 512
    ts = 8
    n = 100000
    rng = np.random.Generator(np.random.Philox(key=[123, 7]))
    s = np.zeros((n, 80), dtype=np.float32)
    s[:, 0:2] = rng.uniform(-0.95, 0.95, (n, 2)).astype(np.float32)
    s[:, 2] = 1.0 + rng.uniform(0.0, 4.0, n).astype(np.float32)
    s[:, 4:7] = np.log(rng.uniform(0.07, 0.12, (n, 3)) * s[:, 2:3]).astype(np.float32)  # ~18-30 pixels of sigma at every depth
    s[:, 8] = 1.0
    s[:, 8:12] += 0.3 * rng.standard_normal((n, 4)).astype(np.float32)
    s[:, 12] = rng.uniform(-3.5, -2.5, n).astype(np.float32)  # opacity 0.03 .. 0.08
    s[:, 16:19] = rng.uniform(0.2, 1.5, (n, 3)).astype(np.float32)
    u = np.zeros(40, dtype=np.float32)
    u[0] = u[5] = u[10] = 1.0
    u[15] = 1.0
    # proj: x, y pass through, w = z (a plain perspective divide), so that splats keep their screen size over the depth range
    u[16] = u[21] = 1.0
    u[26] = 1.0
    u[27] = 1.0  # row 3 (w) takes z: column-major m[2*4+3]
    u[35] = u[36] = 1.0
    u[37] = u[38] = 256.0
    u[39] = 1.0
    ref = oracle.render(s, u, W, H, ts, want_illcond=True)
    r = _mk(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
    cap0 = r.stats()["row_capacity"]
    r.render_uniforms(u)
    r.wait()
    st = r.stats()
    assert st["tight_binning"] == 1
    assert st["num_row_slots"] > cap0, "the scene is meant to outgrow the default arena (%d slots of %d)" % (st["num_row_slots"], cap0)
    assert st["row_capacity"] >= st["num_row_slots"] and st["capacity"] >= st["num_intersections"]
    check_product_lists(r, ref, oracle, W, H, ts)
    check_image(r, ref, True)
    r.render_uniforms(u)  # and again with the grown arrays: nothing is flagged any more
    r.wait()
    assert r.stats()["row_capacity"] == st["row_capacity"]
    check_image(r, ref, True)
    r.destroy()
    # the same overflow in an EARLIER frame of a batch (two frames enqueued, one wait): reported, arrays grown, last frame complete
    r = _mk(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.set_option(_abi.GS_OPT_FRAMES_IN_FLIGHT, 1)
    r.render_uniforms(u)
    r.render_uniforms(u)
    with pytest.raises(_abi.GsError) as e:
        r.wait()
    assert e.value.code == -9
    st2 = r.stats()
    assert st2["row_capacity"] >= st2["num_row_slots"] > cap0 and st2["truncated_frames"] == 1
    check_image(r, ref, True)
    r.destroy()


def test_row_items_of_very_long_runs(oracle):
    """A row item stores the sub-block intervals of its run in 8 bits each (gs_tight.h); a run of more than 127 tiles has more
    sub-block columns than that and is flagged whole (conservative).  A 2032x32 canvas of 8-pixel tiles (254 tile columns, the widest
    the row pipeline takes) with splats stretched over most of its width, mixed with small ones; tile 16 on 4064x32 likewise."""
    from gsplat import _abi
    from gpu_checks import check_image
    for W, H, ts in ((2032, 32, 8), (4064, 32, 16)):
        n = 600
        rng = np.random.Generator(np.random.Philox(key=[321, ts]))
        s = np.zeros((n, 80), dtype=np.float32)
        s[:, 0] = rng.uniform(-0.9, 0.9, n).astype(np.float32)
        s[:, 1] = rng.uniform(-0.9, 0.9, n).astype(np.float32)
        s[:, 2] = 1.0 + rng.uniform(0.0, 2.0, n).astype(np.float32)
        long_ = rng.uniform(0.0, 1.0, n) < 0.3
        sx = np.where(long_, rng.uniform(0.15, 0.6, n), rng.uniform(0.002, 0.01, n))   # in units of the half width: up to ~1200 pixels of sigma
        sy = rng.uniform(0.05, 0.4, n)                                                 # 1 .. 6 pixels of sigma (half height = 16 pixels)
        s[:, 4] = np.log(sx * s[:, 2]).astype(np.float32)
        s[:, 5] = np.log(sy * s[:, 2]).astype(np.float32)
        s[:, 6] = np.log(0.01).astype(np.float32)
        s[:, 8] = 1.0
        s[:, 11] = (0.02 * rng.standard_normal(n)).astype(np.float32)  # a slight roll: the runs are not axis-aligned
        s[:, 12] = rng.uniform(-3.0, 1.0, n).astype(np.float32)
        s[:, 16:19] = rng.uniform(0.2, 1.5, (n, 3)).astype(np.float32)
        u = np.zeros(40, dtype=np.float32)
        u[0] = u[5] = u[10] = u[15] = 1.0
        u[16] = u[21] = 1.0
        u[26] = 1.0
        u[27] = 1.0
        u[35] = u[36] = 1.0
        u[37], u[38] = W / 2.0, H / 2.0
        u[39] = 1.0
        ref = oracle.render(s, u, W, H, ts, want_illcond=True)
        rect = ref["gdata"].reshape(-1, 16)[:, 12:16].astype(np.int64)
        vis = ref["tile_counts"] > 0
        assert ((rect[vis, 2] - rect[vis, 0]) > 200).sum() >= 20  # rects wider than 200 tiles really occur
        for flags, exact in ((_abi.GS_FLAG_EXACT_BLEND, True), (0, False)):
            r = _mk(s, W, H, ts, flags=flags)
            r.render_uniforms(u)
            r.wait()
            assert r.stats()["tight_binning"] == 1
            check_product_lists(r, ref, oracle, W, H, ts)
            check_image(r, ref, exact, max_ill=0.5)
            r.destroy()


@pytest.mark.parametrize("shape", [(4080, 32, 16), (32, 4080, 16), (2040, 2040, 8), (2048, 64, 8)])
def test_row_pipeline_at_its_grid_limits(oracle, shape):
    """Row items hold tile rows and columns in 8 bits: canvases of exactly 255 tile columns / rows (the widest and tallest the
    row pipeline takes: digit 255 is the row sort's and the expansion's `hole`), 255 x 255 tiles, and 256 columns (one too many: the
    product frame must fall back to the reference's binning by itself).  EXACT image bit-equal, lists checked where tight."""
    from conftest import scene
    from gsplat import _abi
    from gpu_checks import check_image
    W, H, ts = shape
    s = scene(20000)
    u = _uniforms(W, H, step=13)
    ref = oracle.render(s, u, W, H, ts)
    r = _mk(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(u)
    r.wait()
    ntx, nty = -(-W // ts), -(-H // ts)
    assert r.stats()["tight_binning"] == (1 if max(ntx, nty) <= 255 else 0)
    if r.stats()["tight_binning"]:
        check_product_lists(r, ref, oracle, W, H, ts)
    check_image(r, ref, True)
    r.destroy()


def test_row_sort_tiles_of_single_slot_gaussians(oracle):
    """A row-sort tile (4096 slots) whose gaussians have ONE slot each holds 4096 gaussians -- the most its owner search has to
    cover (k_rows.hip: 64 samples + one segment).  A 17-pixel wide canvas of 32-pixel tiles: one tile column, almost every visible
    gaussian in one tile row.  Found by tools/fuzz_product.py when the search still assumed 3072-slot tiles: misordered lists."""
    from gsplat import _abi, synth
    from gpu_checks import check_image
    W, H, ts = 17, 334, 32
    s = synth.bicycle_like(60000, synth.BASE_SEED + 122)
    u = synth.orbit_camera(5, W, H).uniforms(W, H)
    ref = oracle.render(s, u, W, H, ts)
    r = _mk(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(u)
    r.wait()
    st = r.stats()
    assert st["tight_binning"] == 1 and st["num_row_slots"] < 1.2 * st["num_visible"] and st["num_visible"] > 30000
    check_product_lists(r, ref, oracle, W, H, ts)
    check_image(r, ref, True)
    r.destroy()


def test_fused_blend_splat_centres_on_pixel_centres(oracle):
    """The fused blend's loop drops the reference's `power <= 0` test (compute_tiles.wgsl:61) for batches whose conics are all
    positive definite: there the power can only exceed 0 by rounding, which happens where dx, dy are (almost) 0.  This scene puts
    4096 splat centres within ~1e-5 pixel of pixel centres (identity view / projection, dyadic positions), with strongly
    correlated, anisotropic conics and high opacities: the pixel under a centre sees power = -0 .. -1e-9.  The fused image must
    agree with the oracle within 1e-4 off the oracle's ill-conditioned pixels, in both binnings, and the EXACT one bit for bit."""
    from gsplat import _abi
    W = H = 256
    ts = 16
    rng = np.random.Generator(np.random.Philox(key=[77, 1]))
    g = 64
    s = np.zeros((g * g, 80), dtype=np.float32)
    kx, ky = np.meshgrid(np.arange(g), np.arange(g))
    px = (kx.ravel() * 3 + 32).astype(np.float32)  # pixel centres 32, 35, ... (integer pixel coordinates ARE the centres: compute_tiles.wgsl:40)
    py = (ky.ravel() * 3 + 32).astype(np.float32)
    s[:, 0] = 2.0 * px / W - 1.0  # ndc = pos / (1 + 1e-7): uv * W lands within ~1e-5 of the integer
    s[:, 1] = 2.0 * py / H - 1.0
    s[:, 2] = 1.0
    s[:, 4:7] = np.log(rng.uniform(0.004, 0.03, (g * g, 3))).astype(np.float32)  # 0.5 .. 4 pixels at focal 128, anisotropic
    s[:, 8:12] = rng.standard_normal((g * g, 4)).astype(np.float32)
    s[:, 12] = rng.uniform(1.0, 6.0, g * g).astype(np.float32)  # opacity 0.73 .. 0.998 (the 0.99 clamp included)
    s[:, 16:19] = rng.uniform(0.5, 2.0, (g * g, 3)).astype(np.float32)
    u = np.zeros(40, dtype=np.float32)
    u[0] = u[5] = u[10] = u[15] = 1.0       # view = I (column-major)
    u[16] = u[21] = u[26] = u[31] = 1.0     # proj = I: hom = (x, y, z, 1)
    u[35] = u[36] = 0.5                     # tan_fov
    u[37] = u[38] = 128.0                   # focal
    u[39] = 1.0
    ref = oracle.render(s, u, W, H, ts, want_illcond=True)
    uvw = ref["gdata"].view(np.float32).reshape(-1, 16)[:, 0] * np.float32(W)
    vis = ref["tile_counts"] > 0
    assert vis.sum() > 3500 and np.abs(uvw[vis] - np.round(uvw[vis])).max() < 1e-3  # the centres really sit on pixel centres
    from gpu_checks import check_image
    for flags, exact in ((_abi.GS_FLAG_EXACT_BLEND, True), (0, False)):
        r = _mk(s, W, H, ts, flags=flags)
        for debug in (True, False):  # the reference's binning, then the tight row pipeline
            r.render_uniforms(u, debug=debug)
            r.wait()
            if debug:
                _check_stages(r, ref, exact_image=True) if exact else None
            else:
                check_product_lists(r, ref, oracle, W, H, ts)
            rep = {}
            # the oracle flags every pixel whose power is EXACTLY 0 (half of the centres here) as ill-conditioned; no decision
            # actually flips on this scene, so the whole frame -- flagged pixels included -- is held to 1e-4
            check_image(r, ref, exact, max_ill=0.1, report=None if exact else rep)
            if not exact:
                assert rep["max_err"] <= 1e-4 and rep["rgba8_max_lsb"] <= 1, rep
        r.destroy()


@pytest.mark.parametrize("variant", [0, 8, 5 * 256])  # GS_OPT_BLEND_ABLATION: 0 = default (8x8-block waves), 8 = 4-wave workgroup per tile, 5<<8 = strip width 5
@pytest.mark.parametrize("exact", [True, False])
def test_blend_kernel_variants(oracle, variant, exact):
    from gsplat import _abi
    n, W, H = 60000, 640, 368
    s, u = scene(n), _uniforms(W, H, step=7)
    ref = oracle.render(s, u, W, H, 16, want_illcond=not exact)
    r = _mk(s, W, H, 16, flags=_abi.GS_FLAG_EXACT_BLEND if exact else 0)
    r.set_option(_abi.GS_OPT_BLEND_ABLATION, variant)
    r.render_uniforms(u)
    r.wait()
    _check_stages(r, ref, exact_image=exact, debug=False)
    r.destroy()


@pytest.mark.parametrize("variant", [0, 8])  # tile 32: 0 = sixteen 8x8-block waves per tile (default), 8 = one 1024-thread workgroup per tile
@pytest.mark.parametrize("exact", [True, False])
def test_blend_kernel_variants_tile32(oracle, variant, exact):
    from gsplat import _abi
    n, W, H = 60000, 640, 368
    s, u = scene(n), _uniforms(W, H, step=7)
    ref = oracle.render(s, u, W, H, 32, want_illcond=not exact)
    r = _mk(s, W, H, 32, flags=_abi.GS_FLAG_EXACT_BLEND if exact else 0)
    r.set_option(_abi.GS_OPT_BLEND_ABLATION, variant)
    r.render_uniforms(u)
    r.wait()
    _check_stages(r, ref, exact_image=exact, debug=False)
    assert r.stats()["num_processed"] > 0
    r.destroy()


def test_sort_kat_reference_testsort():
    """radix_sort/utils.ts:55-81: 8192 keys n-1-i must come out 0..n-1."""
    from gsplat import _abi
    n = 8192
    k, _ = _abi.sort_pairs(np.arange(n - 1, -1, -1, dtype=np.uint32))
    np.testing.assert_array_equal(k, np.arange(n, dtype=np.uint32))


@pytest.mark.parametrize("n,bits", [(1, 32), (63, 32), (4096, 32), (4097, 23), (100003, 32), (1 << 20, 25)])
def test_sort_pairs_stable(n, bits):
    from gsplat import _abi
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << bits, size=n, dtype=np.uint64).astype(np.uint32)
    keys[:: 3] = keys[0]  # many duplicates: stability matters
    vals = np.arange(n, dtype=np.uint32)
    k, v = _abi.sort_pairs(keys, vals, key_bits=bits)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(k, keys[order])
    np.testing.assert_array_equal(v, vals[order])


@pytest.mark.parametrize("n", [1, 511, 512, 4096, 4097, 262144 + 77, 3000001])
def test_scan_definition(n):
    """exclusive_scan.ts:105-112: out[0]=0, out[i]=in[i-1]+out[i-1]; returns out[n-1]+in[n-1]."""
    from gsplat import _abi
    rng = np.random.default_rng(n)
    data = rng.integers(0, 50, size=n, dtype=np.uint32)
    out, total = _abi.exclusive_scan(data)
    ref = np.concatenate([[0], np.cumsum(data[:-1], dtype=np.uint64)]).astype(np.uint32)
    np.testing.assert_array_equal(out, ref)
    assert total == int(data.sum())


def test_determinism_and_reuse(oracle):
    """Same frame twice => identical bytes; a second camera on the same ctx matches the oracle."""
    from gsplat import _abi
    n, W, H = 50000, 512, 288
    s = scene(n)
    r = _mk(s, W, H, 16)
    u = _uniforms(W, H, step=2)
    r.render_uniforms(u); r.wait()
    a = r.read_rgba8()
    r.render_uniforms(_uniforms(W, H, step=30)); r.wait()
    r.render_uniforms(u); r.wait()
    np.testing.assert_array_equal(a, r.read_rgba8())
    u2 = _uniforms(W, H, step=40)
    ref = oracle.render(s, u2, W, H, 16)
    r.render_uniforms(u2); r.wait()
    check_product_lists(r, ref, oracle, W, H, 16)
    r.destroy()


def test_edge_cases(oracle):
    from gsplat import _abi
    W, H = 128, 96
    u = _uniforms(W, H)
    # no gaussians at all
    r = _mk(np.zeros((0, 80), np.float32), W, H)
    r.render_uniforms(u); r.wait()
    assert r.stats()["num_intersections"] == 0
    assert not r.read_rgba8()[..., :3].any() and (r.read_rgba8()[..., 3] == 255).all()
    r.destroy()
    # everything behind the camera
    s = scene(2000).copy()
    far = oracle.preprocess(s, u, W, H)[1] > 0
    s2 = s[~far][:500]
    r = _mk(s2, W, H)
    r.render_uniforms(u, debug=True); r.wait()
    assert r.stats()["num_intersections"] == 0 and r.stats()["num_visible"] == 0
    np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_RANGES), np.zeros(8 * 6, np.uint32))
    r.destroy()
    # one huge splat covering every tile (+ the aliased column/row one past the grid)
    one = scene(1).copy()
    one[0, 0:3] = 0.0
    one[0, 4:7] = np.log(2.0)
    ref = oracle.render(one, u, W, H)
    assert ref["num_intersections"] == (8 + 1) * (6 + 1)
    r = _mk(one, W, H, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(u, debug=True); r.wait()
    _check_stages(r, ref, exact_image=True)
    r.destroy()


def test_non_finite_splats(oracle):
    """Poisoned records (zero quaternion -> NaN conic, infinite scale, NaN position, NaN SH, huge opacity logit) must not fault
    and must bin, sort and blend like the canonical semantics (NaN compares false, f32->i32 saturates, NaN -> 0)."""
    from gsplat import _abi
    n, W, H = 6000, 192, 128
    s = scene(n).copy()
    s[10, 8:12] = 0.0                      # |q| = 0: 0/0 in the rotation
    s[11, 4:7] = np.float32(80.0)          # exp(80) overflows to +inf scales
    s[12, 0] = np.float32(np.nan)          # NaN position
    s[13, 16:19] = np.float32(np.nan)      # NaN SH DC
    s[14, 12] = np.float32(1e30)           # sigmoid(+huge)
    s[15, 12] = np.float32(-1e30)          # sigmoid(-huge)
    s[16, 4:7] = np.float32(-200.0)        # exp(-200) underflows to 0 scales
    s[17, 8:12] = np.float32(np.inf)       # inf/inf in the rotation
    u = _uniforms(W, H, step=5)
    ref = oracle.render(s, u, W, H, 16, want_illcond=True)
    for flags in (_abi.GS_FLAG_EXACT_BLEND, 0):
        r = _mk(s, W, H, 16, flags=flags)
        for debug in (True, False):
            r.render_uniforms(u, debug=debug); r.wait()
            if debug:
                np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_TILE_COUNTS), ref["tile_counts"])
                np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_KEYS), ref["sorted_keys"])
                np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_VALUES), ref["sorted_values"])
                np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_RANGES), ref["ranges"])
            else:
                check_product_lists(r, ref, oracle, W, H, 16)
            img = r.read_rgba8()
            if flags:
                np.testing.assert_array_equal(img, ref["rgba8"])
            else:
                ill = ref["illcond"].astype(bool)
                d8 = np.abs(img.astype(np.int32) - ref["rgba8"].astype(np.int32))
                assert d8[~ill].max(initial=0) <= 1
        r.destroy()


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_configs_exact(oracle, seed):
    """Seeded random canvases (odd sizes), tile sizes, scene sizes, cameras and scale modifiers: every stage bit-equal."""
    from gsplat import _abi, synth
    rng = np.random.default_rng(1000 + seed)
    ts = int(rng.choice([8, 16, 16, 32]))
    W, H = int(rng.integers(33, 420)), int(rng.integers(17, 300))
    n = int(rng.integers(1, 40000))
    s = synth.bicycle_like(n, seed=synth.BASE_SEED + 50 + seed)
    u = synth.orbit_camera(int(rng.integers(0, 64)), W, H, radius=float(rng.uniform(2.0, 9.0)), height=float(rng.uniform(-2.0, 3.0))).uniforms(W, H).copy()
    u[39] = np.float32(rng.choice([0.25, 1.0, 1.0, 3.0]))  # scale_modifier
    ref = oracle.render(s, u, W, H, ts)
    r = _mk(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(u, debug=True); r.wait()
    _check_stages(r, ref, exact_image=True)
    r.set_option(_abi.GS_OPT_EMIT_ORDER, int(rng.integers(0, 3)))
    r.render_uniforms(u); r.wait()
    _check_stages(r, ref, exact_image=True, debug=False)
    r.destroy()


def test_reupload_and_two_contexts_in_threads(oracle):
    """gs_upload_splats may be called again on a live context (smaller, then larger scene), and two contexts driven from two
    host threads at once give their own frames (the ABI is thread-safe across contexts, gs_abi.h)."""
    import threading
    from gsplat import _abi
    W, H = 256, 160
    u = _uniforms(W, H, step=6)
    sa, sb = scene(9000), scene(30000)[9000:]
    ra = oracle.render(sa, u, W, H, 16)
    rb = oracle.render(sb, u, W, H, 16)
    r = _mk(sb, W, H, 16, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(u); r.wait()
    np.testing.assert_array_equal(r.read_rgba8(), rb["rgba8"])
    arr = np.ascontiguousarray(sa, dtype=np.float32)
    _abi.check(r._L.gs_upload_splats(r._ctx, arr.ctypes.data, arr.shape[0]))  # smaller scene into the same context
    r.render_uniforms(u); r.wait()
    np.testing.assert_array_equal(r.read_rgba8(), ra["rgba8"])
    check_product_lists(r, ra, oracle, W, H, 16)
    arr = np.ascontiguousarray(sb, dtype=np.float32)
    _abi.check(r._L.gs_upload_splats(r._ctx, arr.ctypes.data, arr.shape[0]))  # and a larger one again
    r.render_uniforms(u); r.wait()
    np.testing.assert_array_equal(r.read_rgba8(), rb["rgba8"])
    r2 = _mk(sa, W, H, 16, flags=_abi.GS_FLAG_EXACT_BLEND)
    out = {}

    def work(name, rr, n_frames):
        img = None
        for _ in range(n_frames):
            rr.render_uniforms(u); rr.wait()
            img = rr.read_rgba8()
        out[name] = img

    ta = threading.Thread(target=work, args=("b", r, 12)), threading.Thread(target=work, args=("a", r2, 12))
    for t in ta:
        t.start()
    for t in ta:
        t.join()
    np.testing.assert_array_equal(out["b"], rb["rgba8"])
    np.testing.assert_array_equal(out["a"], ra["rgba8"])
    r.destroy(); r2.destroy()


def test_capacity_growth(oracle):
    """A frame that overflows the (key,value) capacity is re-rendered after growing it."""
    from gsplat import _abi
    n, W, H = 20000, 256, 256
    s, u = scene(n), _uniforms(W, H)
    ref = oracle.render(s, u, W, H)
    assert ref["num_intersections"] > 4096
    r = _mk(s, W, H, max_intersections=4096, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(u, debug=True)
    r.wait()
    _check_stages(r, ref, exact_image=True)
    r.destroy()


def test_slab_union_equals_full_frame(oracle):
    """tile-column slabs only filter; their union is the single-GPU image byte for byte."""
    from gsplat import _abi
    n, W, H = 40000, 512, 256
    s, u = scene(n), _uniforms(W, H, step=21)
    full = _mk(s, W, H, flags=_abi.GS_FLAG_EXACT_BLEND)
    full.render_uniforms(u); full.wait()
    img = full.read_rgba8()
    ntx = 32
    bounds = [0, 5, 16, 17, 32]
    parts = []
    for c0, c1 in zip(bounds[:-1], bounds[1:]):
        r = _mk(s, W, H, flags=_abi.GS_FLAG_EXACT_BLEND, cols=(c0, c1))
        r.render_uniforms(u, debug=True); r.wait()
        ref = oracle.render(s, u, W, H, 16, cols=(c0, c1))
        _check_stages(r, ref, exact_image=True)
        r.render_uniforms(u); r.wait()
        _check_stages(r, ref, exact_image=True, debug=False)
        parts.append(r.read_rgba8())
        r.set_option(_abi.GS_OPT_EMIT_ORDER, 0)  # depth-ordered emission inside a slab
        r.render_uniforms(u); r.wait()
        assert r.stats()["depth_ordered"] == 1
        _check_stages(r, ref, exact_image=True, debug=False)
        np.testing.assert_array_equal(r.read_rgba8(), parts[-1])
        r.destroy()
    np.testing.assert_array_equal(np.concatenate(parts, axis=1), img)
    full.destroy()


def test_debug_views():
    """GS_OPT_DEBUG_VIEW: the developer views left commented out in compute_tiles.wgsl:35-38,67-70."""
    from gsplat import _abi
    n, W, H, ts = 20000, 200, 120, 16
    s, u = scene(n), _uniforms(W, H, step=2)
    r = _mk(s, W, H, ts)
    r.render_uniforms(u); r.wait()
    plain = r.read_rgba8()
    rg = r.read_buffer(_abi.GS_BUF_RANGES).astype(np.int64)
    ln = np.diff(np.concatenate([[0], rg]))
    ntx = (W + ts - 1) // ts
    yy, xx = np.mgrid[0:H, 0:W]
    per_px = ln[(xx // ts) + (yy // ts) * ntx].astype(np.float32)

    def unorm(v):
        return np.floor(np.clip(v, 0.0, 1.0).astype(np.float32) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)

    r.set_option(_abi.GS_OPT_DEBUG_VIEW, 2)  # list length / 1000 as grey (:67)
    r.render_uniforms(u); r.wait()
    img = r.read_rgba8()
    g = unorm(per_px / np.float32(1000.0))
    np.testing.assert_array_equal(img[..., 0], g)
    np.testing.assert_array_equal(img[..., 1], g)
    np.testing.assert_array_equal(img[..., 3], 255)
    r.set_option(_abi.GS_OPT_DEBUG_VIEW, 1)  # tile borders in red (:35-38)
    r.render_uniforms(u); r.wait()
    img = r.read_rgba8()
    border = (xx % ts == ts - 1) | (yy % ts == ts - 1)
    assert (img[border] == np.array([255, 0, 0, 255], dtype=np.uint8)).all()
    np.testing.assert_array_equal(img[~border], plain[~border])
    r.set_option(_abi.GS_OPT_DEBUG_VIEW, 4)  # list length / 100 in red and green (:70)
    r.render_uniforms(u); r.wait()
    img = r.read_rgba8()
    np.testing.assert_array_equal(img[..., 0], unorm(per_px / np.float32(100.0)))
    assert (img[..., 2] == 0).all()
    r.set_option(_abi.GS_OPT_DEBUG_VIEW, 0)
    r.render_uniforms(u); r.wait()
    np.testing.assert_array_equal(r.read_rgba8(), plain)
    r.destroy()


def test_frames_in_flight_match_sequential(oracle):
    """gs_share_splats + PipelinedRenderer: three frames in flight over shared splats give the frames of a single context."""
    import gsplat
    from gsplat import _abi
    n, W, H = 50000, 512, 288
    s = scene(n)
    us = [_uniforms(W, H, step=k) for k in range(7)]
    seq = _mk(s, W, H)
    want = []
    for u in us:
        seq.render_uniforms(u); seq.wait()
        want.append(seq.read_rgba8())
    pr = gsplat.PipelinedRenderer(gsplat.Canvas(W, H), None, 0, gsplat.PackedGaussians(s), 16, frames_in_flight=3)
    slots = [pr.render_uniforms(u) for u in us[:3]]
    got = {}
    for k in range(3, len(us)):
        got[k - 3] = pr.read_rgba8(slots[(k - 3) % 3])  # frame k-3 before its slot is reused
        slots[(k - 3) % 3] = pr.render_uniforms(us[k])
    for k in range(len(us) - 3, len(us)):
        got[k] = pr.read_rgba8(slots[k % 3])
    for k in range(len(us)):
        np.testing.assert_array_equal(got[k], want[k])
    ref = oracle.render(s, us[-1], W, H, 16, want_illcond=True)
    d8 = np.abs(got[len(us) - 1].astype(np.int32) - ref["rgba8"].astype(np.int32))
    assert d8[~ref["illcond"].astype(bool)].max(initial=0) <= 1
    pr.destroy()
    seq.destroy()


def test_render_host_tickets(oracle):
    """gs_render_host / gs_wait_ticket (pipelined presentation): frames enqueued back to back, each with an asynchronous copy of its
    pixels into its own page-locked sink; tickets waited for out of order; every sink holds exactly what a sequential render of
    that camera reads back."""
    import ctypes
    from gsplat import _abi
    L = _abi.load()
    n, W, H = 60000, 640, 360
    s = scene(n)
    us = [_uniforms(W, H, step=k) for k in range(7)]
    seq = _mk(s, W, H, 16)
    seq.set_option(_abi.GS_OPT_FRAMES_IN_FLIGHT, 1)
    want = []
    for u in us:
        seq.render_uniforms(u); seq.wait()
        want.append(seq.read_rgba8())
    seq.destroy()
    r = _mk(s, W, H, 16)
    for u in us:  # capacities of every member of the ring
        r.render_uniforms(u)
    r.wait()
    nbytes = W * H * 4
    sinks, tickets = [], []
    for _ in us:
        p = ctypes.c_void_p()
        _abi.check(L.gs_host_alloc(nbytes, ctypes.byref(p)))
        sinks.append(p)
    for u, p in zip(us, sinks):
        t = ctypes.c_uint64()
        uu = np.ascontiguousarray(u, dtype=np.float32)
        _abi.check(L.gs_render_host(r._ctx, uu.ctypes.data, p, nbytes, ctypes.byref(t)))
        tickets.append(t.value)
    assert tickets == list(range(tickets[0], tickets[0] + len(us)))
    for k in (3, 0, 6, 1, 2, 5, 4):
        _abi.check(L.gs_wait_ticket(r._ctx, tickets[k]))
        got = np.ctypeslib.as_array(ctypes.cast(sinks[k], ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).reshape(H, W, 4)
        np.testing.assert_array_equal(got, want[k])
    with pytest.raises(_abi.GsError):
        _abi.check(L.gs_wait_ticket(r._ctx, tickets[-1] + 1))  # never issued
    r.wait()
    assert r.stats()["frames_in_flight"] >= 2
    r.destroy()
    for p in sinks:
        L.gs_host_free(p)


def test_blend_culls_change_no_bit():
    """The blend parks an entry only if it can still change a LIVE pixel of the block: its alpha >= 1/255 ellipse must reach the
    bounding box of the pixels that are not final yet, and Tmax (1 - alpha_lo) over that box must reach 1e-4.  Both tests are
    conservative, a final pixel ignores every entry and a live one skips what fails `T (1 - alpha) >= 1e-4` anyway: the frame with the
    culls (default) and without them (GS_OPT_BLEND_ABLATION 4) is the same frame, bit for bit, in BOTH blend modes -- on a scene deep
    enough that most blocks saturate (hundreds of splats per pixel), with opaque and faint splats, at tiles 16 and 32."""
    from gsplat import _abi
    n, W, H = 400000, 256, 144
    s = scene(n).copy()
    rng = np.random.default_rng(11)
    idx = rng.integers(0, n, n // 5)
    s[idx, 12] = rng.choice([6.0, 2.0, -4.0, -5.4], idx.size).astype(np.float32)  # opacity logits: opaque ... just above 1/255
    for ts in (16, 32):
        for flags in (0, _abi.GS_FLAG_EXACT_BLEND):
            r = _mk(s, W, H, ts, flags=flags)
            for step in (2, 33):
                u = _uniforms(W, H, step=step)
                r.set_option(_abi.GS_OPT_BLEND_ABLATION, 0)
                r.render_uniforms(u); r.wait()
                ev_cull = r.stats()["num_evaluated"]
                img = r.read_rgba8()
                f32 = r.read_buffer(_abi.GS_BUF_RGB_F32, np.float32).copy()
                r.set_option(_abi.GS_OPT_BLEND_ABLATION, 4)
                r.render_uniforms(u); r.wait()
                ev_all = r.stats()["num_evaluated"]
                np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_RGB_F32, np.float32).view(np.uint32), f32.view(np.uint32))
                np.testing.assert_array_equal(r.read_rgba8(), img)
                assert ev_cull < ev_all, (ts, flags, ev_cull, ev_all)  # the culls do remove work (10 % here; 75 % at config B, where most blocks saturate)
            r.destroy()


def test_frame_graph_replays_the_same_frames():
    """GS_OPT_FRAME_GRAPH: frames replayed from the captured hipGraph (uniforms patched into the projection's node) are the
    frames the directly issued launches give -- moving camera, an option change (re-capture), a capacity regrow on the first
    frame, a slab writing to a caller's buffer, and the ring of frames in flight."""
    import torch
    from gsplat import _abi
    n, W, H = 60000, 640, 368
    s = scene(n)
    us = [_uniforms(W, H, step=k) for k in range(6)]
    direct = _mk(s, W, H)
    want = []
    for u in us:
        direct.render_uniforms(u); direct.wait()
        want.append(direct.read_rgba8())
    g = _mk(s, W, H, max_intersections=4096)  # the first frame overflows: gs_wait grows the arrays and re-renders
    g.set_option(_abi.GS_OPT_FRAME_GRAPH, 1)
    g.set_option(_abi.GS_OPT_FRAMES_IN_FLIGHT, 1)
    for k, u in enumerate(us):
        g.render_uniforms(u); g.wait()
        np.testing.assert_array_equal(g.read_rgba8(), want[k])
    st = g.stats()
    assert st["graph_frames"] >= len(us), st["graph_frames"]
    assert st["num_intersections"] == direct.stats()["num_intersections"]
    # an option change drops the capture; the next frames come from a new one
    for r_ in (g, direct):
        r_.set_option(_abi.GS_OPT_TILE_CULL, 0)
    direct.render_uniforms(us[2]); direct.wait()
    g.render_uniforms(us[2]); g.wait()
    np.testing.assert_array_equal(g.read_rgba8(), direct.read_rgba8())
    assert g.stats()["tight_binning"] == 0
    g.set_option(_abi.GS_OPT_TILE_CULL, 1)
    # a debug frame (reference binning: another projection kernel, grid and outputs) between two replays of one capture: the
    # replay restores its own launch descriptor and notes (it used to patch the debug frame's into the captured node)
    g.render_uniforms(us[0]); g.wait()
    before = g.stats()["graph_frames"]
    g.render_uniforms(us[1]); g.render_uniforms(us[3], debug=True); g.wait()
    direct.set_option(_abi.GS_OPT_TILE_CULL, 1)
    direct.render_uniforms(us[3], debug=True); direct.wait()
    np.testing.assert_array_equal(g.read_rgba8(), direct.read_rgba8())
    g.render_uniforms(us[4]); g.wait()
    np.testing.assert_array_equal(g.read_rgba8(), want[4])
    assert g.stats()["graph_frames"] == before + 2 and g.stats()["tight_binning"] == 1
    # frames back to back without a wait, three in flight (the shadows of the ring replay their own captures)
    g.set_option(_abi.GS_OPT_FRAMES_IN_FLIGHT, 3)
    before = g.stats()["graph_frames"]
    for u in us:
        g.render_uniforms(u)
    g.wait()
    np.testing.assert_array_equal(g.read_rgba8(), want[-1])
    assert g.stats()["graph_frames"] == before + len(us)
    g.destroy()
    # a slab on the caller's stream, blending into the caller's buffer
    c0, c1 = 10, 27
    stream = torch.cuda.Stream()
    sl = _mk(s, W, H, cols=(c0, c1), stream=stream.cuda_stream)
    sl.set_option(_abi.GS_OPT_FRAME_GRAPH, 1)
    out = torch.zeros((H, (c1 - c0) * 16, 4), dtype=torch.uint8, device="cuda")
    for k, u in enumerate(us):
        sl.render_uniforms(u, out_ptr=out.data_ptr())
    sl.wait()
    np.testing.assert_array_equal(out.cpu().numpy(), want[-1][:, c0 * 16:c1 * 16])
    assert sl.stats()["graph_frames"] == len(us)
    sl.destroy()
    direct.destroy()


import glob as _glob
import os as _os


@pytest.mark.parametrize("path", sorted(_glob.glob(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "*.npz"))))
def test_gpu_matches_committed_golden_vectors(path):
    """The HIP path against tests/golden/*.npz directly (no oracle in the loop)."""
    import hashlib
    from gsplat import _abi, synth
    g = np.load(path, allow_pickle=False)
    n, W, H, ts, step = (int(v) for v in g["params"])
    s = synth.bicycle_like(n)
    r = _mk(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(g["uniforms"], debug=True)
    r.wait()
    assert r.stats()["num_intersections"] == int(g["num_intersections"])
    np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_TILE_COUNTS), g["tile_counts"])
    gd = r.read_buffer(_abi.GS_BUF_GAUSSIAN_DATA)
    assert hashlib.sha256(gd.tobytes()).hexdigest() == str(g["gdata_sha256"])
    np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_KEYS), g["sorted_keys"])
    np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_VALUES), g["sorted_values"])
    np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_RANGES), g["ranges"])
    np.testing.assert_array_equal(r.read_rgba8(), g["rgba8"])
    f32 = r.read_buffer(_abi.GS_BUF_RGB_F32, np.float32)
    assert hashlib.sha256(f32.tobytes()).hexdigest() == str(g["rgbf_sha256"])
    r.destroy()


def test_ply_file_to_frame(tmp_path, oracle):
    """.ply on disk -> native loader (gs_ply_load / gs_upload_ply) -> frame, against the oracle
    on the records the file was written from."""
    import ctypes
    import sys
    sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools"))
    from ply_bench import write_ply
    import gsplat
    from gsplat import _abi
    n, W, H = 12000, 320, 192
    s, u = scene(n), _uniforms(W, H, step=14)
    path = str(tmp_path / "scene.ply")
    write_ply(path, s)
    ref = oracle.render(s, u, W, H, 16)
    pg = gsplat.PackedGaussians.from_ply(path)
    assert pg.numGaussians == n and pg.sphericalHarmonicsDegree == 3
    np.testing.assert_array_equal(np.asarray(pg.gaussiansBuffer).view(np.uint32), s.view(np.uint32))
    r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, flags=_abi.GS_FLAG_EXACT_BLEND | _abi.GS_FLAG_F32_TAP)
    r.render_uniforms(u)
    r.wait()
    scene_from_records = r.read_buffer(12)  # (tests only) the resident scene arrays, as gs_upload_splats laid them out
    # replace the scene through the streaming path: file -> pinned chunks -> device arrays, no 320-byte records anywhere
    cnt = ctypes.c_uint64()
    _abi.check(_abi.load().gs_upload_ply(r._ctx, path.encode(), ctypes.byref(cnt)))
    assert cnt.value == n
    r.render_uniforms(u, debug=True)
    r.wait()
    np.testing.assert_array_equal(r.read_buffer(12), scene_from_records)  # byte-identical device scene
    _check_stages(r, ref, exact_image=True)
    r.destroy()


def test_ply_streaming_chunks_uchar_and_low_degree(tmp_path, oracle):
    """gs_upload_ply beyond one 64 Ki-vertex chunk (three chunks), with a property order of its own, a property the packer does not
    use (nx), a uchar property (value / 255, ply.ts:113-119: vertices of 93 bytes, so the floats of most vertices are unaligned)
    and SH degree 1 (zero-padded to the 16 coefficients the shader reads)."""
    import ctypes
    import gsplat
    from gsplat import _abi
    n, W, H = 150_000, 320, 192
    rec = scene(n).copy()
    rec[:, 16 + 4 * 4:16 + 4 * 16] = 0.0  # degree 1: coefficients 4..15 absent
    op_u8 = np.clip(np.round(127.5 + 20.0 * rec[:, 12]), 0, 255).astype(np.uint8)
    rec[:, 12] = (op_u8.astype(np.float64) / 255.0).astype(np.float32)  # the opacity logit travels as a uchar
    names = ["rot_0", "rot_1", "rot_2", "rot_3", "x", "y", "z", "nx", "scale_0", "scale_1", "scale_2", "f_dc_0", "f_dc_1", "f_dc_2"] + \
            ["f_rest_%d" % i for i in range(9)]
    cols = {"rot_0": rec[:, 8], "rot_1": rec[:, 9], "rot_2": rec[:, 10], "rot_3": rec[:, 11], "x": rec[:, 0], "y": rec[:, 1], "z": rec[:, 2],
            "nx": 0 * rec[:, 0], "scale_0": rec[:, 4], "scale_1": rec[:, 5], "scale_2": rec[:, 6]}
    for c in range(3):
        cols["f_dc_%d" % c] = rec[:, 16 + c]
        for i in range(3):
            cols["f_rest_%d" % (c * 3 + i)] = rec[:, 16 + 4 * (i + 1) + c]
    dt = np.dtype([(k, "<f4") for k in names] + [("opacity", "u1")], align=False)  # 93-byte vertices: floats at odd offsets
    arr = np.zeros(n, dtype=dt)
    for k in names:
        arr[k] = cols[k]
    arr["opacity"] = op_u8
    path = str(tmp_path / "odd.ply")
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join("property float %s\n" % k for k in names) +
                 "property uchar opacity\nend_header\n").encode())
        f.write(arr.tobytes())
    u = _uniforms(W, H, step=21)
    ref = oracle.render(rec, u, W, H, 16)
    pg = gsplat.PackedGaussians.from_ply(path)  # gs_ply_load: the packed 320-byte records
    assert pg.numGaussians == n and pg.sphericalHarmonicsDegree == 1
    np.testing.assert_array_equal(np.asarray(pg.gaussiansBuffer).view(np.uint32), rec.view(np.uint32))
    r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, flags=_abi.GS_FLAG_EXACT_BLEND | _abi.GS_FLAG_F32_TAP)
    r.render_uniforms(u)
    r.wait()
    want = r.read_buffer(12)
    cnt = ctypes.c_uint64()
    _abi.check(_abi.load().gs_upload_ply(r._ctx, path.encode(), ctypes.byref(cnt)))
    assert cnt.value == n
    r.render_uniforms(u, debug=True)
    r.wait()
    np.testing.assert_array_equal(r.read_buffer(12), want)
    _check_stages(r, ref, exact_image=True)
    r.destroy()
