"""Shared checks of the GPU parity tests: one frame of the HIP path (through the C ABI) against the CPU oracle.

Contract (include/gsplat/gs_abi.h):
  * gs_render_debug keeps the reference's binning (the 3-sigma rect of process_gaussians.wgsl:74-86): every tap is
    bit-equal to the oracle.
  * gs_render (product path) may use TIGHT binning: an instance (gaussian, tile) is dropped only if no pixel of the tile
    can pass `power <= 0 && alpha >= 1/255` (compute_tiles.wgsl:57-63), so the image does not change by one bit.  Its
    sorted (key, value) list is then an ordered SUBSET of the reference's: checked here by removing from the oracle's
    list exactly the instances the GPU dropped and proving (oracle.instance_masks) that each of them is non-contributing,
    and that the per-instance 8x8-block masks the blend uses cover every contributing block.
"""
import numpy as np


def make_renderer(splats, W, H, ts=16, flags=0, cols=None, **kw):
    import gsplat
    from gsplat import _abi
    pg = splats if hasattr(splats, "numGaussians") else gsplat.PackedGaussians(splats)
    return gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, ts, flags=flags | _abi.GS_FLAG_F32_TAP, cols=cols, **kw)


def orbit_uniforms(W, H, step=3):
    from gsplat import synth
    return synth.orbit_camera(step, W, H).uniforms(W, H)


def _composite(keys, values):
    return (keys.astype(np.uint64) << np.uint64(32)) | values.astype(np.uint64)


def check_image(r, ref, exact_image, max_ill=0.005, report=None):
    from gsplat import _abi
    img = r.read_rgba8()
    f32 = r.read_buffer(_abi.GS_BUF_RGB_F32, np.float32).reshape(img.shape[0], img.shape[1], 3)
    x0, w = r.slab_x0, r.slab_width
    ref8 = ref["rgba8"][:, x0:x0 + w]
    reff = ref["rgbf"][:, x0:x0 + w]
    if exact_image:
        np.testing.assert_array_equal(f32.view(np.uint32), reff.view(np.uint32))
        np.testing.assert_array_equal(img, ref8)
        return
    ill = ref["illcond"][:, x0:x0 + w].astype(bool)
    err = np.abs(f32 - reff).max(axis=2)
    d8 = np.abs(img.astype(np.int32) - ref8.astype(np.int32)).max(axis=2)
    if report is not None:
        # flagged = the oracle found a keep/skip decision of that pixel within rounding distance of its threshold (the
        # reference has these discontinuities itself); what matters to a viewer is how many pixels REALLY moved
        report.update({"flagged_fraction": float(ill.mean()), "max_err_unflagged": float(err[~ill].max(initial=0.0)),
                       "max_err_flagged": float(err[ill].max(initial=0.0)),
                       "fraction_over_1e-4": float((err > 1e-4).mean()), "fraction_over_1e-3": float((err > 1e-3).mean()),
                       "pixels_over_1e-4": int((err > 1e-4).sum()), "max_err": float(err.max(initial=0.0)), "pixels": int(err.size),
                       "err_p50": float(np.percentile(err, 50)), "err_p99": float(np.percentile(err, 99)),
                       "rgba8_fraction_differing": float((d8 > 0).mean()), "rgba8_fraction_over_1_lsb": float((d8 > 1).mean()),
                       "rgba8_max_lsb": int(d8.max(initial=0))})
    assert err[~ill].max(initial=0.0) <= 1e-4, "fused blend deviates by %g" % err[~ill].max()
    assert ill.mean() <= max_ill, "too many ill-conditioned pixels: %g" % ill.mean()
    assert err.max(initial=0.0) <= 0.05
    assert d8[~ill].max(initial=0) <= 1


def check_product_lists(r, ref, oracle, W, H, ts, report=None):
    """Sorted lists / ranges / counts of a gs_render frame: equal to the oracle's when the frame used the reference's rect
    binning, an ordered, provably harmless subset when it used tight binning."""
    from gsplat import _abi
    st = r.stats()
    keys = r.read_buffer(_abi.GS_BUF_KEYS)
    vals = r.read_buffer(_abi.GS_BUF_VALUES)
    rng = r.read_buffer(_abi.GS_BUF_RANGES)
    assert st["num_intersections"] == keys.size == vals.size
    if not st["tight_binning"]:
        assert st["num_intersections"] == ref["num_intersections"]
        assert st["num_visible"] == int((ref["tile_counts"] > 0).sum())
        np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_TILE_COUNTS), ref["tile_counts"])
        np.testing.assert_array_equal(keys, ref["sorted_keys"])
        np.testing.assert_array_equal(vals, ref["sorted_values"])
        np.testing.assert_array_equal(rng, ref["ranges"])
        return
    ntx, nty = oracle.num_tiles(W, H, ts)
    T = ntx * nty
    gc, rc = _composite(keys, vals), _composite(ref["sorted_keys"], ref["sorted_values"])
    # both lists are ordered by (key, gaussian index): the reference's order inside a tile IS (bucket, index)
    assert (np.diff(rc.view(np.int64)) >= 0).all() if rc.size else True
    assert (gc[1:] >= gc[:-1]).all(), "the product list is not in the reference's order"
    # multiset inclusion (a pair can occur twice: the column-aliasing quirk)
    first = np.searchsorted(gc, gc, "left")
    pos = np.searchsorted(rc, gc, "left") + (np.arange(gc.size) - first)
    assert gc.size == 0 or (pos.max() < rc.size and (rc[pos] == gc).all()), "the product list holds an instance the reference does not"
    kept = np.zeros(rc.size, dtype=bool)
    kept[pos] = True
    ref_masks = oracle.instance_masks(ref["gdata"], ref["sorted_keys"], ref["sorted_values"], W, H, ts)
    bad = ref_masks[~kept] != 0
    assert not bad.any(), "%d dropped instances would have contributed (first: key %d value %d)" % (
        int(bad.sum()), int(ref["sorted_keys"][~kept][bad][0]), int(ref["sorted_values"][~kept][bad][0]))
    gmask = r.read_buffer(_abi.GS_BUF_BLOCK_MASKS)
    assert gmask.size == keys.size
    miss = ref_masks[kept] & ~gmask
    assert not miss.any(), "%d kept instances lack a contributing 8x8 block in their mask" % int((miss != 0).sum())
    np.testing.assert_array_equal(rng, oracle.ranges(keys, T))
    counts = r.read_buffer(_abi.GS_BUF_TILE_COUNTS)
    np.testing.assert_array_equal(counts, np.bincount(vals, minlength=counts.size).astype(np.uint32))
    assert st["num_visible"] == int((counts > 0).sum())
    gd = r.read_buffer(_abi.GS_BUF_GAUSSIAN_DATA).reshape(-1, 16)
    vis = counts > 0
    a, b = gd[vis], ref["gdata"][vis]
    same = (a == b) | (np.isnan(a.view(np.float32)) & np.isnan(b.view(np.float32)))  # a NaN's sign/payload is not defined
    same[:, 12:] = a[:, 12:] == b[:, 12:]  # the rect words are integers
    assert same.all(), "GaussianData differs in %d words" % int((~same).sum())
    if report is not None:
        report.update(reference_instances=int(rc.size), product_instances=int(gc.size),
                      kept_fraction=float(gc.size) / max(int(rc.size), 1),
                      exact_contributing_fraction=float((ref_masks != 0).mean()) if rc.size else 0.0)


def check_stages(r, ref, exact_image, debug=True, oracle=None, W=None, H=None, ts=16, report=None):
    """debug=True: the frame came from gs_render_debug (the reference's gaussian-index emission order and rect binning,
    every tap valid and bit-equal); debug=False: from gs_render (see check_product_lists)."""
    from gsplat import _abi
    if debug:
        st = r.stats()
        assert st["tight_binning"] == 0
        assert st["num_intersections"] == ref["num_intersections"]
        assert st["num_visible"] == int((ref["tile_counts"] > 0).sum())
        np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_TILE_COUNTS), ref["tile_counts"])
        np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_TILE_OFFSETS), ref["offsets"])
        gd = r.read_buffer(_abi.GS_BUF_GAUSSIAN_DATA).reshape(-1, 16)
        np.testing.assert_array_equal(gd, ref["gdata"])  # floats compared as bits
        np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_KEYS_UNSORTED), ref["keys"])
        np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_VALUES_UNSORTED), ref["values"])
        np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_KEYS), ref["sorted_keys"])
        np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_VALUES), ref["sorted_values"])
        np.testing.assert_array_equal(r.read_buffer(_abi.GS_BUF_RANGES), ref["ranges"])
    else:
        if oracle is None:
            from oracle import gs_oracle as oracle
        if W is None:
            W, H = r.canvas.width, r.canvas.height
        if r.slab_width != W:
            # a tile-column slab: the oracle's lists for that slab are in ref already (rendered with cols=)
            pass
        check_product_lists(r, ref, oracle, W, H, r.tileSize, report)
    check_image(r, ref, exact_image, report=report)
