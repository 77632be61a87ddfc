"""Independent numpy restatement of the reference pipeline (second opinion on gs_oracle.c).

TEST INFRASTRUCTURE ONLY.  Written array-at-a-time from the WGSL sources, not from the C oracle,
using the same canonical float semantics (oracle/gs_oracle.c header): float32 numpy ops round each
operation individually; fma is emulated exactly (float64 product of two float32 is exact; the one
possible double-rounding case is detected and resolved with rationals).
"""
from fractions import Fraction

import numpy as np

F = np.float32


def fma(a, b, c):
    a, b, c = np.broadcast_arrays(np.asarray(a, F), np.asarray(b, F), np.asarray(c, F))
    d = a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)
    out = d.astype(F)
    # double rounding can only bite when d sits exactly on a float32 midpoint
    bits = d.view(np.uint64) & np.uint64((1 << 29) - 1)
    tie = np.flatnonzero((bits == np.uint64(1 << 28)).ravel() & np.isfinite(d).ravel())
    if tie.size:
        o = out.copy().ravel()
        for i in tie:
            exact = Fraction(float(a.ravel()[i])) * Fraction(float(b.ravel()[i])) + Fraction(float(c.ravel()[i]))
            lo = np.nextafter(o[i], F(-np.inf))
            hi = np.nextafter(o[i], F(np.inf))
            best = min((lo, o[i], hi), key=lambda v: (abs(Fraction(float(v)) - exact), int(np.float32(v).view(np.uint32)) & 1))
            o[i] = best
        out = o.reshape(out.shape)
    return out


def expf(x):
    """Canonical exp (gs_oracle.c: gso_expf) on float32 arrays."""
    x = np.asarray(x, F)
    with np.errstate(all="ignore"):
        nf = np.rint(x * F(1.44269502162933349609375))
        nf = np.where(np.isfinite(nf), nf, F(0))
        r = fma(-nf, F(0.693145751953125), x)
        r = fma(-nf, F(1.42860677465796470642e-06), r)
        p = np.full_like(r, F(1.9875691500e-4))
        for c in (1.3981999507e-3, 8.3334519073e-3, 4.1665795894e-2, 1.6666665459e-1, 5.0000001201e-1):
            p = fma(p, r, F(c))
        y = fma(p, r * r, r) + F(1.0)
        n = nf.astype(np.int32)
        a = n >> 1
        b = n - a
        sa = ((a + 127).astype(np.uint32) << np.uint32(23)).view(F)
        sb = ((b + 127).astype(np.uint32) << np.uint32(23)).view(F)
        res = (y * sa) * sb
    res = np.where(x > F(88.72283935546875), F(np.inf), res)
    res = np.where(x < F(-103.97208404541015625), F(0), res)
    return np.where(np.isnan(x), x, res).astype(F)


def wmax(a, b):
    return np.where(a < b, b, a)


def wmin(a, b):
    return np.where(b < a, b, a)


def f2i(x):
    x = np.asarray(x, F)
    with np.errstate(invalid="ignore"):
        v = np.trunc(np.clip(np.nan_to_num(x, nan=0.0, posinf=3e9, neginf=-3e9).astype(np.float64), -2147483648.0, 2147483647.0))
    return v.astype(np.int64)


def tdiv(a, b):
    """i32 division truncating toward zero"""
    return (np.abs(a) // b) * np.sign(a)


def mat4v(m, x, y, z):
    """column-major mat4 times (x,y,z,1), ((c0*x + c1*y) + c2*z) + c3*1"""
    return [((m[0 + r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * F(1.0) for r in range(4)]


def m3mul(A, B):
    """A, B: [c][r] lists of arrays; (A*B)[c][r] = sum_k A[k][r]*B[c][k], k ascending"""
    return [[(A[0][r] * B[c][0] + A[1][r] * B[c][1]) + A[2][r] * B[c][2] for r in range(3)] for c in range(3)]


def m3t(A):
    return [[A[r][c] for r in range(3)] for c in range(3)]


SH_C0 = F(0.28209479177387814)
SH_C1 = F(0.4886025119029199)
SH_C2 = [F(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)]
SH_C3 = [F(v) for v in (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                        -0.4570457994644658, 1.445305721320277, -0.5900435899266435)]


def preprocess(splats, uni, W, H, ts=16):
    """process_gaussians.wgsl:35-106 -> dict of per-gaussian arrays (culled entries zero)."""
    s = np.asarray(splats, F).reshape(-1, 80)
    u = np.asarray(uni, F)
    V, PV, cam = u[0:16], u[16:32], u[32:35]
    tan_x, tan_y, fx, fy, mod = u[35], u[36], u[37], u[38], u[39]
    px, py, pz = s[:, 0], s[:, 1], s[:, 2]
    with np.errstate(all="ignore"):
        hom = mat4v(PV, px, py, pz)
        pw = F(1.0) / (hom[3] + F(0.0000001))
        ndx, ndy = hom[0] * pw, hom[1] * pw
        view = mat4v(V, px, py, pz)
        culled = (view[2] <= F(0.2)) | (ndx <= F(-1.1)) | (ndx >= F(1.1)) | (ndy <= F(-1.1)) | (ndy >= F(1.1))
        uvx, uvy = ndx * F(0.5) + F(0.5), ndy * F(0.5) + F(0.5)
        # cov3d :127-162
        sc = [expf(s[:, 4 + k]) * mod for k in range(3)]
        q = [s[:, 8 + k] for k in range(4)]
        ln = np.sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3])
        r, x, y, z = (q[k] / ln for k in range(4))
        one, two = F(1.0), F(2.0)
        R = [[one - two * (y * y + z * z), two * (x * y - r * z), two * (x * z + r * y)],
             [two * (x * y + r * z), one - two * (x * x + z * z), two * (y * z - r * x)],
             [two * (x * z - r * y), two * (y * z + r * x), one - two * (x * x + y * y)]]
        M = [[sc[rr] * R[c][rr] for rr in range(3)] for c in range(3)]
        Sig = m3mul(m3t(M), M)
        # cov2d :165-218
        t = list(view)
        limx, limy = F(1.3) * tan_x, F(1.3) * tan_y
        t[0] = wmin(limx, wmax(-limx, t[0] / t[2])) * t[2]
        t[1] = wmin(limy, wmax(-limy, t[1] / t[2])) * t[2]
        zero = np.zeros_like(px)
        J = [[fx / t[2], zero, -(fx * t[0]) / (t[2] * t[2])], [zero, fy / t[2], -(fy * t[1]) / (t[2] * t[2])],
             [zero, zero, zero]]
        # W[c][r] = V[r][c]: transpose of the upper-left 3x3 of the view matrix (:195-199)
        Wm = [[V[0 + 0] + zero, V[4 + 0] + zero, V[8 + 0] + zero], [V[0 + 1] + zero, V[4 + 1] + zero, V[8 + 1] + zero],
              [V[0 + 2] + zero, V[4 + 2] + zero, V[8 + 2] + zero]]
        T = m3mul(Wm, J)
        Vrk = [[Sig[0][0], Sig[0][1], Sig[0][2]], [Sig[0][1], Sig[1][1], Sig[1][2]], [Sig[0][2], Sig[1][2], Sig[2][2]]]
        cov = m3mul(m3mul(m3t(T), m3t(Vrk)), T)
        a, b, c = cov[0][0] + F(0.3), cov[0][1], cov[1][1] + F(0.3)
        det = a * c - b * b
        culled = culled | (det == 0)
        inv = F(1.0) / det
        conic = [c * inv, (-b) * inv, a * inv]
        mid = F(0.5) * (a + c)
        sq = np.sqrt(wmax(F(0.1), mid * mid - det))
        radius = np.ceil(F(3.0) * np.sqrt(wmax(mid + sq, mid - sq)))
        ntx = int(np.ceil(F(W) / F(ts)))
        nty = int(np.ceil(F(H) / F(ts)))
        ppx, ppy = uvx * F(W), uvy * F(H)
        rminx = np.minimum(ntx, np.maximum(0, tdiv(f2i(ppx - radius), ts)))
        rminy = np.minimum(nty, np.maximum(0, tdiv(f2i(ppy - radius), ts)))
        rmaxx = np.minimum(ntx, np.maximum(0, tdiv(f2i(ppx + radius), ts))) + 1
        rmaxy = np.minimum(nty, np.maximum(0, tdiv(f2i(ppy + radius), ts))) + 1
        # colour :240-280
        d = [px - cam[0], py - cam[1], pz - cam[2]]
        dl = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        x, y, z = d[0] / dl, d[1] / dl, d[2] / dl
        xx, yy, zz, xy, xz, yz = x * x, y * y, z * z, x * y, x * z, y * z
        k = [None] * 16
        k[4], k[5], k[6] = SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * ((F(2) * zz - xx) - yy)
        k[7], k[8] = SH_C2[3] * xz, SH_C2[4] * (xx - yy)
        k[9] = (SH_C3[0] * y) * (F(3) * xx - yy)
        k[10] = (SH_C3[1] * xy) * z
        k[11] = (SH_C3[2] * y) * ((F(4) * zz - xx) - yy)
        k[12] = (SH_C3[3] * z) * ((F(2) * zz - F(3) * xx) - F(3) * yy)
        k[13] = (SH_C3[4] * x) * ((F(4) * zz - xx) - yy)
        k[14] = (SH_C3[5] * z) * (xx - yy)
        k[15] = (SH_C3[6] * x) * (xx - F(3) * yy)
        color = []
        for ch in range(3):
            sh = lambda i: s[:, 16 + 4 * i + ch]
            res = SH_C0 * sh(0)
            res = res + SH_C1 * (((-y) * sh(1) + z * sh(2)) - x * sh(3))
            for i in range(4, 16):
                res = res + k[i] * sh(i)
            color.append(wmax(res + F(0.5), F(0.0)))
        # sigmoid :282-294
        o = s[:, 12]
        zz_ = expf(o)
        cond = (o >= 0).astype(F)
        opacity = (cond * (F(1.0) / (F(1.0) + expf(-o)))) + ((F(1.0) - cond) * (zz_ / (F(1.0) + zz_)))
    vis = ~culled
    count = np.where(vis, (rmaxy - rminy) * (rmaxx - rminx), 0).astype(np.uint32)
    z32 = lambda a_: np.where(vis, a_, 0).astype(F)
    return dict(uv=np.stack([z32(uvx), z32(uvy)], 1), conic=np.stack([z32(v) for v in conic], 1), depth=z32(view[2]),
                color=np.stack([z32(v) for v in color], 1), opacity=z32(opacity),
                rect=np.where(vis[:, None], np.stack([rminx, rminy, rmaxx, rmaxy], 1), 0).astype(np.uint32),
                count=count)


def keys_values(pre, W, ts=16):
    """scan + write_tile_ids.wgsl:18-35, gaussian order, y outer, x inner."""
    ntx = int(np.ceil(F(W) / F(ts)))
    keys, vals = [], []
    bucket = wmin(F(50.0) * pre["depth"], F(999.0)).astype(np.uint32)
    for i in np.flatnonzero(pre["count"]):
        x0, y0, x1, y1 = (int(v) for v in pre["rect"][i])
        yy, xx = np.meshgrid(np.arange(y0, y1, dtype=np.uint32), np.arange(x0, x1, dtype=np.uint32), indexing="ij")
        keys.append(((yy * np.uint32(ntx) + xx) * np.uint32(1000) + bucket[i]).ravel())
        vals.append(np.full(yy.size, i, dtype=np.uint32))
    if not keys:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    return np.concatenate(keys), np.concatenate(vals)


def sort_kv(keys, vals):
    order = np.argsort(keys, kind="stable")
    return keys[order], vals[order]


def ranges(sorted_keys, T):
    return np.searchsorted(sorted_keys // np.uint32(1000), np.arange(T, dtype=np.uint32), side="right").astype(np.uint32)


def blend(pre, svals, rng, W, H, ts=16):
    """compute_tiles.wgsl:30-75, one tile at a time, all pixels of the tile as a vector."""
    ntx = int(np.ceil(F(W) / F(ts)))
    nty = int(np.ceil(F(H) / F(ts)))
    out = np.zeros((H, W, 3), dtype=F)
    c255 = F(1.0 / 255.0)
    for ty in range(nty):
        for tx in range(ntx):
            tile = tx + ty * ntx
            start = int(rng[tile - 1]) if tile > 0 else 0
            end = int(rng[tile])
            gy, gx = np.meshgrid(np.arange(ty * ts, min(H, ty * ts + ts)), np.arange(tx * ts, min(W, tx * ts + ts)), indexing="ij")
            pxf, pyf = gx.astype(F), gy.astype(F)
            acc = [np.zeros_like(pxf) for _ in range(3)]
            T = np.ones_like(pxf)
            for j in range(start, end):
                g = int(svals[j])
                dx = pre["uv"][g, 0] * F(W) - pxf
                dy = pre["uv"][g, 1] * F(H) - pyf
                cx, cy, cz = pre["conic"][g]
                power = F(-0.5) * (cx * dx * dx + cz * dy * dy) - cy * dx * dy
                alpha = wmin(F(0.99), pre["opacity"][g] * expf(power))
                test = T * (F(1.0) - alpha)
                cond = ((power <= 0) & (alpha >= c255) & (test >= F(0.0001))).astype(F)
                for ch in range(3):
                    acc[ch] = acc[ch] + cond * pre["color"][g, ch] * alpha * T
                T = cond * test + (F(1.0) - cond) * T
            for ch in range(3):
                out[gy, gx, ch] = acc[ch]
    return out


def to_rgba8(rgbf):
    v = np.clip(np.nan_to_num(rgbf, nan=0.0), 0.0, 1.0).astype(F)
    rgb = np.floor(v * F(255.0) + F(0.5)).astype(np.uint8)
    return np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
