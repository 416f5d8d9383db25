"""TEST INFRASTRUCTURE — a SECOND, independent restatement of the reference's per-sample arithmetic, vectorised in
numpy, written from the reference's sources rather than from the C++ oracle.  The reference holds exact known answers
only for GaussNewton::step, the kd-tree, exp / compose / transform and project; for the ImageIcp pixel loop, the
normals and the bilateral grid it has smoke thresholds.  Two restatements that were written separately and agree
bit for bit (per-sample f32 values) are the strongest pin available without a Rust toolchain: a misreading would
have to be made twice, identically.  tests/test_oracle_kat.py compares this module with the C++ oracle.

  image_icp_terms        src/icp/image_icp.rs:101-139, cost_function.rs:5-57, camera.rs:64-89, intensity_map.rs:150-210
  compute_normals        src/range_image/structure.rs:184-262
  bilateral_filter_u16   src/bilateral/grid.rs:32-162, edge_aware_filter.rs:57-135"""
import numpy as np

F = np.float32


def _dot(a, b):  # nalgebra: (a0 b0 + a1 b1) + a2 b2
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def _rotate(q_ijkw, v):  # UnitQuaternion * Vector3: t = 2 (q_v x v); (t w + q_v x t) + v
    qv = np.broadcast_to(np.asarray(q_ijkw[:3], F), v.shape)
    t = _cross(qv, v) * F(2.0)
    return (t * F(q_ijkw[3]) + _cross(qv, t)) + v


def _as_usize(x):  # Rust `f32 as usize`: NaN and negatives -> 0, truncation toward zero
    return np.where(np.isnan(x) | (x <= 0), 0, np.minimum(x, F(2 ** 31))).astype(np.int64)


def _bilinear(m, u, v):  # IntensityMap::bilinear (intensity_map.rs:150-169)
    ui, vi = _as_usize(u), _as_usize(v)
    uf, vf = u - ui.astype(F), v - vi.astype(F)
    v00, v10, v01, v11 = m[vi, ui], m[vi, ui + 1], m[vi + 1, ui], m[vi + 1, ui + 1]
    u0 = v00 * (F(1) - uf) + v10 * uf
    u1 = v01 * (F(1) - uf) + v11 * uf
    return u0 * (F(1) - vf) + u1 * vf


def image_icp_terms(prm, tgt, src, t_xyz, q_ijkw):
    """One pass of the pixel loop from the transform (t, q): returns per accepted pixel the geometric (r, J) and, where
    the colour gate passes, the colour (r, J), as f32 arrays (rows = pixels in scan order)."""
    h, w = tgt.mask.shape
    pts = src.points.reshape(-1, 3)
    live = src.mask.reshape(-1) != 0                                          # :102
    idx = np.nonzero(live)[0]
    p = _rotate(q_ijkw, pts[idx]) + np.asarray(t_xyz, F)                      # :106
    fx, fy, cx, cy = F(tgt.fx), F(tgt.fy), F(tgt.cx), F(tgt.cy)
    with np.errstate(all="ignore"):
        u = p[:, 0] * fx / p[:, 2] + cx                                       # camera.rs:64-70
        v = p[:, 1] * fy / p[:, 2] + cy
        ui = np.where(np.isnan(u + F(0.5)), 0, np.clip(np.trunc(u + F(0.5)), -2 ** 31, 2 ** 31 - 1)).astype(np.int64)  # :108
        vi = np.where(np.isnan(v + F(0.5)), 0, np.clip(np.trunc(v + F(0.5)), -2 ** 31, 2 ** 31 - 1)).astype(np.int64)
    ok = (ui >= 0) & (ui < w) & (vi >= 0) & (vi < h)                          # get_point: `as usize` of a negative is huge
    ok[ok] &= tgt.mask[vi[ok], ui[ok]] == 1                                   # structure.rs:176
    idx, p, u, v, ui, vi = idx[ok], p[ok], u[ok], v[ok], ui[ok], vi[ok]
    q = tgt.points[vi, ui]
    d = q - p
    ok = ~(_dot(d, d) > F(prm.max_distance) * F(prm.max_distance))           # :114
    n = tgt.normals[vi, ui]
    with np.errstate(invalid="ignore"):
        ang = np.abs(np.arccos(_dot(p, n)))                                   # extra_math.rs:13-15 on the POINT p (NaN passes)
    ok &= ~(ang >= F(prm.max_normal_angle))                                   # :118-123
    idx, p, u, v, q, n = idx[ok], p[ok], u[ok], v[ok], q[ok], n[ok]
    rg = _dot(q - p, n)                                                       # cost_function.rs:33-41
    Jg = np.concatenate([n, _cross(p, n)], 1)
    m = tgt.intensity_map
    Hh = F(0.005)
    val = _bilinear(m, u, v)                                                  # intensity_map.rs:184-210
    du = (_bilinear(m, u + Hh, v) - val) * (F(1.0) / Hh)
    dv = (_bilinear(m, u, v + Hh) - val) * (F(1.0) / Hh)
    sc = src.intensities.reshape(-1)[idx].astype(F) * F(0.003921569)          # :131
    z = p[:, 2]
    zz = z * z
    dfx, dcx, dfy, dcy = fx / z, -p[:, 0] * fx / zz, fy / z, -p[:, 1] * fy / zz   # camera.rs:82-89
    g = np.stack([du * dfx, dv * dfy, du * dcx + dv * dcy], 1)
    rc = sc - val
    Jc = np.concatenate([g, _cross(p, g)], 1)
    col = rc * rc <= F(prm.max_color_distance) * F(prm.max_color_distance)   # :136
    return rg.astype(F), Jg.astype(F), rc[col].astype(F), Jc[col].astype(F)


def gn_sums(r, J):
    """GaussNewton::step summed in f64 over all samples: (H upper triangle 21, g 6, sum r^2, count)."""
    J64, r64 = J.astype(np.float64), r.astype(np.float64)
    H = J64.T @ J64
    return H, J64.T @ r64, float((r64 * r64).sum()), len(r)


def compute_normals(points, mask):
    """RangeImage::compute_normals: every pixel, centre mask not checked, masked / out-of-range neighbours = 0."""
    h, w = mask.shape
    P = np.where((mask == 1)[..., None], points, F(0)).astype(F)
    pad = np.zeros((h + 2, w + 2, 3), F)
    pad[1:-1, 1:-1] = P
    c = points.astype(F)
    left, right, top, bottom = pad[1:-1, :-2], pad[1:-1, 2:], pad[:-2, 1:-1], pad[2:, 1:-1]

    def pick(a, b):  # a = left / bottom, b = right / top
        da, db = _dot(a - c, a - c), _dot(b - c, b - c)
        with np.errstate(all="ignore"):
            ratio = da / db
        both = (ratio < F(4.0)) & (ratio > F(1.0) / F(4.0))
        return np.where(both[..., None], b - a, np.where((da < db)[..., None], c - a, b - c))

    nrm = _cross(pick(left, right), pick(bottom, top))
    mag = np.sqrt(_dot(nrm, nrm))
    with np.errstate(all="ignore"):
        unit = nrm / mag[..., None]
    return np.where((mag > F(1e-6))[..., None], unit, F(0)).astype(F)


def bilateral_filter_u16(img, sigma_space=4.50000000225, sigma_color=29.9999880000072):
    """BilateralFilter::<u16>::filter: splat, 3 axes x 2 passes on ping-pong buffers through FLAT offsets (so that the
    reference's un-offset channel loop — channel 0's "previous" aliases the previous column's last channel — is
    reproduced literally), normalise, trilinear slice, truncating cast.  Returns (image, (gh, gw, gd))."""
    img = np.asarray(img, np.uint16)
    h, w = img.shape
    cmin, cmax = int(img.min()), int(img.max())
    pad = 2
    gh, gw = int((h - 1) / sigma_space) + 1 + 2 * pad, int((w - 1) / sigma_space) + 1 + 2 * pad
    gd = int((cmax - cmin) / sigma_color) + 1 + 2 * pad
    data = np.zeros(gh * gw * gd * 2, np.float64)
    rows, cols = np.nonzero(img > 0)
    vals = img[rows, cols].astype(np.float64)
    gr = np.floor(rows / sigma_space + 0.5).astype(np.int64) + pad
    gc = np.floor(cols / sigma_space + 0.5).astype(np.int64) + pad
    gz = np.floor((vals - cmin) / sigma_color + 0.5).astype(np.int64) + pad
    cell = ((gr * gw + gc) * gd + gz) * 2
    np.add.at(data, cell, vals)
    np.add.at(data, cell + 1, 1.0)
    # convolution (edge_aware_filter.rs:57-115)
    rs, cs, zs = gw * gd * 2, gd * 2, 2
    r_i, c_i, z_i = np.meshgrid(np.arange(1, gh - 1), np.arange(1, gw - 1), np.arange(0, gd - 1), indexing="ij")
    at = (r_i * rs + c_i * cs + z_i * zs).reshape(-1)       # value slots written by every pass
    buf = np.zeros_like(data)
    for off in (rs, cs, zs):
        for _ in range(2):
            data, buf = buf, data                            # swap(&mut data_ptr, &mut buffer_ptr)
            for k in (0, 1):                                 # value, weight
                a = at + k
                data[a] = (buf[a - off] + buf[a + off] + 2.0 * buf[a]) * 0.25
    grid = data.reshape(gh, gw, gd, 2)                       # six swaps: back in the grid's own buffer
    with np.errstate(all="ignore"):
        value = np.where(grid[..., 1] > 0.0, grid[..., 0] / grid[..., 1], grid[..., 0])   # normalize (grid.rs:90-104)
    rr, cc = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    row = rr * (1.0 / sigma_space) + pad
    col = cc * (1.0 / sigma_space) + pad
    ch = (img.astype(np.float64) - cmin) * (1.0 / sigma_color) + pad

    def split(x, hi):
        i0 = np.minimum(x.astype(np.int64), hi)
        i1 = np.minimum((x + 1.0).astype(np.int64), hi)
        return i0, i1, x - i0

    y, yy, ya = split(row, gh - 1)
    x, xx, xa = split(col, gw - 1)
    z, zz, za = split(ch, gd - 1)
    out = ((1.0 - ya) * (1.0 - xa) * (1.0 - za) * value[y, x, z] + (1.0 - ya) * xa * (1.0 - za) * value[y, xx, z]
           + ya * (1.0 - xa) * (1.0 - za) * value[yy, x, z] + ya * xa * (1.0 - za) * value[yy, xx, z]
           + (1.0 - ya) * (1.0 - xa) * za * value[y, x, zz] + (1.0 - ya) * xa * za * value[y, xx, zz]
           + ya * (1.0 - xa) * za * value[yy, x, zz] + ya * xa * za * value[yy, xx, zz])
    assert np.all((out > -1.0) & (out < 65536.0)), "num::cast would panic"
    return out.astype(np.uint16), (gh, gw, gd)
