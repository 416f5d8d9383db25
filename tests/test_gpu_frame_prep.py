"""GPU parity: compute_normals and the bilateral filter are bit-exact against the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from align3d_amd import A3dError, BilateralFilter
from data_util import SlamTbSample, uniform01
from gpu_util import oracle_frame, to_range_image

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("sample,frame", [("sample1", 0), ("sample2", 1)])
def test_compute_normals_bit_exact(ctx, sample, frame):
    fr = oracle_frame(sample, frame)
    ri = to_range_image(fr)
    ri.normals = None
    ri.compute_normals(ctx)
    assert np.array_equal(_bits(ri.normals), _bits(fr.normals))
    # device-resident form gives the same array
    dev = to_range_image(fr).device(ctx)
    dev.compute_normals()
    assert np.array_equal(_bits(dev.download_normals()), _bits(fr.normals))


def test_compute_normals_batch_bit_exact(ctx):
    """a3d_range_image_compute_normals_batch: many resident images in one launch (more than one 64-image launch here),
    every image bit for bit the oracle's normals (structure.rs:184-262) — real frames, and a ragged size that is neither
    a multiple of the 64 x 16 tile nor of a wave."""
    from align3d_amd import CameraIntrinsics, RangeImage, compute_normals_batch

    frames = [oracle_frame("sample1", k) for k in (0, 1, 4, 5)] + [oracle_frame("sample2", k) for k in (0, 1)]
    devs, refs = [], []
    for rep in range(11):  # 66 images of 640 x 480
        for fr in frames:
            ri = RangeImage(fr.points, fr.mask, CameraIntrinsics(fr.fx, fr.fy, fr.cx, fr.cy, fr.w, fr.h))
            devs.append(ri.device(ctx))
            refs.append(fr.normals)
    compute_normals_batch(devs)
    for k in (0, 5, 63, 64, 65):
        assert np.array_equal(devs[k].download_normals().view(np.uint32), refs[k].view(np.uint32)), k
    for d in devs:
        d.free()
    rng = np.random.default_rng(5)
    small = []
    for k in range(5):
        pts = rng.normal(size=(37, 131, 3)).astype(np.float32)
        mask = (rng.random((37, 131)) > 0.2).astype(np.uint8) * rng.integers(1, 3, size=(37, 131)).astype(np.uint8)
        small.append((pts, mask))
    devs = [RangeImage(p, m, CameraIntrinsics(100, 100, 65, 18, 131, 37)).device(ctx) for p, m in small]
    compute_normals_batch(devs)
    for d, (p, m) in zip(devs, small):
        assert np.array_equal(d.download_normals().view(np.uint32), O.compute_normals(p, m).view(np.uint32))
    # images of different sizes in one call are refused
    from align3d_amd import A3dError
    other = RangeImage(frames[0].points, frames[0].mask, CameraIntrinsics(1, 1, 0, 0, 640, 480)).device(ctx)
    with pytest.raises(A3dError):
        compute_normals_batch([devs[0], other])


def test_compute_normals_ragged_and_empty_inputs(ctx):
    # sizes that are not multiples of the tile, all-invalid mask, NaN points, single row / column
    rng = np.random.default_rng(7)
    for (h, w) in [(1, 1), (1, 37), (45, 1), (33, 70), (17, 129)]:
        pts = rng.normal(size=(h, w, 3)).astype(np.float32)
        pts[..., 2] += 3
        mask = (rng.random((h, w)) > 0.3).astype(np.uint8)
        if h * w > 10:
            mask.flat[3] = 2  # mask value other than 0/1 is "invalid" for get_point (== 1 test)
            pts.flat[30] = np.nan
        ref = O.compute_normals(pts, mask)
        from align3d_amd import CameraIntrinsics, RangeImage

        ri = RangeImage(pts, mask, CameraIntrinsics(500, 500, w / 2, h / 2, w, h)).compute_normals(ctx)
        assert np.array_equal(_bits(ri.normals), _bits(ref)), (h, w)
    pts = np.zeros((8, 8, 3), np.float32)
    ri = RangeImage(pts, np.zeros((8, 8), np.uint8), CameraIntrinsics(1, 1, 0, 0, 8, 8)).compute_normals(ctx)
    assert not ri.normals.any()


@pytest.mark.parametrize("sample,frame", [("sample1", 0), ("sample1", 5)])
def test_bilateral_bit_exact_real_depth(ctx, sample, frame):
    depth, _ = SlamTbSample(sample).load(frame)
    st, ref, dims = O.bilateral(depth)
    assert st == 0
    f = BilateralFilter.default()
    out = f.filter(ctx, depth)
    assert f.last_grid_dims == dims
    assert np.array_equal(out, ref)


def test_bilateral_bit_exact_synthetic_and_edges(ctx):
    # ragged sizes, other sigmas, all-zero image, constant image, full u16 range
    cases = []
    u = uniform01(11, 61 * 47).reshape(61, 47)
    cases.append(((u * 3000).astype(np.uint16), 4.50000000225, 29.9999880000072))
    cases.append(((u * 65535).astype(np.uint16), 7.0, 900.0))
    cases.append((np.zeros((16, 16), np.uint16), 4.5, 30.0))
    cases.append((np.full((20, 33), 1234, np.uint16), 3.0, 10.0))
    holes = (u * 2000 + 500).astype(np.uint16)
    holes[10:20, 5:25] = 0
    cases.append((holes, 4.5, 30.0))
    for img, ss, sc in cases:
        st, ref, dims = O.bilateral(img, ss, sc)
        f = BilateralFilter.new(ss, sc)
        out = f.filter(ctx, img)
        assert st == 0 and f.last_grid_dims == dims
        assert np.array_equal(out, ref)


def test_bilateral_cast_never_overflows_on_the_extreme_u16_inputs(ctx):
    """`num::cast::cast(trilinear).unwrap()` (src/bilateral/grid.rs:129) panics on a value outside u16 -> A3D_CAST_OVERFLOW.
    With positive sigmas the trilinear value is a convex combination of weighted means of u16 inputs, so the status is
    unreachable (the argument is written out at its check in bilateral.hip); the inputs that come closest — 0 / 65535
    checkerboards and stripes, all-65535, colour sigmas from far below one grey level to far above the range — stay
    A3D_OK and bit-identical to the oracle, whose own cast check is the same.  Non-positive sigmas (where the reference
    divides by them and then casts garbage) are rejected up front."""
    yy, xx = np.mgrid[0:70, 0:90]
    board = (((yy + xx) & 1) * 65535).astype(np.uint16)
    stripes = ((xx // 7 & 1) * 65535).astype(np.uint16)
    full = np.full((40, 50), 65535, np.uint16)
    for img in (board, stripes, full):
        for ss, sc in ((4.5, 30.0), (1.0, 0.25), (2.0, 1e6), (9.0, 65535.0)):
            if sc < 1.0 and img.max() > 2000:
                continue  # (a 65535 / 0.25 = 262 k-channel grid: the reference allocates 70 x 90 x 262 k cells; not a case)
            st, ref, dims = O.bilateral(img, ss, sc)
            out = BilateralFilter.new(ss, sc).filter(ctx, img)
            assert st == 0 and np.array_equal(out, ref), (ss, sc)
    for ss, sc in ((0.0, 30.0), (4.5, 0.0), (-4.5, 30.0), (4.5, -30.0), (float("nan"), 30.0), (4.5, float("nan"))):
        with pytest.raises(A3dError) as e:
            BilateralFilter.new(ss, sc).filter(ctx, board)
        assert e.value.status == 1


def test_bilateral_bit_exact_on_the_reference_known_answer_image(ctx):
    """The image of the reference's own bilateral KAT (bloei.jpg -> luma16 / 13, src/unit_test/images.rs:28-40,
    grid 138 x 104 x 173 of src/bilateral/grid.rs:183-185; the oracle is near-pinned on it in test_oracle_kat.py):
    the device filter at the KAT's sigmas equals the oracle bit for bit on all 600 x 450 pixels."""
    from test_oracle_kat import bloei_luma16

    img = bloei_luma16()
    st, ref, dims = O.bilateral(img, 4.5, 30.0)
    f = BilateralFilter.new(4.5, 30.0)
    out = f.filter(ctx, img)
    assert st == 0 and dims == (138, 104, 173) and f.last_grid_dims == dims
    assert np.array_equal(out, ref)


def _assert_same_level(dev_level, ref):
    got = dev_level.download()
    assert got.mask.shape == ref.mask.shape
    assert np.array_equal(got.mask, ref.mask)
    assert np.array_equal(_bits(got.points), _bits(ref.points))
    assert np.array_equal(_bits(got.normals), _bits(ref.normals))
    assert np.array_equal(got.colors, ref.colors)
    assert np.array_equal(got.intensities, ref.intensities)
    assert np.array_equal(_bits(got.intensity_map), _bits(ref.intensity_map))
    k = got.intrinsics
    assert (k.fx, k.fy, k.cx, k.cy) == (ref.fx, ref.fy, ref.cx, ref.cy)


@pytest.mark.parametrize("sample,frame,bilateral", [("sample1", 0, True), ("sample1", 5, False), ("sample2", 1, True)])
def test_device_pyramid_builder_bit_exact(ctx, sample, frame, bilateral):
    """RangeImageBuilder::build on the device (bilateral, back-projection, normals, 2x2 nearest-to-mean
    resize of points and normals, RGB blur + halve, luma, intensity map with its border quirks) equals the
    oracle's pyramid bit for bit on every level."""
    from align3d_amd import CameraIntrinsics, RangeImageBuilder
    from gpu_util import oracle_pyramid

    s = SlamTbSample(sample)
    depth, rgb = s.load(frame)
    fx, fy, cx, cy = s.intrinsics(frame)
    ref = oracle_pyramid(sample, frame, levels=3, use_bilateral=bilateral)
    b = RangeImageBuilder(ctx)
    if bilateral:
        b = b.with_bilateral_filter(BilateralFilter.default())
    levels = b.build_device(CameraIntrinsics(fx, fy, cx, cy, 640, 480), depth, rgb, s.depth_scale(frame))
    assert [lv.shape for lv in levels] == [(480, 640), (240, 320), (120, 160)]
    for lv, r in zip(levels, ref):
        _assert_same_level(lv, r)


def test_device_pyramid_builder_options_and_errors(ctx):
    from align3d_amd import A3dError, CameraIntrinsics, RangeImageBuilder

    s = SlamTbSample("sample1")
    depth, rgb = s.load(1)
    cam = CameraIntrinsics(*s.intrinsics(1), 640, 480)
    one = RangeImageBuilder(ctx).pyramid_levels(1).with_normals(False).with_intensity(False).build_device(
        cam, depth, rgb, s.depth_scale(1))
    assert len(one) == 1
    with pytest.raises(A3dError) as e:
        one[0].download(normals=True, intensity=False)
    assert e.value.status == 2  # no normals were requested
    ri = one[0].download(normals=False, intensity=False)
    assert int(ri.mask.sum()) == int((depth > 0).sum())
    with pytest.raises(A3dError):
        RangeImageBuilder(ctx).pyramid_levels(12).build_device(cam, depth, rgb, 0.001)  # too many levels for 640x480


@pytest.mark.parametrize("w,h,sigma", [(150, 90, 1.0), (150, 90, 0.6), (262, 134, 2.0), (640, 480, 3.0), (66, 34, 1.7)])
def test_device_pyramid_builder_ragged_sizes_and_blur_sigmas(ctx, w, h, sigma):
    """The fused blur + halve kernel (even rows/columns only, vertical sums through LDS) against the oracle's full
    two-pass blur: odd sizes, tiles that end inside the image, tap counts 3..14 (sigma up to 3)."""
    from align3d_amd import A3dError, CameraIntrinsics, RangeImageBuilder

    rng = np.random.default_rng(w * 1000 + h)
    depth = rng.integers(400, 5000, size=(h, w), dtype=np.uint16)
    depth[rng.random((h, w)) < 0.1] = 0
    rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    rgb[: h // 3] = 255  # saturated block: the clamp + round-half-away path
    rgb[-(h // 4):, :, 1] = 0
    fx = fy = 0.85 * w
    cx, cy = w / 2.0, h / 2.0
    ref = O.build_pyramid(depth, rgb, fx, fy, cx, cy, 0.001, levels=3, use_bilateral=False, sigma=sigma)
    levels = RangeImageBuilder(ctx).blur_sigma(sigma).build_device(CameraIntrinsics(fx, fy, cx, cy, w, h), depth, rgb, 0.001)
    for lv, r in zip(levels, ref):
        _assert_same_level(lv, r)
    with pytest.raises(A3dError) as e:
        RangeImageBuilder(ctx).blur_sigma(3.5).build_device(CameraIntrinsics(fx, fy, cx, cy, w, h), depth, rgb, 0.001)
    assert e.value.status == 1


def test_bilateral_fused_and_unfused_paths_agree(ctx, diag_ctx, monkeypatch):
    """The packed-integer splat + LDS-tiled six-pass blur and the original pass-per-launch path are the same
    function (and both are the oracle's): grids smaller than one tile, tiles cut by every grid face, deep grids."""
    rng = np.random.default_rng(11)
    cases = [(rng.integers(0, 65536, size=(97, 131)).astype(np.uint16), 3.0, 700.0),
             (rng.integers(300, 9000, size=(480, 640)).astype(np.uint16), 4.5, 30.0),
             (rng.integers(0, 3, size=(50, 60)).astype(np.uint16), 1.5, 0.7),
             (np.full((13, 17), 40000, np.uint16), 20.0, 5.0),
             # 32 772 channels: 2 731 channel tiles, more tile marks than the splat's LDS window holds (direct stores),
             # far more than four channels under a footprint (slot overflow), a grid that outgrows the first capacity
             (rng.integers(0, 65536, size=(50, 60)).astype(np.uint16), 3.0, 2.0)]
    for img, ss, sc in cases:
        st, ref, dims = O.bilateral(img, ss, sc)
        outs = {}
        for mode in ("fused", "sync", "unfused"):  # one enqueue / min-max via the host / pass per launch
            monkeypatch.setenv("A3D_BILATERAL", mode)  # (read by the diagnostics build only)
            f = BilateralFilter.new(ss, sc)
            outs[mode] = f.filter(diag_ctx, img)
            assert f.last_grid_dims == dims
        f = BilateralFilter.new(ss, sc)
        outs["product"] = f.filter(ctx, img)
        assert f.last_grid_dims == dims
        assert st == 0 and all(np.array_equal(o, ref) for o in outs.values())


def test_frames_in_page_locked_host_memory_build_the_same_pyramid(ctx):
    from align3d_amd import CameraIntrinsics, RangeImageBuilder

    s = SlamTbSample("sample1")
    depth, rgb = s.load(4)
    cam = CameraIntrinsics(*s.intrinsics(4), 640, 480)
    pd, pr = ctx.pinned_empty(depth.shape, depth.dtype), ctx.pinned_empty(rgb.shape, rgb.dtype)
    pd[...], pr[...] = depth, rgb
    b = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    a_levels = b.build_device(cam, depth, rgb, s.depth_scale(4))
    p_levels = b.build_device(cam, pd, pr, s.depth_scale(4))
    for x, y in zip(a_levels, p_levels):
        gx, gy = x.download(), y.download()
        assert np.array_equal(_bits(gx.points), _bits(gy.points)) and np.array_equal(gx.mask, gy.mask)
        assert np.array_equal(_bits(gx.normals), _bits(gy.normals)) and np.array_equal(gx.colors, gy.colors)
        assert np.array_equal(_bits(gx.intensity_map), _bits(gy.intensity_map))


def test_device_builder_without_minmax_round_trip_and_scratch_regrow():
    """From the second frame on a context the builder enqueues the bilateral filter without reading min/max back
    (grid dimensions, capacity check and launch bounds live on the device).  A later frame whose depth range needs
    a larger grid than the scratch region holds is detected on the device and rebuilt after growing the region.
    All pyramids equal the oracle's bit for bit."""
    from align3d_amd import CameraIntrinsics, Context, RangeImageBuilder

    rng = np.random.default_rng(5)
    w, h = 320, 240
    cam = CameraIntrinsics(300.0, 300.0, w / 2.0, h / 2.0, w, h)
    rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    base = (1000 + 400 * np.sin(np.linspace(0, 6, w))[None, :] + 300 * np.cos(np.linspace(0, 4, h))[:, None])
    frames = [
        (base + rng.integers(0, 30, size=(h, w))).astype(np.uint16),            # first frame: synchronous path
        (base * 1.1 + rng.integers(0, 30, size=(h, w))).astype(np.uint16),      # fits: asynchronous path
        (base * 9.0 + rng.integers(0, 3000, size=(h, w))).astype(np.uint16),    # far larger depth range: regrow
        (base * 0.5 + rng.integers(0, 10, size=(h, w))).astype(np.uint16),      # small again, large region
    ]
    frames[1][10:40, 20:90] = 0
    own = Context(0)
    try:
        b = RangeImageBuilder(own).with_bilateral_filter(BilateralFilter.default())
        for depth in frames:
            ref = O.build_pyramid(depth, rgb, cam.fx, cam.fy, cam.cx, cam.cy, 0.001, levels=3, use_bilateral=True)
            levels = b.build_device(cam, depth, rgb, 0.001)
            for lv, r in zip(levels, ref):
                _assert_same_level(lv, r)
            for lv in levels:
                lv.free()
    finally:
        own.close()


def test_batched_builder_equals_oracle_and_single_builds(ctx):
    """a3d_range_image_build_pyramids: 20 real frames in one launch sequence — every level of every frame
    equals the single-frame build bit for bit, and the oracle's pyramid on the frames checked against it; frames with
    very different depth ranges (= bilateral grids of different sizes, one of which outgrows the scratch region and
    forces the chunk to run again) share a batch."""
    from align3d_amd import CameraIntrinsics, Context, RangeImageBuilder
    from gpu_util import oracle_pyramid

    s = SlamTbSample("sample1")
    cam = CameraIntrinsics(*s.intrinsics(0), 640, 480)
    frames = [s.load(i) for i in range(20)]
    own = Context(0)  # its own scratch region: the capacity guess and the regrow path are exercised from scratch
    try:
        b = RangeImageBuilder(own).with_bilateral_filter(BilateralFilter.default())
        many = b.build_many(cam, frames, s.depth_scale(0))
        assert len(many) == 20 and all(len(p) == 3 for p in many)
        for i in (0, 1, 7, 15, 16, 19):
            for lv, r in zip(many[i], oracle_pyramid("sample1", i)):
                _assert_same_level(lv, r)
        for i in (3, 12, 18):
            single = b.build(cam, *frames[i], s.depth_scale(0))
            for lv, one in zip(many[i], single):
                a, c = lv.download(), one.download()
                for name in ("points", "mask", "normals", "colors", "intensities", "intensity_map"):
                    assert np.array_equal(getattr(a, name), getattr(c, name), equal_nan=True), (i, name)
            for lv in single:
                lv.free()
        for lv in (lv for p in many for lv in p):
            lv.free()
        # mixed depth ranges in one batch: frame 1's grid is ~9x deeper than the others'
        rng = np.random.default_rng(5)
        w, h = 262, 134
        cam2 = CameraIntrinsics(300.0, 300.0, w / 2.0, h / 2.0, w, h)
        rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        base = (1000 + 400 * np.sin(np.linspace(0, 6, w))[None, :] + 300 * np.cos(np.linspace(0, 4, h))[:, None])
        depths = [(base + rng.integers(0, 30, size=(h, w))).astype(np.uint16),
                  (base * 30.0 + rng.integers(0, 9000, size=(h, w))).astype(np.uint16),
                  np.zeros((h, w), np.uint16),
                  (base * 0.5).astype(np.uint16)]
        depths[0][10:40, 20:90] = 0
        mixed = b.build_many(cam2, [(d, rgb) for d in depths], 0.001)
        for d, pyr in zip(depths, mixed):
            ref = O.build_pyramid(d, rgb, cam2.fx, cam2.fy, cam2.cx, cam2.cy, 0.001, levels=3, use_bilateral=True)
            for lv, r in zip(pyr, ref):
                _assert_same_level(lv, r)
    finally:
        own.close()


def test_builder_rejects_bad_sigmas_and_mismatched_frames(ctx):
    from align3d_amd import A3dError, CameraIntrinsics, InvalidParameter, RangeImageBuilder

    cam = CameraIntrinsics(300.0, 300.0, 32.0, 24.0, 64, 48)
    depth = np.full((48, 64), 1000, np.uint16)
    rgb = np.zeros((48, 64, 3), np.uint8)
    for ss, sc in ((0.0, 30.0), (4.5, -1.0), (float("nan"), 30.0), (4.5, float("inf"))):
        with pytest.raises(A3dError) as e:
            RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.new(ss, sc)).build(cam, depth, rgb, 0.001)
        assert e.value.status == 1
    with pytest.raises(A3dError):
        RangeImageBuilder(ctx).blur_sigma(float("nan")).build(cam, depth, rgb, 0.001)
    with pytest.raises(InvalidParameter):
        RangeImageBuilder(ctx).build(cam, depth, rgb[:40], 0.001)          # rgb smaller than depth
    with pytest.raises(InvalidParameter):
        RangeImageBuilder(ctx).build(cam, depth.reshape(-1), rgb, 0.001)   # depth not 2-D
    with pytest.raises(InvalidParameter):
        RangeImageBuilder(ctx).build_many(cam, [(depth, rgb), (depth[:24], rgb[:24])], 0.001)  # frames of two sizes


def test_builder_reports_what_it_processed(ctx):
    """a3d_context_last_build_stats (the figures bench.py's frame-build roofline is made of): frames, cells of their
    bilateral grids, marked blur tiles; the grid cells equal the oracle's grid dimensions for each frame."""
    from align3d_amd import CameraIntrinsics, RangeImageBuilder

    s1 = SlamTbSample("sample1")
    frames = [s1.load(0), s1.load(1), s1.load(4)]
    built = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default()).build_many(
        CameraIntrinsics(*s1.intrinsics(0), 640, 480), frames, s1.depth_scale(0))
    st = ctx.last_build_stats()
    want_cells = 0
    for depth, _ in frames:
        status, _, dims = O.bilateral(depth)
        assert status == 0
        want_cells += int(np.prod(dims))
    assert st["frames"] == 3 and st["grid_cells"] == want_cells
    assert 0 < st["marked_tiles"] * 12 ** 3 < 3 * want_cells and st["zero_tiles"] > 0
    for p in built:
        for lv in p:
            lv.free()
    RangeImageBuilder(ctx).build_many(CameraIntrinsics(*s1.intrinsics(0), 640, 480), frames[:1], s1.depth_scale(0))
    assert ctx.last_build_stats() == {"frames": 1, "grid_cells": 0, "marked_tiles": 0, "zero_tiles": 0}


def test_compute_normals_wide_dynamic_range_and_signed_zeros(ctx):
    """The kernels decide the two ratio tests by exact comparisons and take n / |n| through a shared reciprocal
    (devmath.hpp, normal_from_neighbours_dev): point clouds scaled from 1e-12 to 1e12 (quotients outside the fast
    division's range take the plain one), axis-aligned planes (cross products with -0.0 components), zero and huge
    neighbour distances (ratios 0, inf, NaN, exactly 4 and exactly 1/4) — all bit for bit the oracle's."""
    from align3d_amd import CameraIntrinsics, RangeImage

    rng = np.random.default_rng(99)
    h, w = 48, 80
    cases = []
    for scale in (1e-12, 1e-6, 1e-3, 1.0, 1e4, 1e9, 1e12):
        cases.append((rng.normal(size=(h, w, 3)) * scale).astype(np.float32))
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    for plane in ((xx, yy, np.zeros_like(xx)), (-xx, yy, np.full_like(xx, 2.0)), (xx, -yy, xx * 0), (yy * 0, xx, -yy)):
        cases.append(np.stack(plane, axis=-1).astype(np.float32))
    # neighbour distances in exact ratios 4, 1/4, 1 and with repeated points (0 / 0)
    steps = np.array([1.0, 2.0, 1.0, 0.5, 0.0, 4.0, 1.0, 1.0], np.float32)
    xs = np.concatenate([[0.0], np.cumsum(np.tile(steps, w // 8 + 1))])[:w].astype(np.float32)
    cases.append(np.stack([np.broadcast_to(xs, (h, w)), np.broadcast_to(xs[:h, None], (h, w)) if h <= w else yy, np.ones((h, w), np.float32)], axis=-1).astype(np.float32))
    for k, pts in enumerate(cases):
        pts = np.ascontiguousarray(pts)
        mask = (rng.random((h, w)) > 0.1).astype(np.uint8)
        ref = O.compute_normals(pts, mask)
        got = RangeImage(pts, mask, CameraIntrinsics(100, 100, w / 2, h / 2, w, h)).compute_normals(ctx).normals
        assert np.array_equal(_bits(got), _bits(ref)), (k, int(np.sum(_bits(got) != _bits(ref))))


def test_builder_keeps_its_grid_scratch_clean_between_enqueues():
    """The builder never clears its bilateral grids: the last kernel of an enqueue zeroes the cells its splat wrote and
    the tile flags (a3d_context::grid_clean).  One context, a sequence chosen to break that if anything is left behind:
    two launch sequences in one call (50 frames), frames whose grids are sparse, dense (full-range noise: every
    footprint spans dozens of channels, i.e. several splat rounds), empty and 2000 channels deep (more channel tiles
    than the splat's tile bits hold and more tiles than the blur lists in LDS: the general paths), the host filter on
    another image size in between (same scratch region, another layout), then the first frames again.  Everything
    equals the oracle bit for bit."""
    from align3d_amd import CameraIntrinsics, Context, RangeImageBuilder

    rng = np.random.default_rng(11)

    def scene(w, h, kind):
        base = 1500 + 500 * np.sin(np.linspace(0, 5, w))[None, :] + 300 * np.cos(np.linspace(0, 3, h))[:, None]
        if kind == "smooth":
            d = base + rng.integers(0, 20, size=(h, w))
        elif kind == "steps":   # depth edges: foreground squares 1.2 m in front of the background
            d = base + 6000.0
            d[h // 4: h // 2, w // 4: w // 2] -= 6000.0
            d[h // 2 + 3: h - 9, 5: w // 3] = 0
        elif kind == "noise":   # every pixel anywhere in the u16 range
            d = rng.integers(1, 65000, size=(h, w)).astype(np.float64)
        else:
            d = np.zeros((h, w))
        return d.astype(np.uint16), rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)

    def check(b, cam, frames, which=None):
        many = b.build_many(cam, frames, 0.001)
        for i in (range(len(frames)) if which is None else which):
            d, rgb = frames[i]
            ref = O.build_pyramid(d, rgb, cam.fx, cam.fy, cam.cx, cam.cy, 0.001, levels=3, use_bilateral=True)
            for lv, r in zip(many[i], ref):
                _assert_same_level(lv, r)
        for lv in (lv for p in many for lv in p):
            lv.free()

    own = Context(0)
    try:
        b = RangeImageBuilder(own).with_bilateral_filter(BilateralFilter.default())
        w, h = 320, 240
        cam = CameraIntrinsics(300.0, 300.0, w / 2.0, h / 2.0, w, h)
        big = [scene(w, h, k) for k in ("smooth", "steps", "noise", "zeros", "smooth")]
        check(b, cam, big)
        check(b, cam, [big[4], big[1]])
        # the host filter on the same context and scratch region, another image size
        fw, fh = 200, 120
        fd, _ = scene(fw, fh, "steps")
        f = BilateralFilter.default()
        got = f.filter(own, fd)
        assert np.array_equal(got, O.bilateral(fd, f.sigma_space, f.sigma_color)[1])
        check(b, cam, [big[0], big[2], big[1]])
        # 50 small frames: two launch sequences of 25
        sw, sh = 96, 64
        cam_s = CameraIntrinsics(90.0, 90.0, sw / 2.0, sh / 2.0, sw, sh)
        small = [scene(sw, sh, ("smooth", "steps", "noise", "zeros")[i % 4]) for i in range(50)]
        check(b, cam_s, small, which=(0, 1, 2, 3, 24, 25, 26, 49))
        check(b, cam, [big[1]])
    finally:
        own.close()


def test_bilateral_wide_cells_for_a_large_sigma_space():
    """sigma_space >= 15: more than 255 pixels can fall into one grid (row, column), so the packed cells are u64
    (value sum << 24 | count) and the blur carries f64 throughout — the other instantiation of splat / blur / unsplat.
    The host filter and the frame builder (two enqueues on one context: the grid scratch must come back clean) against
    the oracle, bit for bit; footprints of 16-17 x 16-17 pixels also take several load patches per grid column."""
    from align3d_amd import CameraIntrinsics, Context, RangeImageBuilder

    rng = np.random.default_rng(21)
    w, h = 200, 150
    base = 1200 + 500 * np.sin(np.linspace(0, 5, w))[None, :] + 300 * np.cos(np.linspace(0, 3, h))[:, None]
    d0 = (base + rng.integers(0, 25, size=(h, w))).astype(np.uint16)
    d0[40:70, 30:90] = 0
    d1 = base.copy()
    d1[h // 3: 2 * h // 3, w // 3: 2 * w // 3] += 5000.0   # a depth step
    d1 = d1.astype(np.uint16)
    rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    cam = CameraIntrinsics(180.0, 180.0, w / 2.0, h / 2.0, w, h)
    own = Context(0)
    try:
        for ss, sc in ((16.0, 40.0), (23.5, 300.0)):
            f = BilateralFilter.new(ss, sc)
            for d in (d0, d1):
                st, ref, dims = O.bilateral(d, ss, sc)
                out = f.filter(own, d)
                assert st == 0 and f.last_grid_dims == dims
                assert np.array_equal(out, ref), (ss, sc)
            b = RangeImageBuilder(own).with_bilateral_filter(f)
            for rep in range(2):
                many = b.build_many(cam, [(d0, rgb), (d1, rgb), (d0, rgb)], 0.001)
                for (d, c), pyr in zip([(d0, rgb), (d1, rgb), (d0, rgb)], many):
                    filtered = O.bilateral(d, ss, sc)[1]  # (the oracle's builder with this filter: filter, then build)
                    ref = O.build_pyramid(filtered, c, cam.fx, cam.fy, cam.cx, cam.cy, 0.001, levels=3, use_bilateral=False)
                    for lv, r in zip(pyr, ref):
                        _assert_same_level(lv, r)
                for lv in (lv for p in many for lv in p):
                    lv.free()
    finally:
        own.close()


@pytest.mark.parametrize("w,h,levels,bilateral", [(640, 480, 3, True), (640, 480, 4, False), (100, 68, 3, True), (97, 61, 3, True),
                                                  (160, 96, 4, True), (34, 30, 2, True), (36, 40, 3, False)])
def test_level0_quad_kernel_equals_the_patch_kernel_and_the_oracle(ctx, diag_ctx, monkeypatch, w, h, levels, bilateral):
    """Round 6: level0_quad_kernel (32 x 32 aligned patches, a 2 x 2 quad per thread, level-1 picks from registers, level-2
    picks from the patch's level-1 picks) against the round-2..5 kernel (A3D_BUILDER_L0=patch), against the separate
    resize kernel for level 2 / for levels 1 and 2 (A3D_BUILDER_FUSE_L2=0 / _L1=0) and against the oracle: sizes that are
    multiples of 32, of 4 only, of 2 only, odd; three- and four-level pyramids; with and without the bilateral filter.
    The diagnostics-build runs start from a blurred grid full of NaN (A3D_BILATERAL_POISON): a slice that read a cell no
    marked tile wrote (the margin-1 tile marking of splat_packed_kernel) fails the u16 cast check instead of passing on
    an earlier frame's value."""
    from align3d_amd import CameraIntrinsics, RangeImageBuilder

    rng = np.random.default_rng(w * 977 + h)
    yy, xx = np.mgrid[0:h, 0:w]
    depth = (900 + 6.0 * xx + 3.0 * yy + 250 * (xx > w // 2) + rng.integers(0, 25, size=(h, w))).astype(np.uint16)
    depth[rng.random((h, w)) < 0.08] = 0
    depth[h // 3: h // 3 + 7, w // 4: w // 4 + 9] = 0
    rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    cam = CameraIntrinsics(0.9 * w, 0.9 * w, w / 2.0, h / 2.0, w, h)
    if levels > 1 and ((w >> (levels - 1)) < 2 or (h >> (levels - 1)) < 2):
        pytest.skip("image too small")

    def build(c):
        b = RangeImageBuilder(c).pyramid_levels(levels)
        if bilateral:
            b = b.with_bilateral_filter(BilateralFilter.default())
        pyr = b.build_device(cam, depth, rgb, 0.001)
        out = [lv.download() for lv in pyr]
        for lv in pyr:
            lv.free()
        return out

    for k in ("A3D_BUILDER_L0", "A3D_BUILDER_FUSE_L1", "A3D_BUILDER_FUSE_L2", "A3D_BUILDER_UNSPLAT", "A3D_BILATERAL_MINMAX"):
        monkeypatch.delenv(k, raising=False)
    got = build(ctx)
    ref = O.build_pyramid(depth, rgb, cam.fx, cam.fy, cam.cx, cam.cy, 0.001, levels=levels, use_bilateral=bilateral)
    for g, r in zip(got, ref):
        assert np.array_equal(g.mask, r.mask) and np.array_equal(_bits(g.points), _bits(r.points))
        assert np.array_equal(_bits(g.normals), _bits(r.normals))
    monkeypatch.setenv("A3D_BILATERAL_POISON", "1")
    knobs = ("A3D_BUILDER_L0", "A3D_BUILDER_FUSE_L1", "A3D_BUILDER_FUSE_L2", "A3D_BUILDER_UNSPLAT", "A3D_BILATERAL_MINMAX")
    # (also two fusions that were measured no faster and live in the diagnostics build only: the filter's zeros put back by
    # the level-0 kernel instead of by unsplat_kernel; min / max and the grid sizing in one launch, the frame's last block
    # sizing — alternating with the product's path, so that each leaves the scratch region's zero invariant for the other)
    for env in ({}, {"A3D_BUILDER_L0": "patch"}, {"A3D_BUILDER_FUSE_L2": "0"}, {"A3D_BUILDER_UNSPLAT": "fused"}, {},
                {"A3D_BILATERAL_MINMAX": "fused"}, {"A3D_BUILDER_FUSE_L1": "0"}, {}):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        other = build(diag_ctx)
        for g, o in zip(got, other):
            for name in ("points", "mask", "normals", "colors", "intensities", "intensity_map"):
                assert np.array_equal(getattr(g, name), getattr(o, name), equal_nan=True), (env, name)


def test_bilateral_filter_on_resident_images_equals_the_oracle(ctx):
    """a3d_bilateral_filter_u16_device: u16 images in device memory in, u16 out, a batch per call —
    every image bit for bit the oracle's BilateralFilter::filter (and so the host-pointer form's): real depth frames, a
    synthetic scene with depth edges and holes, an all-zero image, a batch larger than one launch sequence, a grid that
    outgrows the scratch region on a fresh context."""
    from align3d_amd import Context

    s = SlamTbSample("sample1")
    real = np.stack([s.load(i)[0] for i in (0, 5, 11)]).astype(np.uint16)
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:96, 0:128]
    synth_imgs = []
    for k in range(37):  # more than one launch sequence of 32
        d = (700 + 9.0 * xx + 4.0 * yy + 600 * ((xx + 5 * k) % 128 > 70) + rng.integers(0, 30, size=xx.shape)).astype(np.uint16)
        d[rng.random(d.shape) < 0.05] = 0
        if k == 7:
            d[:] = 0
        if k == 9:
            d = (d.astype(np.uint32) * 13 % 60000).astype(np.uint16)  # a deep grid: the scratch region grows
        synth_imgs.append(d)
    synth_imgs = np.stack(synth_imgs)
    f = BilateralFilter.default()
    own = Context(0)
    try:
        for imgs in (real, synth_imgs):
            n, h, w = imgs.shape
            d_in, d_out = own.to_device(np.ascontiguousarray(imgs)), own.malloc(imgs.nbytes)
            f.filter_device(own, d_in, n, w, h, d_out)
            got = np.empty_like(imgs)
            own.to_host(d_out, got)
            own.free(d_in), own.free(d_out)
            for k in range(n):
                st, ref, _ = O.bilateral(imgs[k], f.sigma_space, f.sigma_color)
                assert st == 0 and np.array_equal(got[k], ref), k
    finally:
        own.close()
