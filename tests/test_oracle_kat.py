"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md §4 / §8c).  Runs on CPU."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from align3d_amd._abi import GnStateC, PoseC, ptr
from data_util import GOLDEN, SlamTbSample


# src/optim/gaussnewton.rs:141-167
def test_gauss_newton_kat(orc):
    r = np.array([1, 2, 3], np.float32)
    J = np.tile(np.arange(1, 7, dtype=np.float32), (3, 1))
    gn = GnStateC()
    orc.orc_gn_steps(C.c_void_p(ptr(r)), C.c_void_p(ptr(J)), 3, C.byref(gn))
    d = gn.as_dict()
    expected_h = np.outer(np.arange(1, 7), np.arange(1, 7)).astype(np.float32) * 3
    assert np.array_equal(d["H"], expected_h)
    assert np.array_equal(d["g"], np.array([6, 12, 18, 24, 30, 36], np.float32))
    assert d["count"] == 3 and d["ssq"] == 14.0


# gaussnewton.rs:115-133: add_weighted squares the weight on H only; counts add
def test_gauss_newton_add_weighted_semantics(orc):
    a, b = GnStateC(), GnStateC()
    r = np.array([2.0], np.float32)
    J = np.array([[1, 0, 0, 0, 0, 0]], np.float32)
    orc.orc_gn_steps(C.c_void_p(ptr(r)), C.c_void_p(ptr(J)), 1, C.byref(a))
    orc.orc_gn_steps(C.c_void_p(ptr(r)), C.c_void_p(ptr(J)), 1, C.byref(b))
    orc.orc_gn_add_weighted(C.byref(a), C.byref(b), 1.0, 0.5)
    d = a.as_dict()
    assert d["H"][0, 0] == np.float32(1.0 + 0.25)
    assert d["g"][0] == np.float32(2.0 + 1.0)
    assert d["ssq"] == np.float32(4.0 + 2.0)
    assert d["count"] == 2
    assert orc.orc_gn_mean_squared_residual(C.byref(a)) == np.float32(3.0)


def test_gauss_newton_solve_none_cases(orc):
    empty = GnStateC()
    out = (C.c_float * 6)()
    assert orc.orc_gn_solve(C.byref(empty), out) == 0  # count == 0
    singular = GnStateC()
    r = np.array([1.0], np.float32)
    J = np.array([[1, 2, 3, 4, 5, 6]], np.float32)
    orc.orc_gn_steps(C.c_void_p(ptr(r)), C.c_void_p(ptr(J)), 1, C.byref(singular))
    assert orc.orc_gn_solve(C.byref(singular), out) == 0  # rank-1 H: Cholesky fails
    # a well-posed system solves H x = g
    rng = np.random.default_rng(0)
    Jr = rng.normal(size=(50, 6)).astype(np.float32)
    rr = rng.normal(size=50).astype(np.float32)
    gn = GnStateC()
    orc.orc_gn_steps(C.c_void_p(ptr(rr)), C.c_void_p(ptr(Jr)), 50, C.byref(gn))
    assert orc.orc_gn_solve(C.byref(gn), out) == 1
    d = gn.as_dict()
    x = np.linalg.solve(d["H"].astype(np.float64), d["g"].astype(np.float64))
    assert np.allclose(np.array(out[:]), x, rtol=1e-5, atol=1e-6)


# src/transform.rs:364-388
def test_exp_kat():
    T = O.exp_se3([1.0, 2.0, 3.0, 0.4, 0.5, 0.3])
    out = O.transform_points(T, np.array([[5.5, 6.4, 7.8]], np.float32))
    assert np.all(np.abs(out - np.array([[8.9848175, 6.9635687, 9.880962]], np.float32)) < 1e-5)
    out = O.transform_points(T, np.array([[1.0, 2.0, 3.0]], np.float32))
    assert np.linalg.norm(out[0] - np.array([3.5280778, 2.8378963, 5.8994026], np.float32)) < 1e-5
    m = O.pose_to_matrix(T)
    v = m @ np.array([1, 2, 3, 1], np.float32)
    assert np.allclose(v, [3.5280778, 2.8378963, 5.8994026, 1.0], atol=2e-6)


# src/transform.rs:321-362 (test_mul_op, test_transform): rotate by pi about y, translate z+3
def test_transform_vector_kat():
    eye = O.pose()
    pts = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float32)
    assert np.array_equal(O.transform_points(eye, pts), pts)
    half = np.float32(np.pi) / np.float32(2)
    T = O.pose(t=(0, 0, 3), q=(0, np.sin(half), 0, np.cos(half)))
    out = O.transform_points(T, np.array([[1, 2, 3], [1, 2, 3]], np.float32))
    assert np.all(np.abs(out - np.array([[-1, 2, 0], [-1, 2, 0]], np.float32)) < 1e-5)


# src/transform.rs:390-411
def test_compose_kat():
    T1 = O.pose(t=(0, 0, 3))
    q = np.float32(np.pi) / np.float32(4)
    T2 = O.pose(t=(0, 0, 3), q=(0, np.sin(q), 0, np.cos(q)))
    T = O.compose(T1, T2)
    out = O.transform_points(T, np.array([[1, 2, 3]], np.float32))
    assert np.all(np.abs(out - np.array([[2.9999998, 2.0, 5.0]], np.float32)) < 1e-5)


# src/camera.rs:210-243
def test_project_kat(orc):
    uv = np.zeros(2, np.float32)
    p = np.array([1, 1, 1], np.float32)
    orc.orc_project(50.0, 50.0, 0.0, 0.0, ptr(p), ptr(uv))
    assert tuple(uv) == (50.0, 50.0)
    p = np.array([1, 1.5, 1], np.float32)
    orc.orc_project(50.0, 50.0, 0.0, 0.0, ptr(p), ptr(uv))
    assert tuple(uv) == (50.0, 75.0)


# src/metrics.rs:78-93
def test_transform_metrics_kat():
    a = O.pose(t=(0.00022050377, 7.3633055e-5, -1.51071e-5), q=(0.00888227, 0.0008264509, 0.99996024, 2.059626e-5))
    ang, tr = O.transform_metrics(a, a)
    assert tr == 0.0 and abs(ang) < 1e-3  # identical transforms (the reference builds q via normalisation)


# src/kdtree.rs:121-139
def test_kdtree_small_kat():
    pts = np.array([[1, 2, 3], [2, 3, 4], [5, 6, 7], [8, 9, 1]], np.float32)
    tree = O.KdTree(pts)
    q = np.array([[8, 9.1, 1.3], [5.1, 6.4, 7.0], [1.5, 2.1, 3.3], [2.2, 3.1, 4.2]], np.float32)
    idx, _ = tree.nearest(q)
    assert list(idx) == [3, 2, 0, 1]


# src/kdtree.rs:142-170 in property form: for ANY permutation every point finds its own index
@pytest.mark.parametrize("seed", [0, 5, 1234])
def test_kdtree_500_self_query(seed):
    ordered = np.arange(1500, dtype=np.float32).reshape(500, 3)
    perm = np.random.default_rng(seed).permutation(500)
    randomized = np.empty_like(ordered)
    randomized[perm] = ordered
    tree = O.KdTree(randomized)
    idx, d = tree.nearest(ordered)
    assert np.array_equal(idx, perm.astype(np.uint64))
    assert np.all(d == 0)


def test_kdtree_shape_is_function_of_n():
    # N = 500 -> leaves of 15/16 points
    tree = O.KdTree(np.random.default_rng(1).random((500, 3), dtype=np.float32))
    leaves, internal, depth = tree.stats()
    assert (leaves, internal, depth) == (32, 31, 5)


def test_kdtree_nan_is_an_error():
    pts = np.random.default_rng(2).random((40, 3), dtype=np.float32)
    pts[7, 0] = np.nan
    assert O.KdTree(pts).status == 5  # A3D_NAN_IN_INPUT: partial_cmp().unwrap() panics


# src/range_image/structure.rs:479-485 and :453-476, src/io/dataset/slamtb.rs:161-173
def test_sample1_valid_points_and_normals():
    s = SlamTbSample("sample1")
    assert s.intrinsics(0) == (544.4732666015625, 544.4732666015625, 320.0, 240.0)
    depth, rgb = s.load(0)
    fr = O.build_frame(depth, rgb, *s.intrinsics(0), s.depth_scale(0))
    assert int(fr.mask.sum()) == 270213
    assert fr.normals.shape == (480, 640, 3)
    assert abs(np.linalg.norm(fr.normals[44, 42]) - 1.0) < 1e-6


# src/intensity_map.rs:229-262 (property form on any u8 image)
def test_intensity_map_properties(orc):
    rng = np.random.default_rng(3)
    luma = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    m = O.intensity_map(luma)
    h, w = luma.shape
    assert m.shape == (h + 2, w + 2)
    assert np.array_equal(m[:h, :w], luma.astype(np.float32) / np.float32(255.0))
    # border_should_repeat: rows h, h+1 repeat row h-1 (all but the last column), same for columns
    assert np.array_equal(m[h, : w - 1], m[h - 1, : w - 1]) and np.array_equal(m[h + 1, : w - 1], m[h - 1, : w - 1])
    assert np.array_equal(m[: h - 1, w], m[: h - 1, w - 1]) and np.array_equal(m[: h - 1, w + 1], m[: h - 1, w - 1])
    # the quirk (intensity_map.rs:60-78): these six cells stay zero
    for rc in [(h, w - 1), (h + 1, w - 1), (h - 1, w), (h - 1, w + 1), (h, w + 1), (h + 1, w)]:
        assert m[rc] == 0.0
    assert m[h, w] == m[h - 1, w - 1] == m[h + 1, w + 1]
    # round_uv_should_match_image: bilinear at integer coordinates equals the pixel
    out = np.zeros(3, np.float32)
    for (r, c) in [(0, 0), (5, 7), (h - 1, w - 1), (20, 1)]:
        orc.orc_intensity_map_bilinear_grad(ptr(m), w, h, float(c), float(r), ptr(out))
        assert out[0] == m[r, c]


# src/bilateral/grid.rs:183-185: grid dims follow the formula (bloei luma16: 600x450, max 5041, min 0)
def test_bilateral_grid_dims_formula():
    img = np.zeros((600, 450), np.uint16)
    img[0, 0] = 5041
    st, out, dims = O.bilateral(img, 4.5, 30.0, blur=False)
    assert st == 0 and dims == (138, 104, 173)


def test_bilateral_constant_image_is_fixed_point():
    img = np.full((48, 64), 1000, np.uint16)
    st, out, dims = O.bilateral(img)
    assert st == 0
    # interior pixels of a constant image stay (nearly) constant; truncation may lose 1
    assert np.all(np.abs(out[8:-8, 8:-8].astype(int) - 1000) <= 1)


# src/icp/image_icp.rs:181-200: sample2 frames 0/1, bilateral, IcpParams::default with 5 iterations
def test_image_icp_smoke_threshold():
    s = SlamTbSample("sample2")
    f0 = O.build_frame(*s.load(0), *s.intrinsics(0), s.depth_scale(0), use_bilateral=True)
    f1 = O.build_frame(*s.load(1), *s.intrinsics(1), s.depth_scale(1), use_bilateral=True)
    st, T, trace = O.image_icp_align(O.params(max_iterations=5), f0, f1, threads=4, want_trace=True)
    assert st == 0
    gt = O.compose(_inverse(O.pose_from_matrix(s.rt_cam(0))), O.pose_from_matrix(s.rt_cam(1)))
    ang, _ = O.transform_metrics(T, gt)
    assert abs(ang) < 0.01
    assert np.all(np.isfinite(trace))


def _inverse(p):
    out = PoseC()
    O.load().orc_inverse(C.byref(p), C.byref(out))
    return out


# src/icp/multiscale.rs:30-34 and :82-96
def test_multiscale_new_length_mismatch_and_smoke():
    s = SlamTbSample("sample1")
    tp = O.build_pyramid(*s.load(0), *s.intrinsics(0), s.depth_scale(0), levels=3)
    sp = O.build_pyramid(*s.load(4), *s.intrinsics(4), s.depth_scale(4), levels=3)
    assert [f.w for f in tp] == [640, 320, 160]
    prm = O.ms_default_params()
    st, _ = O.multiscale_align(prm, 2, tp, sp)
    assert st == 1  # A3D_INVALID_PARAMETER
    for i in range(3):
        prm[i].max_iterations = 3
    st, T = O.multiscale_align(prm, 3, tp, sp, threads=4)
    assert st == 0 and np.all(np.isfinite(O.pose_tuple(T)[0]))


def bloei_luma16():
    """unit_test/images.rs:28-40: bloei.jpg -> into_luma16() -> v /= u16::MAX / 5000.  The JPEG is decoded by Pillow
    here and by the `image` crate's jpeg-decoder 0.3.0 in the reference (IDCT rounding may differ by one grey level);
    image-0.24.7's Rgb8 -> Luma16 is (2126 r + 7152 g + 722 b) / 10000 in integers, then x 257."""
    from PIL import Image

    rgb = np.array(Image.open(os.path.join(GOLDEN, "images", "bloei.jpg")).convert("RGB"), np.uint32)
    l8 = (2126 * rgb[..., 0] + 7152 * rgb[..., 1] + 722 * rgb[..., 2]) // 10000
    return ((l8 * 257).astype(np.uint16) // np.uint16(65535 // 5000)).astype(np.uint16)


# src/bilateral/grid.rs:183-194: the one value-level known answer the reference holds for the bilateral grid
def test_bilateral_grid_slice_known_answer_near_pin():
    img = bloei_luma16()
    assert img.shape == (600, 450)
    st, out, dims = O.bilateral(img, 4.5, 30.0, blur=False)  # BilateralGrid::from_image + normalize + slice
    assert st == 0 and dims == (138, 104, 173)               # verify_grid_creation: dim() == (138, 104, 173, 2)
    assert out.shape == (600, 450)
    # verify_slice: dest_image[(421, 123)] == 2266 with the reference's JPEG decoder; Pillow's decode of the same
    # file differs by at most one grey level per channel, which moves this value by at most one count (2265 here)
    assert abs(int(out[421, 123]) - 2266) <= 1


def test_threaded_normals_equal_the_sequential_loop():
    """orc_compute_normals_mt (bench.py's CPU baseline form: the reference's 1024-pixel rayon chunks over threads)
    writes exactly what the sequential pixel loop writes."""
    s = SlamTbSample("sample1")
    fr = O.build_frame(*s.load(1), *s.intrinsics(1), s.depth_scale(1))
    for threads in (2, 5, 16):
        assert np.array_equal(O.compute_normals(fr.points, fr.mask, threads=threads).view(np.uint32),
                              fr.normals.view(np.uint32))


# ---- the oracle against a second, independently written restatement (tests/numpy_restatement.py) ----------------

def test_two_restatements_agree_on_the_bilateral_filter_bit_for_bit():
    import numpy_restatement as NP

    s = SlamTbSample("sample1")
    depth, _ = s.load(4)
    crop = np.ascontiguousarray(depth[100:292, 200:456])  # 192 x 256 of real depth with holes and edges
    st, ref, dims = O.bilateral(crop)
    got, gdims = NP.bilateral_filter_u16(crop)
    assert st == 0 and gdims == dims and np.array_equal(got, ref)
    img = bloei_luma16()[150:330, 100:300]                # the reference's KAT image, other sigmas
    st, ref, dims = O.bilateral(img, 4.5, 30.0)
    got, gdims = NP.bilateral_filter_u16(img, 4.5, 30.0)
    assert st == 0 and gdims == dims and np.array_equal(got, ref)


def test_two_restatements_agree_on_the_normals_bit_for_bit():
    import numpy_restatement as NP

    for sample, frame in (("sample1", 0), ("sample2", 1)):
        s = SlamTbSample(sample)
        fr = O.build_frame(*s.load(frame), *s.intrinsics(frame), s.depth_scale(frame))
        got = NP.compute_normals(fr.points, fr.mask)
        assert np.array_equal(got.view(np.uint32), fr.normals.view(np.uint32)), sample


@pytest.mark.parametrize("which", ["default", "msdefault"])
def test_two_restatements_agree_on_the_image_icp_pixel_loop(which):
    """Inlier counts exact; H, g, sum r^2 of both terms equal to f64 round-off (the per-sample f32 values are the
    same numbers in both restatements, only the order of the f64 additions differs)."""
    import numpy_restatement as NP

    s = SlamTbSample("sample1")
    ft = O.build_frame(*s.load(0), *s.intrinsics(0), s.depth_scale(0))
    fs = O.build_frame(*s.load(5), *s.intrinsics(5), s.depth_scale(5))
    prm = O.params() if which == "default" else O.ms_default_params()[0]
    for T in (O.pose(), O.exp_se3(np.array([0.004, -0.003, 0.002, 0.003, 0.002, -0.004], np.float32))):
        st, g_ref, c_ref = O.image_icp_accumulate(prm, ft, fs, T, accum_f64=True)
        assert st == 0
        rg, Jg, rc, Jc = NP.image_icp_terms(prm, ft, fs, np.array(T.t[:], np.float32), np.array(T.q[:], np.float32))
        for (r, J), ref in (((rg, Jg), g_ref.as_dict()), ((rc, Jc), c_ref.as_dict())):
            H, g, ssq, count = NP.gn_sums(r, J)
            assert count == ref["count"] and count > 100000
            assert np.allclose(H, ref["H"], rtol=2e-7, atol=0) and np.allclose(g, ref["g"], rtol=2e-6, atol=1e-9)
            assert abs(ssq - float(ref["ssq"])) <= 2e-7 * ssq
