"""GPU parity: compute_normals and the bilateral filter are bit-exact against the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from align3d_amd import BilateralFilter
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
