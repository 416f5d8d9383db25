"""GPU parity: R3dTree::nearest indices and squared distances are bit-exact against the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from align3d_amd import A3dError, R3dTree
from data_util import uniform01
from gpu_util import oracle_frame

pytestmark = pytest.mark.gpu


# src/kdtree.rs:121-139
def test_kdtree_small_kat(ctx):
    pts = np.array([[1, 2, 3], [2, 3, 4], [5, 6, 7], [8, 9, 1]], np.float32)
    q = np.array([[8, 9.1, 1.3], [5.1, 6.4, 7.0], [1.5, 2.1, 3.3], [2.2, 3.1, 4.2]], np.float32)
    idx, _ = R3dTree.new(ctx, pts).nearest(q)
    assert list(idx) == [3, 2, 0, 1]


# src/kdtree.rs:142-170 (property form)
def test_kdtree_500_self_query(ctx):
    ordered = np.arange(1500, dtype=np.float32).reshape(500, 3)
    perm = np.random.default_rng(5).permutation(500)
    randomized = np.empty_like(ordered)
    randomized[perm] = ordered
    idx, d = R3dTree.new(ctx, randomized).nearest(ordered)
    assert np.array_equal(idx, perm.astype(np.uint64)) and np.all(d == 0)


@pytest.mark.parametrize("n,m", [(1, 10), (16, 100), (17, 100), (33, 1000), (1000, 5000), (50000, 50000),
                                 (270213, 20000)])
def test_kdtree_random_bit_exact(ctx, n, m):
    db = uniform01(10 + n, 3 * n).reshape(n, 3)
    q = uniform01(11 + n, 3 * m).reshape(m, 3)
    ref = O.KdTree(db)
    tree = R3dTree.new(ctx, db)
    assert tree.stats() == ref.stats()
    ridx, rd = ref.nearest(q)
    gidx, gd = tree.nearest(q)
    assert np.array_equal(gidx, ridx)
    assert np.array_equal(gd.view(np.uint32), rd.view(np.uint32))


def test_kdtree_duplicates_ties_and_signed_zero(ctx):
    # many equal coordinates (stable-sort order matters), +-0.0, queries exactly on split planes
    rng = np.random.default_rng(3)
    db = rng.integers(-3, 4, size=(5000, 3)).astype(np.float32)
    db[::7, 0] = -0.0
    q = np.concatenate([db[:2000], rng.integers(-4, 5, size=(3000, 3)).astype(np.float32)])
    ridx, rd = O.KdTree(db).nearest(q)
    gidx, gd = R3dTree.new(ctx, db).nearest(q)
    assert np.array_equal(gidx, ridx) and np.array_equal(gd.view(np.uint32), rd.view(np.uint32))


def test_kdtree_range_image_cloud_and_nan_queries(ctx):
    fr = oracle_frame("sample1", 0)
    m = fr.mask.reshape(-1) != 0
    db = np.ascontiguousarray(fr.points.reshape(-1, 3)[m])
    fr1 = oracle_frame("sample1", 1)
    q = np.ascontiguousarray(fr1.points.reshape(-1, 3)[fr1.mask.reshape(-1) != 0][:100000])
    q[5] = np.nan
    q[6, 1] = np.inf
    ridx, rd = O.KdTree(db).nearest(q)
    gidx, gd = R3dTree.new(ctx, db).nearest(q)
    assert np.array_equal(gidx, ridx) and np.array_equal(gd.view(np.uint32), rd.view(np.uint32))


def test_kdtree_nan_point_is_an_error(ctx):
    db = uniform01(1, 300).reshape(100, 3)
    db[50, 0] = np.nan
    with pytest.raises(A3dError) as e:
        R3dTree.new(ctx, db)
    assert e.value.status == 5


def test_kdtree_full_size_500k_bit_exact(ctx):
    """BASELINE's kd-tree shape: 500 000 database x 500 000 queries, uniform [0,1)^3 (benches/bench_kdtree.rs)."""
    n = 500_000
    db = uniform01(10, 3 * n).reshape(n, 3)
    q = uniform01(11, 3 * n).reshape(n, 3)
    ref = O.KdTree(db)
    tree = R3dTree.new(ctx, db)
    assert tree.stats() == ref.stats() == (32768, 32767, 15)
    ridx, rd = ref.nearest(q)
    gidx, gd = tree.nearest(q)
    assert np.array_equal(gidx, ridx) and np.array_equal(gd.view(np.uint32), rd.view(np.uint32))
    # size-independent properties: every database point finds itself at distance 0 (distinct coordinates),
    # and the search is a pure function of the query (idempotent on repeat)
    sidx, sd = tree.nearest(db)
    assert np.array_equal(sidx, np.arange(n, dtype=np.uint64)) and not sd.any()
    g2, d2 = tree.nearest(q)
    assert np.array_equal(g2, gidx) and np.array_equal(d2, gd)


def _build(ctx, db, mode, monkeypatch, wide_len=None, sort=None):
    monkeypatch.setenv("A3D_KDTREE_BUILD", mode)
    if sort is not None:
        monkeypatch.setenv("A3D_KDTREE_SORT", sort)
    else:
        monkeypatch.delenv("A3D_KDTREE_SORT", raising=False)
    if wide_len is not None:
        monkeypatch.setenv("A3D_KDTREE_WIDE_LEN", str(wide_len))
    else:
        monkeypatch.delenv("A3D_KDTREE_WIDE_LEN", raising=False)
    return R3dTree.new(ctx, db)


def _kd_cases():
    rng = np.random.default_rng(17)
    dup = rng.integers(-3, 4, size=(20000, 3)).astype(np.float32)
    dup[::7, 0] = -0.0
    dup[::5, 1] = 0.0
    neg = (uniform01(3, 3 * 70001).reshape(-1, 3) - 0.5) * np.float32(1e3)
    neg[::11] *= np.float32(1e-30)  # subnormal-range magnitudes keep their order too
    return {"n1": uniform01(1, 3).reshape(1, 3), "n16": uniform01(2, 48).reshape(16, 3),
            "n17": uniform01(2, 51).reshape(17, 3), "n33": uniform01(4, 99).reshape(33, 3),
            "n1000": uniform01(5, 3000).reshape(1000, 3), "dup": dup, "neg": neg.astype(np.float32),
            "n270213": uniform01(6, 3 * 270213).reshape(-1, 3)}


@pytest.mark.parametrize("case", ["n1", "n16", "n17", "n33", "n1000", "dup", "neg", "n270213"])
@pytest.mark.parametrize("wide_len", [None, 64, 1 << 30])
@pytest.mark.parametrize("sort", [None, "rocprim"])
def test_kdtree_device_build_is_bit_identical_to_host_build(ctx, diag_ctx, monkeypatch, case, wide_len, sort):
    """The device build (hand-written device-wide radix sort for the long ranges + LDS bitonic sort on
    (key, position) words for the short ones; rocPRIM's stable sorts as the cross-check) lays out exactly the tree
    of the host build (std::stable_sort per node = R3dTree::new, src/kdtree.rs:28-58): split table and every leaf
    slot."""
    db = _kd_cases()[case]
    host = _build(diag_ctx, db, "host", monkeypatch)  # (the host build and the knobs exist in the diagnostics build only)
    dev = _build(diag_ctx, db, "device", monkeypatch, wide_len, sort)
    assert dev.stats() == host.stats()
    hs, hl = host.download()
    ds, dl = dev.download()
    assert np.array_equal(ds, hs)
    assert np.array_equal(dl, hl)
    if wide_len is None and sort is None:  # and the PRODUCT library's build (no knobs) is that tree too
        prod = R3dTree.new(ctx, db)
        ps, pl = prod.download()
        assert prod.stats() == host.stats() and np.array_equal(ps, hs) and np.array_equal(pl, hl)


@pytest.mark.parametrize("case", ["n1000", "dup", "n270213"])
def test_kdtree_device_build_with_the_separate_scan_kernel(diag_ctx, monkeypatch, case):
    ctx = diag_ctx
    """Above 4M keys the radix passes keep their digit-major table and one-block scan kernel instead of deriving the
    offsets inside the scatter blocks; A3D_KDTREE_SCAN=unfused forces that form at test sizes.  Same tree."""
    db = _kd_cases()[case]
    host = _build(ctx, db, "host", monkeypatch)
    monkeypatch.setenv("A3D_KDTREE_SCAN", "unfused")
    dev = _build(ctx, db, "device", monkeypatch, 64)
    monkeypatch.delenv("A3D_KDTREE_SCAN")
    hs, hl = host.download()
    ds, dl = dev.download()
    assert dev.stats() == host.stats() and np.array_equal(ds, hs) and np.array_equal(dl, hl)


def test_kdtree_device_build_nan_rules(ctx, monkeypatch):
    """NaN in a coordinate that gets compared is the reference's panic; one that never is compared is not."""
    for mode in ("host", "device"):
        db = uniform01(1, 300).reshape(100, 3)
        db[50, 2] = np.nan  # 100 points: depth 0 sorts x (100), depth 1 y (50), depth 2 z (25) -> compared
        with pytest.raises(A3dError) as e:
            _build(ctx, db, mode, monkeypatch)
        assert e.value.status == 5
        db = uniform01(1, 90).reshape(30, 3)
        db[3, 2] = np.nan  # 30 points: x at depth 0, leaves of 15 afterwards: z never compared
        t = _build(ctx, db, mode, monkeypatch)
        assert t.stats() == (2, 1, 1)
