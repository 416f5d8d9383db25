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
