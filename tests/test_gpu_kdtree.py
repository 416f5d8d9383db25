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


_KD_KNOBS = ("A3D_KDTREE_BUILD", "A3D_KDTREE_SORT", "A3D_KDTREE_WIDE_LEN", "A3D_KDTREE_NARROW_LEN", "A3D_KDTREE_SCAN",
             "A3D_KDTREE_SORTNET", "A3D_KDTREE_BUCKETS", "A3D_KDTREE_WIDE_PLACE", "A3D_KDTREE_WIDE_CAP", "A3D_KDTREE_FUSE")

# The device builds (diagnostics build: knobs).  "select" is what the product library runs: the selection build
# (kdtree_select.hip); NARROW_LEN makes its wide-level kernels run at test sizes; "sorted" is the sorting build (the
# cross-check, diagnostics build only: hand-written radix + bitonic sorts, or rocPRIM's), WIDE_LEN moves its wide / narrow border.
# WIDE_PLACE=1 adds the chip-wide placement launches for oversized median buckets (what a context switches to by itself
# once a cloud had one; 0 keeps them off whatever the context has seen), WIDE_CAP makes buckets count as oversized at test sizes.
_KD_BUILDS = {
    "select": {"A3D_KDTREE_WIDE_PLACE": "0"},
    "select_narrow32": {"A3D_KDTREE_NARROW_LEN": "32", "A3D_KDTREE_WIDE_PLACE": "0"},
    "select_narrow64": {"A3D_KDTREE_NARROW_LEN": "64", "A3D_KDTREE_WIDE_PLACE": "0"},
    "select_narrow512": {"A3D_KDTREE_NARROW_LEN": "512", "A3D_KDTREE_WIDE_PLACE": "0"},
    # the resolve step as a launch of its own (the product runs it in the last block of the node's split launch: round 6)
    "select_unfused": {"A3D_KDTREE_FUSE": "0", "A3D_KDTREE_WIDE_PLACE": "0"},
    "select_unfused_narrow32": {"A3D_KDTREE_FUSE": "0", "A3D_KDTREE_NARROW_LEN": "32", "A3D_KDTREE_WIDE_PLACE": "0"},
    "select_lds": {"A3D_KDTREE_SORTNET": "lds"},  # the in-block levels by the sorting network with its words in LDS (the product keeps them in registers: round 6)
    "select_inblock": {"A3D_KDTREE_SORTNET": "select"},  # ... by selection inside the block (slower: a cross-check)
    "select_inblock_narrow64": {"A3D_KDTREE_SORTNET": "select", "A3D_KDTREE_NARROW_LEN": "64", "A3D_KDTREE_WIDE_PLACE": "0"},
    "select_place": {"A3D_KDTREE_WIDE_PLACE": "1"},
    "select_place_cap64": {"A3D_KDTREE_WIDE_PLACE": "1", "A3D_KDTREE_WIDE_CAP": "64"},
    "select_place_cap16_narrow64": {"A3D_KDTREE_WIDE_PLACE": "1", "A3D_KDTREE_WIDE_CAP": "16", "A3D_KDTREE_NARROW_LEN": "64"},
    "select_place_cap256_narrow512": {"A3D_KDTREE_WIDE_PLACE": "1", "A3D_KDTREE_WIDE_CAP": "256", "A3D_KDTREE_NARROW_LEN": "512",
                                      "A3D_KDTREE_BUCKETS": "64"},
    "sorted": {"A3D_KDTREE_BUILD": "sorted"},
    "sorted_wide64": {"A3D_KDTREE_BUILD": "sorted", "A3D_KDTREE_WIDE_LEN": "64"},
    "sorted_all_wide": {"A3D_KDTREE_BUILD": "sorted", "A3D_KDTREE_WIDE_LEN": str(1 << 30)},
    "sorted_rocprim": {"A3D_KDTREE_BUILD": "sorted", "A3D_KDTREE_SORT": "rocprim"},
    "sorted_rocprim_wide64": {"A3D_KDTREE_BUILD": "sorted", "A3D_KDTREE_SORT": "rocprim", "A3D_KDTREE_WIDE_LEN": "64"},
}


def _build(ctx, db, env, monkeypatch, device_pointer=False):
    for k in _KD_KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    try:
        if device_pointer:
            d = ctx.to_device(np.ascontiguousarray(db, np.float32))
            try:
                return R3dTree.new_device(ctx, d, len(db))
            finally:
                ctx.free(d)  # (read during the call only)
        return R3dTree.new(ctx, db)
    finally:
        for k in env:
            monkeypatch.delenv(k, raising=False)


def _kd_cases():
    rng = np.random.default_rng(17)
    dup = rng.integers(-3, 4, size=(20000, 3)).astype(np.float32)
    dup[::7, 0] = -0.0
    dup[::5, 1] = 0.0
    neg = (uniform01(3, 3 * 70001).reshape(-1, 3) - 0.5) * np.float32(1e3)
    neg[::11] *= np.float32(1e-30)  # subnormal-range magnitudes keep their order too
    # clustered: a few tight blobs + far outliers (buckets over the bounding box are very unevenly filled)
    blobs = np.concatenate([rng.normal(c, 0.01, size=(15000, 3)) for c in ((0, 0, 0), (1, 0.2, -3), (1.001, 0.2, -3), (50, 50, 50))]
                           + [rng.uniform(-1e4, 1e4, size=(37, 3))]).astype(np.float32)
    rng.shuffle(blobs)
    # exactly equal points in bulk (every key of L_d ties: the original index decides) beside distinct ones
    twins = np.repeat(uniform01(8, 3 * 3000).reshape(-1, 3), 5, axis=0)
    rng.shuffle(twins)
    # a cloud back-projected from a depth image: z quantised to 1 / 5000 m, half of the pixels on one fronto-parallel plane
    # (tens of thousands of points with the same z), x and y on the pixel lattice scaled by z (round 5: such a cloud cost
    # the in-block entry level 2 ms in run-length-squared counting before it sorted long runs on 128-bit words)
    vv, uu = np.mgrid[0:200, 0:300]
    zz = np.where((uu + vv) % 2 == 0, 2.0, np.round(rng.uniform(1.0, 4.0, size=uu.shape) * 5000) / 5000).astype(np.float32)
    depth_cloud = np.stack([(uu - 150) * zz / 300, (vv - 100) * zz / 300, zz], axis=-1).reshape(-1, 3).astype(np.float32)
    return {"n1": uniform01(1, 3).reshape(1, 3), "n16": uniform01(2, 48).reshape(16, 3), "depth_cloud": depth_cloud,
            "n17": uniform01(2, 51).reshape(17, 3), "n33": uniform01(4, 99).reshape(33, 3),
            "n1000": uniform01(5, 3000).reshape(1000, 3), "n2049": uniform01(7, 3 * 2049).reshape(-1, 3),
            "n4099": uniform01(9, 3 * 4099).reshape(-1, 3), "dup": dup, "neg": neg.astype(np.float32),
            "blobs": blobs, "twins": twins, "n270213": uniform01(6, 3 * 270213).reshape(-1, 3)}


@pytest.mark.parametrize("case", ["n1", "n16", "n17", "n33", "n1000", "n2049", "n4099", "dup", "neg", "blobs", "twins", "depth_cloud",
                                  "n270213"])
@pytest.mark.parametrize("build", list(_KD_BUILDS))
def test_kdtree_device_build_is_bit_identical_to_host_build(ctx, diag_ctx, monkeypatch, case, build):
    """Every device build lays out exactly the tree of the host build (std::stable_sort per node = R3dTree::new,
    src/kdtree.rs:28-58): split table and every leaf slot.  The selection build (per level a bucket selection of the
    median under the closed-form order L_d + an unordered partition; the last levels sorted in LDS) against the sorting
    build (one stable sort per level) against rocPRIM's sorts against the host."""
    db = _kd_cases()[case]
    if case == "n270213" and build not in ("select", "select_narrow512", "select_unfused", "select_lds", "select_inblock", "select_place_cap64", "sorted", "sorted_rocprim"):
        pytest.skip("the large case runs on the default borders only")
    host = _build(diag_ctx, db, {"A3D_KDTREE_BUILD": "host"}, monkeypatch)
    dev = _build(diag_ctx, db, _KD_BUILDS[build], monkeypatch)
    assert host.build_path() == 0
    if build.startswith("sorted"):
        assert dev.build_path() == 2
    assert dev.stats() == host.stats()
    hs, hl = host.download()
    ds, dl = dev.download()
    assert np.array_equal(ds, hs)
    assert np.array_equal(dl, hl)
    if build == "select":  # and the PRODUCT library's build (no knobs) is that tree too, from host and from device memory
        for dp in (False, True):
            prod = _build(ctx, db, {}, monkeypatch, device_pointer=dp)
            ps, pl = prod.download()
            assert prod.stats() == host.stats() and np.array_equal(ps, hs) and np.array_equal(pl, hl)
            assert dev.build_path() == 1 and prod.build_path() in (1, 3)  # (3: the context has met an oversized bucket before)


@pytest.mark.parametrize("n", [16384, 20000, 40000, 100000])
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_kdtree_pairs_of_equal_coordinates_at_every_entry_axis(ctx, diag_ctx, monkeypatch, n, axis):
    """Otherwise distinct coordinates with 400 planted pairs equal along ONE axis.  The in-block kernel takes over at
    depth 3 / 4 / 5 / 6 for these sizes (entry axis x / y / z / x), 6 / 7 / 7 / 8 with the border at 512: every
    combination of tie axis and entry axis, where a run of equal keys has to be put into the order of the older keys.
    (Round 5: the x-entry case was broken by a miscompiled coordinate select and no test had equal x keys at such a
    depth: found by the full-size run.)"""
    rng = np.random.default_rng(5 + n + axis)
    db = np.stack([rng.permutation(n * 4)[:n] for _ in range(3)], axis=1).astype(np.float32) / np.float32(n * 4)
    for i in range(400):
        db[2 * i + 1, axis] = db[2 * i, axis]
    host = _build(diag_ctx, db, {"A3D_KDTREE_BUILD": "host"}, monkeypatch)
    hs, hl = host.download()
    for c, env in ((ctx, {}), (diag_ctx, _KD_BUILDS["select_lds"]), (diag_ctx, _KD_BUILDS["select_unfused"]), (diag_ctx, _KD_BUILDS["select_inblock"]), (diag_ctx, _KD_BUILDS["select_narrow512"]),
                   (diag_ctx, {"A3D_KDTREE_NARROW_LEN": "1024", "A3D_KDTREE_BUCKETS": "256"})):
        t = _build(c, db, env, monkeypatch)
        s_, l_ = t.download()
        assert t.stats() == host.stats() and np.array_equal(s_, hs) and np.array_equal(l_, hl), env


def test_kdtree_selection_build_at_three_million_points(ctx, diag_ctx, monkeypatch):
    """Eleven wide levels (2048 nodes resolved per launch at the last of them), 2048 in-block ranges, histogram tables of
    the deeper levels larger than the top one's: the sizes the 500 k benchmark does not reach.  Same tree as the host build."""
    n = 3_000_017
    db = uniform01(41, 3 * n).reshape(-1, 3)
    db[::977, 2] = db[5, 2]  # ~3000 points share one z exactly
    host = _build(diag_ctx, db, {"A3D_KDTREE_BUILD": "host"}, monkeypatch)
    hs, hl = host.download()
    t = _build(ctx, db, {}, monkeypatch)
    s_, l_ = t.download()
    assert t.build_path() in (1, 3) and t.stats() == host.stats()
    assert np.array_equal(s_, hs) and np.array_equal(l_, hl)
    print(f"[kd-tree build, {n} points] {t.build_ms():.3f} ms of launches")
    # and with the chip-wide placement launches on all ten levels that can hold an oversized bucket at this size (the
    # ~3000 equal z make one at every level down to ranges of 256 points with the threshold lowered to 256)
    for env in ({"A3D_KDTREE_WIDE_PLACE": "1"}, {"A3D_KDTREE_WIDE_PLACE": "1", "A3D_KDTREE_WIDE_CAP": "256"}):
        t = _build(diag_ctx, db, env, monkeypatch)
        s_, l_ = t.download()
        assert t.build_path() == 3 and np.array_equal(s_, hs) and np.array_equal(l_, hl), env
        t.free()


def test_kdtree_selection_build_finishes_degenerate_clouds(ctx, diag_ctx, monkeypatch):
    """Clouds whose median bucket is (nearly) the whole node — one coordinate constant, a handful of distinct values,
    every point the same, tens of thousands of values within 1e-27 of zero — cost the resolve block more narrowing rounds
    (from global memory while the candidates exceed its LDS, component of L_d by component), nothing else: same tree."""
    rng = np.random.default_rng(23)
    plane = rng.uniform(-1, 1, size=(60000, 3)).astype(np.float32)
    plane[:, 1] = 0.25  # y constant: level 1 cannot be bucketed at all
    ties = rng.integers(0, 3, size=(50000, 3)).astype(np.float32)  # ~16 k equal keys per node at the top
    same = np.tile(np.array([[1.5, -2.0, 0.0]], np.float32), (30000, 1))  # only the original index tells points apart
    tiny = (uniform01(3, 3 * 70001).reshape(-1, 3) - 0.5) * np.float32(1e3)
    tiny[::3] *= np.float32(1e-30)  # a third of the cloud within 5e-28 of the origin, all distinct
    # a depth image's cloud at the benchmark's scale: a fronto-parallel wall (one z for 60 % of the pixels, one y per row)
    vv, uu = np.mgrid[0:400, 0:600]
    zz = np.where(uu < 360, 3.0, np.round(rng.uniform(1.0, 4.0, size=uu.shape) * 5000) / 5000).astype(np.float32)
    wall = np.stack([(uu - 300) * zz / 500, (vv - 200) * zz / 500, zz], axis=-1).reshape(-1, 3).astype(np.float32)
    for db in (plane, ties, same, tiny.astype(np.float32), wall):
        host = _build(diag_ctx, db, {"A3D_KDTREE_BUILD": "host"}, monkeypatch)
        hs, hl = host.download()
        for c, env in ((ctx, {}), (diag_ctx, {"A3D_KDTREE_NARROW_LEN": "64", "A3D_KDTREE_WIDE_PLACE": "0"}),
                       (diag_ctx, {"A3D_KDTREE_WIDE_PLACE": "0"}), (diag_ctx, {"A3D_KDTREE_WIDE_PLACE": "1"}),
                       (diag_ctx, {"A3D_KDTREE_WIDE_PLACE": "1", "A3D_KDTREE_WIDE_CAP": "100", "A3D_KDTREE_NARROW_LEN": "64"}),
                       (diag_ctx, {"A3D_KDTREE_WIDE_PLACE": "1", "A3D_KDTREE_WIDE_CAP": "1000", "A3D_KDTREE_BUCKETS": "128"})):
            t = _build(c, db, env, monkeypatch)
            assert t.build_path() == (3 if env.get("A3D_KDTREE_WIDE_PLACE") == "1" else 1) or c is ctx
            s_, l_ = t.download()
            assert t.stats() == host.stats() and np.array_equal(s_, hs) and np.array_equal(l_, hl), env
        # the same five clouds against the ORACLE's node-graph tree (src/kdtree.rs:28-105), not only
        # against this library's host build: tree shape, and `nearest` of every point of the cloud plus 20 000 random
        # queries over (and beyond) its bounding box — indices and squared distances bit for bit
        ref = O.KdTree(db)
        assert ref.status == 0
        lo, hi = db.min(axis=0), db.max(axis=0)
        span = np.maximum(hi - lo, np.float32(1e-3))
        rq = (lo - 0.25 * span + uniform01(77, 3 * 20000).reshape(-1, 3) * 1.5 * span).astype(np.float32)
        queries = np.concatenate([db, rq])
        ri, rd = ref.nearest(queries)
        t = R3dTree.new(ctx, db)
        assert t.stats() == ref.stats()
        gi, gd = t.nearest(queries)
        assert np.array_equal(gi, ri) and np.array_equal(gd.view(np.uint32), rd.view(np.uint32))
        t.free()


def test_kdtree_context_switches_to_wide_placement_after_an_oversized_bucket(diag_ctx, monkeypatch):
    """The product's policy (round 6): a new context's builds have sel_place_kernel's launches (its first depth-image cloud
    does not pay a lone block's streaming rounds), drop them after four builds in a row without a median bucket larger than
    the resolve block's LDS, and take them up again once a cloud had one.  Same tree whatever the path."""
    from align3d_amd import Context

    for k in _KD_KNOBS:
        monkeypatch.delenv(k, raising=False)
    fresh = Context(0)
    benign = uniform01(12, 3 * 50000).reshape(-1, 3)
    wall = benign.copy()
    wall[:30000, 2] = 0.5
    host_b = _build(diag_ctx, benign, {"A3D_KDTREE_BUILD": "host"}, monkeypatch).download()
    host_w = _build(diag_ctx, wall, {"A3D_KDTREE_BUILD": "host"}, monkeypatch).download()
    paths = []
    for db, want in ((benign, host_b),) * 5 + ((wall, host_w), (wall, host_w), (benign, host_b)):
        t = R3dTree.new(fresh, db)
        got = t.download()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        paths.append(t.build_path())
        t.free()
    assert paths == [3, 3, 3, 3, 1, 1, 3, 3]  # on for a new context, off after four quiet builds, on again after the wall
    fresh.close()


@pytest.mark.parametrize("case", ["n1000", "dup", "n270213"])
def test_kdtree_device_build_with_the_separate_scan_kernel(diag_ctx, monkeypatch, case):
    ctx = diag_ctx
    """Above 4M keys the sorting build's radix passes keep their digit-major table and one-block scan kernel instead of
    deriving the offsets inside the scatter blocks; A3D_KDTREE_SCAN=unfused forces that form at test sizes.  Same tree."""
    db = _kd_cases()[case]
    host = _build(ctx, db, {"A3D_KDTREE_BUILD": "host"}, monkeypatch)
    dev = _build(ctx, db, {"A3D_KDTREE_BUILD": "sorted", "A3D_KDTREE_WIDE_LEN": "64", "A3D_KDTREE_SCAN": "unfused"}, monkeypatch)
    hs, hl = host.download()
    ds, dl = dev.download()
    assert dev.stats() == host.stats() and np.array_equal(ds, hs) and np.array_equal(dl, hl)


def test_kdtree_device_build_nan_rules(ctx, diag_ctx, monkeypatch):
    """NaN in a coordinate that gets compared is the reference's panic; one that never is compared is not."""
    for c, env in ((diag_ctx, {"A3D_KDTREE_BUILD": "host"}), (ctx, {}), (diag_ctx, {"A3D_KDTREE_BUILD": "sorted"}),
                   (diag_ctx, {"A3D_KDTREE_NARROW_LEN": "32"})):
        db = uniform01(1, 300).reshape(100, 3)
        db[50, 2] = np.nan  # 100 points: depth 0 sorts x (100), depth 1 y (50), depth 2 z (25) -> compared
        with pytest.raises(A3dError) as e:
            _build(c, db, env, monkeypatch)
        assert e.value.status == 5
        db = uniform01(1, 90).reshape(30, 3)
        db[3, 2] = np.nan  # 30 points: x at depth 0, leaves of 15 afterwards: z never compared
        t = _build(c, db, env, monkeypatch)
        assert t.stats() == (2, 1, 1)
        big = uniform01(2, 3 * 9000).reshape(-1, 3)
        big[4321, 1] = np.nan  # wide levels of the selection build: y is compared at depth 1
        with pytest.raises(A3dError) as e:
            _build(c, big, env, monkeypatch)
        assert e.value.status == 5


def test_icp_from_resident_clouds_gives_the_same_bits(ctx):
    """a3d_pcl_icp_new_device / a3d_pcl_icp_align_device (clouds already in HBM: PointCloud's layout, src/pointcloud.rs:8-12)
    against the host-pointer forms: the same pose bits."""
    from align3d_amd import DevicePointCloud, Icp, IcpParams, PointCloud
    from gpu_util import to_range_image

    tc = PointCloud.from_range_image(to_range_image(oracle_frame("sample1", 0, True)))
    sc = PointCloud.from_range_image(to_range_image(oracle_frame("sample1", 1, True)))
    prm = IcpParams(max_iterations=5)
    ref = Icp.new(ctx, prm, tc).align(sc)
    dt, ds = DevicePointCloud(ctx, tc), DevicePointCloud(ctx, sc)
    icp = Icp.new(ctx, prm, dt)
    dt.free()  # the target's arrays are read during Icp::new only
    got = icp.align(ds)
    again = icp.align(sc)  # a host-pointer source on an Icp made from a resident target
    for T in (got, again):
        assert np.array_equal(np.concatenate([T.t, T.q]).view(np.uint32), np.concatenate([ref.t, ref.q]).view(np.uint32))
    ds.free()


@pytest.mark.gpu
def test_kdtree_product_build_on_random_sizes_and_distributions(ctx, diag_ctx, monkeypatch):
    """The product's build (round 6: resolve step and root plan inside the split / histogram launches' last blocks, in-block
    network in registers) against the host build over sizes that put the wide / in-block border, the last block of a node
    and the placement switch everywhere: uniform, quantised (a depth image's z) and clustered clouds, twice on the same
    context (the second build reuses the first one's workspace: tickets and counters must have been left clean)."""
    rng = np.random.default_rng(2026)
    sizes = [2048, 2049, 4095, 4096, 4097, 6143, 8193, 12289] + [int(v) for v in rng.integers(3000, 400000, size=25)]
    for k in _KD_KNOBS:
        monkeypatch.delenv(k, raising=False)
    for i, n in enumerate(sizes):
        kind = i % 3
        if kind == 0:
            db = rng.random((n, 3), dtype=np.float32)
        elif kind == 1:  # quantised z with a wall, x / y on a lattice scaled by z
            z = np.where(rng.random(n) < 0.4, 2.0, np.round(rng.uniform(1.0, 4.0, n) * 500) / 500).astype(np.float32)
            db = np.stack([rng.integers(-300, 300, n) * z / 300, rng.integers(-200, 200, n) * z / 300, z], axis=-1).astype(np.float32)
        else:
            db = np.concatenate([rng.normal(c, 0.02, size=(n // 3 + 1, 3)) for c in ((0, 0, 0), (1, 1, 1), (1.01, 1, 1))])[:n].astype(np.float32)
            rng.shuffle(db)
        host = _build(diag_ctx, db, {"A3D_KDTREE_BUILD": "host"}, monkeypatch)
        hs, hl = host.download()
        for _ in range(2):
            t = R3dTree.new(ctx, db)
            s_, l_ = t.download()
            assert t.stats() == host.stats() and np.array_equal(s_, hs) and np.array_equal(l_, hl), (n, kind)
            t.free()
        host.free()
