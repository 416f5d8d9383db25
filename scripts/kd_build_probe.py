"""Times R3dTree::new (host build vs device build) at BASELINE's 500k-point shape."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from align3d_amd import Context, R3dTree  # noqa: E402
from data_util import uniform01  # noqa: E402

ctx = Context(0)
for n in (500_000, 270_213, 50_000):
    db = uniform01(10, 3 * n).reshape(n, 3)
    for mode, wl, sort in (("host", None, None), ("device", None, None), ("device", None, "rocprim"), ("device", 512, None),
                           ("device", 1024, None)):
        os.environ["A3D_KDTREE_BUILD"] = mode
        if sort:
            os.environ["A3D_KDTREE_SORT"] = sort
        else:
            os.environ.pop("A3D_KDTREE_SORT", None)
        if wl is None:
            os.environ.pop("A3D_KDTREE_WIDE_LEN", None)
        else:
            os.environ["A3D_KDTREE_WIDE_LEN"] = str(wl)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            t = R3dTree.new(ctx, db)
            best = min(best, time.perf_counter() - t0)
            t.free()
        print(f"n={n} build={mode} sort={sort or 'own'} wide_len={wl}: {best * 1e3:.2f} ms", flush=True)
