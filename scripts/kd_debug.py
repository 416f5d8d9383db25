"""Debug aid: one kd-tree case built by the product / diagnostics library against the host build; prints where they differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from align3d_amd import Context, R3dTree, _abi
from data_util import uniform01
case = sys.argv[1] if len(sys.argv) > 1 else "neg"
if case == "neg":
    db = (uniform01(3, 3 * 70001).reshape(-1, 3) - 0.5) * np.float32(1e3)
    db[::11] *= np.float32(1e-30)
    db = db.astype(np.float32)
elif case == "negbig":
    db = ((uniform01(3, 3 * 70001).reshape(-1, 3) - 0.5) * np.float32(1e3)).astype(np.float32)
elif case == "negtiny1":
    db = (uniform01(3, 3 * 70001).reshape(-1, 3) - 0.5) * np.float32(1e3)
    db[::11, 0] *= np.float32(1e-30)
    db = db.astype(np.float32)
elif case.startswith("ties"):  # ties<axis>:<n>: unique random coordinates except 400 planted pairs equal along one axis
    ax, n = int(case[4]), int(case.split(":")[1])
    rng = np.random.default_rng(5)
    db = np.stack([rng.permutation(n * 4)[:n] for _ in range(3)], axis=1).astype(np.float32) / np.float32(n * 4)
    for i in range(400):
        db[2 * i + 1, ax] = db[2 * i, ax]
else:
    n = int(case)
    db = uniform01(6, 3 * n).reshape(-1, 3)
diag = Context(0, library=_abi.DIAG_LIB_PATH)
os.environ["A3D_KDTREE_BUILD"] = "host"
host = R3dTree.new(diag, db)
del os.environ["A3D_KDTREE_BUILD"]
hs, hl = host.download()
for rep in range(3):
    dev = R3dTree.new(diag, db)
    ds, dl = dev.download()
    bad_s = np.nonzero(ds != hs)[0]
    bad_l = np.nonzero((dl != hl).any(axis=1))[0]
    print(f"rep {rep}: path {dev.build_path()} stats {dev.stats()} split mismatches {len(bad_s)} (first {bad_s[:8]}), leaf-slot mismatches {len(bad_l)} (first {bad_l[:8]})")
    if len(bad_s):
        i = bad_s[0]
        lvl = int(np.floor(np.log2(i + 1)))
        print("   first bad split: heap", i, "level", lvl, "node", i + 1 - (1 << lvl), "dev", ds[i:i+1].view(np.float32), "host", hs[i:i+1].view(np.float32))
    dev.free()
