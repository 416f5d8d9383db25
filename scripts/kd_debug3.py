import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from align3d_amd import Context, R3dTree, _abi
n = 100000
rng = np.random.default_rng(5)
db = np.stack([rng.permutation(n * 4)[:n] for _ in range(3)], axis=1).astype(np.float32) / np.float32(n * 4)
for i in range(400):
    db[2 * i + 1, 0] = db[2 * i, 0]
diag = Context(0, library=_abi.DIAG_LIB_PATH)
os.environ["A3D_KDTREE_BUILD"] = "host"
host = R3dTree.new(diag, db)
del os.environ["A3D_KDTREE_BUILD"]
hs, hl = host.download()
dev = R3dTree.new(diag, db)
ds, dl = dev.download()
D = host.stats()[2]
inf = np.uint32(0x7f800000)
def slot_of(leaves):
    s = np.full(n, -1, np.int64)
    used = leaves[:, 0] != inf
    s[leaves[used, 3]] = np.nonzero(used)[0]
    return s, used
(sh, uh), (sd, ud) = slot_of(hl), slot_of(dl)
print("device used slots", int(ud.sum()), "host", int(uh.sum()))
missing = np.nonzero(sd < 0)[0]
idxs, counts = np.unique(dl[ud, 3], return_counts=True)
dups = idxs[counts > 1]
print("missing", missing[:20], "duplicated", dups[:20])
node6 = (sh // 16) >> (D - 6)
for p in list(missing[:6]):
    mates = np.nonzero((node6 == node6[p]) & (db[:, 0] == db[p, 0]))[0]
    print(f"missing point {p}: level-6 node {node6[p]}, x={db[p,0]!r}, points of that node with the same x: {mates}, pair partner {p ^ 1} x={db[p ^ 1, 0]!r} node {node6[p ^ 1]}")
for p in list(dups[:6]):
    p = int(p)
    mates = np.nonzero((node6 == node6[p]) & (db[:, 0] == db[p, 0]))[0]
    print(f"duplicated point {p}: level-6 node {node6[p]}, x={db[p,0]!r}, same-x points of that node: {mates}")
