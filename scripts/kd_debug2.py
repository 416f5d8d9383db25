"""Debug aid: where do the device and host trees diverge, and are ties involved?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from align3d_amd import Context, R3dTree, _abi
from data_util import uniform01
n = int(sys.argv[1])
db = uniform01(6, 3 * n).reshape(-1, 3)
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
    return s
sh, sd = slot_of(hl), slot_of(dl)
moved = np.nonzero(sh != sd)[0]
print("points in different slots:", len(moved), "missing on device:", int((sd < 0).sum()))
for p in moved[:12]:
    lh, ld = sh[p] // 16, sd[p] // 16
    x = int(lh) ^ int(ld)
    lvl = D - x.bit_length() if x else None
    msg = f"point {p}: host slot {sh[p]} dev slot {sd[p]}"
    if lvl is not None:
        a = lvl % 3
        node = int(lh) >> (D - lvl)
        heap = (1 << lvl) - 1 + node
        sv = hs[heap:heap+1].view(np.float32)[0]
        ties = int((db[:, a] == db[p, a]).sum())
        msg += f"; paths diverge at level {lvl} (axis {a}), host split {sv!r} dev split {ds[heap:heap+1].view(np.float32)[0]!r}, point's coord {db[p, a]!r}, points with that coord: {ties}"
    else:
        msg += "; same leaf, different slot"
    print(msg)
