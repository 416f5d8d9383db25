"""Frame-build throughput (u16 depth + RGB -> resident 3-level pyramid) with N builder threads, each on its own
context (stream + scratch) of the same GPU."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import BilateralFilter, Context, RangeImageBuilder, SyntheticDataset

ds = SyntheticDataset(7, 4)
frames = [ds.get(i) for i in range(4)]
for n_threads in (1, 2, 3, 4, 6, 8):
    ctxs = [Context(0) for _ in range(n_threads)]
    builders = [RangeImageBuilder(c).with_bilateral_filter(BilateralFilter.default()) for c in ctxs]
    per_thread = 60
    def work(k):
        b = builders[k]
        for i in range(per_thread):
            pyr = b.build_device(*frames[i % 4])
            for lv in pyr:
                lv.free()
    for k in range(n_threads):  # warm each context (scratch, arena pool)
        for lv in builders[k].build_device(*frames[0]):
            lv.free()
    ts = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    t0 = time.perf_counter()
    [t.start() for t in ts]; [t.join() for t in ts]
    dt = time.perf_counter() - t0
    print(f"{n_threads} builder thread(s): {n_threads * per_thread / dt:.0f} frames/s ({dt / (n_threads * per_thread) * 1e3:.3f} ms per frame)", flush=True)
    for c in ctxs:
        c.close()
