"""Would a step that keeps its pairs' level-0 arrays in the Infinity Cache pay?  K independent batches of Pb distinct
pairs each run concurrently (one context = one set of streams per batch, one host thread each), so that K x Pb pairs are
in flight and the batches drift out of phase (one's launch gaps under the others' pixel passes): aggregate pairs/s.
  python3 scripts/mall_probe2.py 1x64 2x8 3x5 4x4 2x16 4x16"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
main = Context(0)
prm = MsIcpParams.repeat(3, IcpParams.default())
pyr, _, _ = bench.build_stream_pyramids(main, 1000, 128, 640, 480)
host = [[lv.download() for lv in p] for p in pyr]
for spec in sys.argv[1:] or ["1x64", "2x8"]:
    K, Pb = (int(x) for x in spec.split("x"))
    ctxs = [main] + [Context(0, pair=False, main_slot=k % 3) for k in range(1, K)]
    batches = []
    for k, c in enumerate(ctxs):
        idx = [(k * Pb + p) % 64 for p in range(Pb)]
        if c is main:
            t, s = [pyr[2 * i] for i in idx], [pyr[2 * i + 1] for i in idx]
        else:  # images live on the context that uses them
            t = [[lv.device(c) for lv in host[2 * i]] for i in idx]
            s = [[lv.device(c) for lv in host[2 * i + 1]] for i in idx]
        batches.append(MultiscaleAlignBatch(c, prm, t, s))
    steps = max(20, 1920 // (K * Pb))
    def run(k, n):
        for _ in range(n):
            batches[k].enqueue()
        ctxs[k].synchronize()
    for k in range(K):
        run(k, 5)
    reps = []
    for _ in range(5):
        th = [threading.Thread(target=run, args=(k, steps)) for k in range(K)]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        reps.append(K * Pb * steps / (time.perf_counter() - t0))
    print(f"{K} batches x {Pb} pairs ({K * Pb * 13.2:.0f} MB of level-0 arrays in flight): {np.median(reps) / 1e3:.1f} k pairs/s  {[round(r / 1e3, 1) for r in reps]}", flush=True)
    for b in batches: b.free()
    for c in ctxs[1:]: c.close()
