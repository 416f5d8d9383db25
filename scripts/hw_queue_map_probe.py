"""Which HIP streams of one process share a hardware queue?  N streams created in order; a spin kernel on stream 0 and
one on stream k at the same time: 1x the spin time = they ran concurrently, 2x = one queue serialised them.
(GPU_MAX_HW_QUEUES from the environment is printed; tuning probe, uses torch only for streams and a sleep kernel.)"""
import os, sys, time
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
streams = [torch.cuda.Stream() for _ in range(N)]
cycles = 20_000_000
def both(i, j):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(streams[i]): torch.cuda._sleep(cycles)
    if j is not None:
        with torch.cuda.stream(streams[j]): torch.cuda._sleep(cycles)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
both(0, 1)
one = both(0, None)
print(f"one spin: {one:.2f} ms")
print("stream 0 with k:", " ".join(f"{k}:{both(0, k) / one:.1f}x" for k in range(1, N)))
print("stream 1 with k:", " ".join(f"{k}:{both(1, k) / one:.1f}x" for k in range(2, N)))
