// What does a dependent same-stream kernel boundary cost on this chip, and which property of a dispatch moves it?
// Chains of N launches of trivial kernels, launched the way the library launches image_icp_head_kernel (hipLaunchKernelGGL
// on a non-blocking stream created with a priority), timed by hipEvents around the chain -> us per launch.  Variants:
// blocks per grid (1 / 200 / 2048), kernarg size (8 B / 104 B as HeadArgs by value / 1 KB), static LDS (0 / 5.5 KB /
// 64 KB), a body that reads what the predecessor wrote (the real dependency), stream kind.  Measurement aid.
//   hipcc --offload-arch=gfx950 -O3 scripts/boundary_probe.hip -o scripts/boundary_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Arg104 { unsigned w[24]; };   // + an 8-byte pointer = 104 B, the kernarg size of image_icp_head_kernel
struct Arg1k { unsigned w[254]; };

__global__ void __launch_bounds__(256) k_empty(unsigned* p) { if (p == nullptr) p[0] = 1; }
__global__ void __launch_bounds__(256) k_arg104(unsigned* p, Arg104 a) { if (a.w[3] == 12345u) p[0] = a.w[7]; }
__global__ void __launch_bounds__(256) k_arg1k(unsigned* p, Arg1k a) { if (a.w[3] == 12345u) p[0] = a.w[7]; }
template <int WORDS>
__global__ void __launch_bounds__(256) k_lds(unsigned* p) {
  __shared__ unsigned s[WORDS];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (s[(threadIdx.x + 1) & 255] == 12345u) p[0] = 1;
}
// holds the stream busy while the host enqueues the chain behind it: the chain then runs at the GPU's own pace
__global__ void k_spin(unsigned* p, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (p == nullptr) p[0] = 1;
}
// the real dependency: every block reads what block 0 of the predecessor wrote, block 0 writes the next value
__global__ void __launch_bounds__(256) k_dep(unsigned* p, unsigned seq) {
  const unsigned v = __builtin_nontemporal_load(p + (seq & 1u));
  if (blockIdx.x == 0 && threadIdx.x == 0) p[(seq + 1u) & 1u] = v + 1u;
}
// ~head-sized work: 200 partial rows of 58 floats summed by every block, like head_sum_and_advance
__global__ void __launch_bounds__(256) k_headlike(const float* __restrict__ part, float* __restrict__ out, unsigned tiles) {
  __shared__ float s[64];
  float acc = 0.f;
  if (threadIdx.x < 58) for (unsigned t = 0; t < tiles; ++t) acc += part[t * 58 + threadIdx.x];
  if (threadIdx.x < 58) s[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x * 58] = s[3] + s[57];
}

int main() {
  unsigned* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  float* part; CK(hipMalloc(&part, 4 << 20)); CK(hipMemset(part, 0, 4 << 20));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s_plain, s_nb, s_prio;
  CK(hipStreamCreate(&s_plain));
  CK(hipStreamCreateWithFlags(&s_nb, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&s_prio, hipStreamNonBlocking, hi));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 2000;
  auto chain = [&](const char* name, hipStream_t s, auto launch) {
    for (int i = 0; i < 50; ++i) launch(s, i);
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) launch(s, i);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-58s %6.2f us per launch\n", name, best * 1e3 / N);
    fflush(stdout);
  };
  // the same chain enqueued BEHIND a 20 ms kernel (s_memrealtime ticks at 100 MHz): by the time it starts the host has
  // long finished enqueueing, so what is measured is the GPU-side cost of a dependent same-stream boundary alone
  auto prequeued = [&](const char* name, hipStream_t s, auto launch) {
    const int M = 1000;
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, 2000000ull);
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < M; ++i) launch(s, i);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-58s %6.2f us per launch (queued behind a busy stream)\n", name, best * 1e3 / M);
    fflush(stdout);
  };
  Arg104 a104{}; Arg1k a1k{};
  prequeued("empty, 1 block, non-blocking", s_nb, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, st, d); });
  prequeued("empty, 200 blocks, non-blocking", s_nb, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(200), dim3(256), 0, st, d); });
  prequeued("kernarg 104 B, 200 blocks", s_nb, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_arg104, dim3(200), dim3(256), 0, st, d, a104); });
  prequeued("reads predecessor's word, 200 blocks", s_nb, [&](hipStream_t st, int i) { hipLaunchKernelGGL(k_dep, dim3(200), dim3(256), 0, st, d, (unsigned)i); });
  prequeued("reads predecessor's word, 38 blocks", s_nb, [&](hipStream_t st, int i) { hipLaunchKernelGGL(k_dep, dim3(38), dim3(256), 0, st, d, (unsigned)i); });
  for (auto [sname, s] : {std::pair<const char*, hipStream_t>{"null stream", nullptr}, {"hipStreamCreate", s_plain},
                           {"non-blocking", s_nb}, {"non-blocking, highest priority", s_prio}}) {
    char buf[128];
    snprintf(buf, sizeof buf, "empty, 1 block, %s", sname);
    chain(buf, s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, st, d); });
  }
  hipStream_t s = s_nb;
  chain("empty, 200 blocks", s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(200), dim3(256), 0, st, d); });
  chain("empty, 200 x 1 x 1 grid as (tiles, pairs) = (200, 1)", s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(200, 1), dim3(256), 0, st, d); });
  chain("empty, 2048 blocks", s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, st, d); });
  chain("kernarg 104 B, 200 blocks", s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_arg104, dim3(200), dim3(256), 0, st, d, a104); });
  chain("kernarg 1 KB, 200 blocks", s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_arg1k, dim3(200), dim3(256), 0, st, d, a1k); });
  chain("LDS 5.5 KB, 200 blocks", s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_lds<1370>, dim3(200), dim3(256), 0, st, d); });
  chain("LDS 64 KB, 200 blocks", s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_lds<16384>, dim3(200), dim3(256), 0, st, d); });
  chain("reads predecessor's word, 200 blocks", s, [&](hipStream_t st, int i) { hipLaunchKernelGGL(k_dep, dim3(200), dim3(256), 0, st, d, (unsigned)i); });
  chain("head-like: 200 blocks each sum 200 x 58 partials", s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_headlike, dim3(200), dim3(256), 0, st, part, part + (1 << 19), 200u); });
  chain("head-like: 24 blocks each sum 24 x 58 partials", s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_headlike, dim3(24), dim3(256), 0, st, part, part + (1 << 19), 24u); });
  // two chains alternating between two streams with event dependencies would be the 'three stream group' form; not here
  return 0;
}
