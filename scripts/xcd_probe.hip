// Where do blocks land, and what does a hand-off between two blocks cost when they share an XCD's L2?  (tuning aid)
//   1. XCC_ID of every block of a 1-D grid: is block i on XCD i % 8?
//   2. ping-pong between two blocks (one wave each) — flag + payload — with (a) agent-scope accesses (sc1: what
//      icp_engine.hpp's last-block hand-off uses) and (b) L2-local accesses (plain stores drained with s_waitcnt, loads
//      with sc0 = bypass the CU's L1 only), for a pair of blocks on ONE XCD and a pair on two XCDs.  Every spin is bounded.
//   3. latency of a returning atomic add, agent scope, from one wave.
//   hipcc --offload-arch=gfx950 -O2 scripts/xcd_probe.hip -o scripts/xcd_probe && scripts/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15u;
}
__global__ void where_kernel(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

template <int MODE> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
  if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: write-through
  else asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");        // plain: to this XCD's L2
}
template <int MODE> __device__ __forceinline__ unsigned ld(const unsigned* p) {
  unsigned v;
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// blocks `ba` and `bb` play; ball[0] = flag A->B, ball[16] = flag B->A, ball[32..] payload.  out: [0] ticks, [1] bad
// payloads, [2] timeouts, [3] xcc of a, [4] xcc of b
template <int MODE>
__global__ void pingpong_kernel(unsigned* ball, unsigned ba, unsigned bb, int rounds, unsigned long long* out) {
  if (blockIdx.x != ba && blockIdx.x != bb) return;
  if (threadIdx.x != 0) return;
  const bool is_a = blockIdx.x == ba;
  out[is_a ? 3 : 4] = xcc_id();
  unsigned bad = 0, timeouts = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 1; r <= rounds; ++r) {
    if (is_a) {
      st<MODE>(ball + 32, 1000u + (unsigned)r);  // payload
      drain();
      st<MODE>(ball, (unsigned)r);               // flag
      int spin = 0;
      while (ld<MODE>(ball + 16) != (unsigned)r && ++spin < 200000) {}
      timeouts += spin >= 200000;
    } else {
      int spin = 0;
      while (ld<MODE>(ball) != (unsigned)r && ++spin < 200000) {}
      timeouts += spin >= 200000;
      bad += ld<MODE>(ball + 32) != 1000u + (unsigned)r;
      st<MODE>(ball + 16, (unsigned)r);
    }
    if (timeouts > 3) break;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (is_a) out[0] = t1 - t0;
  if (!is_a) out[1] = bad;
  atomicAdd(&out[2], (unsigned long long)timeouts);
}

__global__ void atomic_latency_kernel(unsigned* ctr, int n, unsigned long long* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) acc += __hip_atomic_fetch_add(ctr + (acc & 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  out[0] = __builtin_amdgcn_s_memrealtime() - t0;
  out[1] = acc;
}

int main() {
  const int NB = 2048;
  unsigned* d_where;
  CK(hipMalloc(&d_where, NB * 4));
  hipLaunchKernelGGL(where_kernel, dim3(NB), dim3(64), 0, 0, d_where);
  std::vector<unsigned> where(NB);
  CK(hipMemcpy(where.data(), d_where, NB * 4, hipMemcpyDeviceToHost));
  int hist[16] = {0}, rr = 0;
  for (int i = 0; i < NB; ++i) hist[where[i] & 15]++, rr += (where[i] == where[0] + 0u ? 0 : 0), rr += ((where[i] & 15u) == (unsigned)((where[0] + i) % 8));
  printf("XCC_ID histogram over %d blocks:", NB);
  for (int x = 0; x < 16; ++x) if (hist[x]) printf(" [%d]=%d", x, hist[x]);
  printf("\nblocks with XCC_ID == (XCC_ID(block 0) + i) %% 8: %d of %d; first 16:", rr, NB);
  for (int i = 0; i < 16; ++i) printf(" %u", where[i]);
  printf("\n");

  unsigned* ball;
  unsigned long long* out;
  CK(hipMalloc(&ball, 4096));
  CK(hipMalloc(&out, 64));
  const int rounds = 300;
  for (int same = 1; same >= 0; --same)
    for (int mode = 0; mode < 2; ++mode) {
      CK(hipMemset(ball, 0, 4096));
      CK(hipMemset(out, 0, 64));
      const unsigned ba = 8, bb = same ? 16 : 9;  // same residue mod 8 = same XCD (if placement is round robin)
      if (mode == 0) hipLaunchKernelGGL(pingpong_kernel<0>, dim3(64), dim3(64), 0, 0, ball, ba, bb, rounds, out);
      else hipLaunchKernelGGL(pingpong_kernel<1>, dim3(64), dim3(64), 0, 0, ball, ba, bb, rounds, out);
      unsigned long long h[8];
      CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
      printf("ping-pong blocks %u,%u (XCC %llu,%llu) %-28s: %.0f ns per round trip (2 hand-offs), bad payloads %llu, timeouts %llu\n", ba, bb,
             h[3], h[4], mode == 0 ? "agent scope (sc1)" : "L2-local (plain st, sc0 ld)", (double)h[0] * 10.0 / rounds, h[1], h[2]);
    }
  CK(hipMemset(ball, 0, 4096));
  hipLaunchKernelGGL(atomic_latency_kernel, dim3(1), dim3(64), 0, 0, ball, 500, out);
  unsigned long long h[2];
  CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  printf("returning agent-scope atomic add: %.0f ns each (dependent chain of 500)\n", (double)h[0] * 10.0 / 500);
  return 0;
}
