// How many VALU wave-instructions per cycle one SIMD of this chip retires with 1, 2 and 4 resident waves: every wave of ONE block
// runs the same loop and stamps its own start and end (s_memtime); the figure is instructions of all waves on a SIMD / (last end -
// first start).  scripts/valu_rate.hip times one wave only (its 4-waves column is not a throughput figure).
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_throughput.hip -o valu_throughput && ./valu_throughput
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
#define REP 512
#define KERNEL(name, decl, stmt, sink)                                                          \
  __global__ void k_##name(u64* out, unsigned seed) {                                           \
    decl;                                                                                       \
    __syncthreads();                                                                            \
    const u64 t0 = __builtin_amdgcn_s_memtime();                                                \
    _Pragma("unroll 1") for (int r = 0; r < REP; ++r) { stmt stmt stmt stmt stmt stmt stmt stmt } \
    const u64 t1 = __builtin_amdgcn_s_memtime();                                                \
    if ((threadIdx.x & 63) == 0) out[2 * (threadIdx.x >> 6)] = t0, out[2 * (threadIdx.x >> 6) + 1] = t1; \
    out[64 + (threadIdx.x & 0)] += sink;                                                        \
  }
KERNEL(add_u32, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7,
       asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));,
       (u64)a0 + a1 + a2 + a3)
KERNEL(fma_f32, float a0 = seed + threadIdx.x; float a1 = a0 * 3; float a2 = a0 * 5; float a3 = a0 * 7,
       asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));,
       (u64)(a0 + a1 + a2 + a3))
// the lane step of the kd-tree network, one word (a dependent chain) and four words (what a thread holds): partner's word by
// DPP, compare, swap mask, two v_bfi — 6 VALU per word (a 32-bit compare stands in for v_cmp_lt_u64: same rate, valu_rate.hip)
#define LANE_WORD(xl, xh, yl, yh, m)                                                                                     \
  "v_mov_b32_dpp " yl ", " xl " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp " yh ", " xh " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
  "v_cmp_lt_u32 vcc, " yh ", " xh "\n v_cndmask_b32 " m ", %0, %1, vcc\n v_bfi_b32 " xl ", " m ", " yl ", " xl "\n v_bfi_b32 " xh ", " m ", " yh ", " xh "\n"
KERNEL(lane_1word, unsigned f = threadIdx.x & 1 ? ~0u : 0u; unsigned nf = ~f; unsigned xl = seed * threadIdx.x; unsigned xh = xl * 7; unsigned yl = 0; unsigned yh = 0; unsigned m = 0,
       asm volatile(LANE_WORD("%2", "%3", "%4", "%5", "%6") : "+v"(f), "+v"(nf), "+v"(xl), "+v"(xh), "+v"(yl), "+v"(yh), "+v"(m) : : "vcc");,
       (u64)xl + xh)
KERNEL(lane_4words, unsigned f = threadIdx.x & 1 ? ~0u : 0u; unsigned nf = ~f; unsigned xl = seed * threadIdx.x; unsigned xh = xl * 7; unsigned yl = 0; unsigned yh = 0; unsigned m = 0;
       unsigned xl1 = xl * 3; unsigned xh1 = xl * 5; unsigned yl1 = 0; unsigned yh1 = 0; unsigned m1 = 0; unsigned xl2 = xl * 9; unsigned xh2 = xl * 11; unsigned yl2 = 0; unsigned yh2 = 0; unsigned m2 = 0;
       unsigned xl3 = xl * 13; unsigned xh3 = xl * 15; unsigned yl3 = 0; unsigned yh3 = 0; unsigned m3 = 0,
       asm volatile(LANE_WORD("%2", "%3", "%4", "%5", "%6") LANE_WORD("%7", "%8", "%9", "%10", "%11") LANE_WORD("%12", "%13", "%14", "%15", "%16") LANE_WORD("%17", "%18", "%19", "%20", "%21")
                    : "+v"(f), "+v"(nf), "+v"(xl), "+v"(xh), "+v"(yl), "+v"(yh), "+v"(m), "+v"(xl1), "+v"(xh1), "+v"(yl1), "+v"(yh1), "+v"(m1),
                      "+v"(xl2), "+v"(xh2), "+v"(yl2), "+v"(yh2), "+v"(m2), "+v"(xl3), "+v"(xh3), "+v"(yl3), "+v"(yh3), "+v"(m3) : : "vcc");,
       (u64)xl + xh + xl1 + xh1 + xl2 + xh2 + xl3 + xh3)
// selects: VOP2 v_cndmask_b32 (implicit vcc) back to back, and the same selects in the VOP3 encoding (scripts/valu_rate.hip cmp:
// a lone wave needs ~22 cycles for each VOP2 select that follows another one, 6 in VOP3)
KERNEL(cnd_vop2, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"),
       asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));,
       (u64)a0 + a1 + a2 + a3)
KERNEL(cnd_vop3, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"),
       asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));,
       (u64)a0 + a1 + a2 + a3)
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  u64* d;
  hipMalloc(&d, 1024);
  hipMemset(d, 0, 1024);
  std::vector<u64> h(64);
#define RUN(name, per)                                                                                       \
  for (int waves = 1; waves <= 4; waves *= 2) {                                                              \
    hipLaunchKernelGGL(k_##name, dim3(1), dim3(256 * waves), 0, 0, d, 1u);                                    \
    hipDeviceSynchronize();                                                                                  \
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);                                                      \
    u64 first = ~0ull, last = 0, own = 0;                                                                    \
    for (int w = 0; w < 4 * waves; ++w) { first = h[2 * w] < first ? h[2 * w] : first; last = h[2 * w + 1] > last ? h[2 * w + 1] : last; own += h[2 * w + 1] - h[2 * w]; } \
    const double instr_per_simd = (double)REP * 8 * per * waves;                                             \
    printf("%-10s %d wave(s)/SIMD: span %7llu cycles, %.2f cycles per wave-instruction per SIMD (a wave's own loop: %.0f cycles)\n", #name, waves, \
           (unsigned long long)(last - first), (double)(last - first) / instr_per_simd, (double)own / (4 * waves)); \
  }
  RUN(add_u32, 4) RUN(fma_f32, 4) RUN(lane_1word, 6) RUN(lane_4words, 24) RUN(cnd_vop2, 4) RUN(cnd_vop3, 4)
  return 0;
}
