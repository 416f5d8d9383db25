// Issue cost of the VALU instructions the bilateral blur is made of, on this chip: cycles per wave-instruction with one
// wave per SIMD (dependent chains of 8 interleaved accumulators, so latency is hidden) — hipcc --offload-arch=gfx950 -O3.
// The loop's own overhead (s_add, s_cmp, s_cbranch per 32 instructions) is inside the figures: read them relative to each other.
//   ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 512
#define OPS(name, decl, stmt)                                                                                  \
  __global__ void k_##name(unsigned long long* out, unsigned seed) {                                           \
    decl;                                                                                                      \
    unsigned long long t0 = __builtin_readcyclecounter();                                                      \
    _Pragma("unroll 1") for (int r = 0; r < REP; ++r) { stmt stmt stmt stmt stmt stmt stmt stmt }               \
    unsigned long long t1 = __builtin_readcyclecounter();                                                      \
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;                                                       \
    SINK                                                                                                       \
  }
#define SINK
typedef unsigned long long u64;
// eight independent accumulators per statement group keep the pipe full
#undef SINK
#define SINK out[1 + (threadIdx.x & 0)] += (u64)a0 + (u64)a1 + (u64)a2 + (u64)a3;
OPS(add_u32, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7,
    asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OPS(lshl_add_u32, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7,
    asm volatile("v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %2\n v_lshl_add_u32 %2, %2, 1, %3\n v_lshl_add_u32 %3, %3, 1, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OPS(lshl_add_u64, u64 a0 = seed + threadIdx.x; u64 a1 = a0 * 3; u64 a2 = a0 * 5; u64 a3 = a0 * 7,
    asm volatile("v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %2\n v_lshl_add_u64 %2, %2, 1, %3\n v_lshl_add_u64 %3, %3, 1, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OPS(add_f64, double a0 = seed + threadIdx.x; double a1 = a0 * 3; double a2 = a0 * 5; double a3 = a0 * 7,
    asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OPS(fma_f64, double a0 = seed + threadIdx.x; double a1 = a0 * 3; double a2 = a0 * 5; double a3 = a0 * 7,
    asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OPS(mul_f64, double a0 = seed + threadIdx.x; double a1 = a0 * 3; double a2 = a0 * 5; double a3 = a0 * 7,
    asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OPS(add_f32, float a0 = seed + threadIdx.x; float a1 = a0 * 3; float a2 = a0 * 5; float a3 = a0 * 7,
    asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OPS(rcp_f64, double a0 = seed + threadIdx.x + 1; double a1 = a0 * 3; double a2 = a0 * 5; double a3 = a0 * 7,
    asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
#undef SINK
#define SINK out[1 + (threadIdx.x & 0)] += (u64)a0 + (u64)a1 + (u64)b0 + (u64)b1;
OPS(cvt_f64_u32, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; double b0 = 0; double b1 = 0,
    asm volatile("v_cvt_f64_u32 %2, %0\n v_cvt_f64_u32 %3, %1\n v_cvt_f64_u32 %2, %1\n v_cvt_f64_u32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));)
OPS(cvt_u32_f64, unsigned a0 = 0; unsigned a1 = 0; double b0 = seed + threadIdx.x; double b1 = b0 * 3,
    asm volatile("v_cvt_u32_f64 %0, %2\n v_cvt_u32_f64 %1, %3\n v_cvt_u32_f64 %0, %3\n v_cvt_u32_f64 %1, %2" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));)
OPS(cvt_f32_u32, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; float b0 = 0; float b1 = 0,
    asm volatile("v_cvt_f32_u32 %2, %0\n v_cvt_f32_u32 %3, %1\n v_cvt_f32_u32 %2, %1\n v_cvt_f32_u32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));)
OPS(mov_dpp, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned b0 = 0; unsigned b1 = 0,
    asm volatile("v_mov_b32_dpp %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));)
OPS(add_dpp, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned b0 = 0; unsigned b1 = 0,
    asm volatile("v_add_u32_dpp %2, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %3, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %0, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %2, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));)
// dependent chains (one accumulator, every instruction waits for the one before): latency per instruction
#undef SINK
#define SINK out[1 + (threadIdx.x & 0)] += (u64)a0;
OPS(dep_fma_f64, double a0 = 1.0 + 1e-9 * (seed + threadIdx.x); double a1 = 1.0000001,
    asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1" : "+v"(a0) : "v"(a1));)
OPS(dep_mul_f64, double a0 = 1.0 + 1e-9 * (seed + threadIdx.x); double a1 = 1.0000001,
    asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(a1));)
OPS(dep_rsq_f64, double a0 = 1.0 + 1e-9 * (seed + threadIdx.x); double a1 = 1.0000001,
    asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0" : "+v"(a0) : "v"(a1));)
OPS(dep_fma_f32, float a0 = 1.0f + 1e-6f * (seed + threadIdx.x); float a1 = 1.0001f,
    asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(a1));)
OPS(dep_cvt_f32_f64, double a0 = 1.0 + 1e-9 * (seed + threadIdx.x); float a1 = 0.f,
    asm volatile("v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1" : "+v"(a0), "+v"(a1));)

// ---- round 6, kd-tree network: what a 64-bit compare-exchange is made of -------------------------------------------------
#undef SINK
#define SINK out[1 + (threadIdx.x & 0)] += (u64)a0 + (u64)a1 + (u64)a2 + (u64)a3 + (u64)c0;
OPS(cmp_lt_u64, u64 a0 = seed + threadIdx.x; u64 a1 = a0 * 3; u64 a2 = a0 * 5; u64 a3 = a0 * 7; unsigned c0 = 0,
    asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %2\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
OPS(cmp_lt_u32, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; unsigned c0 = 0,
    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
// compare -> select on the mask (VALU writes vcc, VALU reads it)
OPS(cmp64_cnd, u64 a0 = seed + threadIdx.x; u64 a1 = a0 * 3; u64 a2 = a0 * 5; u64 a3 = a0 * 7; unsigned c0 = 0,
    asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %4, %4, %4, vcc\n v_cmp_lt_u64 vcc, %2, %3\n v_cndmask_b32 %4, %4, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0) : : "vcc");)
// compare -> SALU xor with a direction mask -> select (what `if ((hi < lo) == up) swap` compiles to)
OPS(cmp64_xor_cnd, u64 a0 = seed + threadIdx.x; u64 a1 = a0 * 3; u64 a2 = a0 * 5; u64 a3 = a0 * 7; unsigned c0 = 0,
    asm volatile("v_cmp_lt_u64 vcc, %0, %1\n s_xor_b64 vcc, vcc, exec\n s_nop 0\n v_cndmask_b32 %4, %4, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0) : : "vcc", "scc");)  // (s_xor writes SCC: the loop's own s_cmp lives there)
// the same decision from the borrow of a 64-bit subtraction (two full-rate instructions?)
OPS(sub_borrow, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; unsigned c0 = 0,
    asm volatile("v_sub_co_u32 %4, vcc, %0, %1\n v_subb_co_u32 %4, vcc, %2, %3, vcc\n v_sub_co_u32 %4, vcc, %1, %0\n v_subb_co_u32 %4, vcc, %3, %2, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0) : : "vcc");)
OPS(cndmask, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; unsigned c0 = 0,
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
OPS(permlane16_swap, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; unsigned c0 = 0,
    asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
OPS(permlane32_swap, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; unsigned c0 = 0,
    asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
// DPP read of a register a VALU instruction has just written (the hazard the compiler pads with s_nop 1)
OPS(dpp_after_write, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; unsigned c0 = 0,
    asm volatile("v_add_u32 %0, %0, %1\n s_nop 1\n v_mov_b32_dpp %2, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32 %1, %1, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)

// ---- v_cndmask and where its mask comes from (round 6) ---------------------------------------------------------------------------
#undef SINK
#define SINK out[1 + (threadIdx.x & 0)] += (u64)a0 + (u64)a1 + (u64)a2 + (u64)a3;
#define CND4 "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc"
#define CND4S "v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]"
// the mask written once by a VALU compare before the loop
OPS(cnd_vcc_valu_once, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"),
    asm volatile(CND4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
// ... once by the scalar unit before the loop
OPS(cnd_vcc_salu_once, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; asm volatile("s_mov_b64 vcc, 0x55555555" : : : "vcc"),
    asm volatile(CND4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
// ... in an SGPR pair other than vcc (VOP3 encoding), written by the scalar unit before the loop
OPS(cnd_sgpr_salu_once, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; asm volatile("s_mov_b64 s[20:21], 0x55555555" : : : "s20", "s21"),
    asm volatile(CND4S : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21");)
// ... rewritten by a VALU compare in front of every four selects
OPS(cnd_vcc_valu_each, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7,
    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n" CND4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
// ... rewritten by the scalar unit in front of every four selects
OPS(cnd_vcc_salu_each, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7,
    asm volatile("s_not_b64 vcc, vcc\n s_nop 0\n" CND4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc", "scc");)
// the same selects as v_bfi_b32 with the mask as data (no SGPR operand at all)
OPS(bfi_mask_data, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; unsigned m = threadIdx.x & 1 ? ~0u : 0u,
    asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));)

// eight of them in ONE asm statement (no scalar instruction of the loop between them) / with an unrelated scalar add between each two
#define CND8 CND4 "\n" CND4
OPS(cnd_vcc_block8, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"),
    asm volatile(CND8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
#define CND4_SALU "v_cndmask_b32 %0, %0, %1, vcc\n s_add_u32 s20, s20, 1\n v_cndmask_b32 %1, %1, %2, vcc\n s_add_u32 s20, s20, 1\n v_cndmask_b32 %2, %2, %3, vcc\n s_add_u32 s20, s20, 1\n v_cndmask_b32 %3, %3, %0, vcc\n s_add_u32 s20, s20, 1"
OPS(cnd_vcc_salu_between, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"),
    asm volatile(CND4_SALU : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "scc");)
#define ADD4_SALU "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %2\n s_add_u32 s20, s20, 1\n v_add_u32 %2, %2, %3\n s_add_u32 s20, s20, 1\n v_add_u32 %3, %3, %0\n s_add_u32 s20, s20, 1"
OPS(add_salu_between, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7,
    asm volatile(ADD4_SALU : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "scc");)

// VOP3 encoding with vcc named as the mask operand / VOP2 selects with an unrelated VALU add between each two
#define CND4E64V "v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %0, vcc"
OPS(cnd_e64_vcc, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"),
    asm volatile(CND4E64V : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
#define CND4_VADD "v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %4, %4, %4\n v_cndmask_b32 %1, %1, %2, vcc\n v_add_u32 %4, %4, %4\n v_cndmask_b32 %2, %2, %3, vcc\n v_add_u32 %4, %4, %4\n v_cndmask_b32 %3, %3, %0, vcc\n v_add_u32 %4, %4, %4"
OPS(cnd_vcc_vadd_between, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; unsigned m = a0; asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"),
    asm volatile(CND4_VADD : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(m));)
// independent selects (no register of one is another's source)
OPS(cnd_vcc_independent, unsigned a0 = seed + threadIdx.x; unsigned a1 = a0 * 3; unsigned a2 = a0 * 5; unsigned a3 = a0 * 7; unsigned b0 = 1; unsigned b1 = 2; asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"),
    asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
#define RUN(name, per)                                                                                   \
  for (int waves = 1; waves <= 4; waves *= 4) {                                                          \
    hipLaunchKernelGGL(k_##name, dim3(1), dim3(256 * waves), 0, 0, d, 1u);                                \
    hipDeviceSynchronize();                                                                              \
    hipMemcpy(h.data(), d, 16, hipMemcpyDeviceToHost);                                                   \
    printf("%-14s %d wave(s)/SIMD: %.2f cycles per instruction per wave (shader clock counter)\n", #name, waves, (double)h[0] / (REP * 8.0 * per) * 1.0); \
  }
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const bool only_new = argc > 1;  // ./valu_rate cmp: the round-6 kd-tree additions only
  u64* d;
  hipMalloc(&d, 1024);
  hipMemset(d, 0, 1024);
  std::vector<u64> h(2);
  if (!only_new) {
  RUN(add_u32, 4) RUN(lshl_add_u32, 4) RUN(lshl_add_u64, 4) RUN(add_f32, 4) RUN(add_f64, 4) RUN(mul_f64, 4) RUN(fma_f64, 4) RUN(rcp_f64, 4)
  RUN(cvt_f64_u32, 4) RUN(cvt_u32_f64, 4) RUN(cvt_f32_u32, 4) RUN(mov_dpp, 4) RUN(add_dpp, 4)
  printf("dependent chains (latency per instruction):\n");
  RUN(dep_fma_f32, 4) RUN(dep_fma_f64, 4) RUN(dep_mul_f64, 4) RUN(dep_rsq_f64, 4) RUN(dep_cvt_f32_f64, 4)
  }
  printf("64-bit compare-exchange parts (kd-tree network):\n");
  RUN(cmp_lt_u64, 4) RUN(cmp_lt_u32, 4) RUN(cmp64_cnd, 4) RUN(cmp64_xor_cnd, 4) RUN(sub_borrow, 4) RUN(cndmask, 4) RUN(permlane16_swap, 4) RUN(permlane32_swap, 4) RUN(dpp_after_write, 4)
  printf("v_cndmask by the origin of its mask (4 selects per statement; the *_each forms: 5 / 6 instructions):\n");
  RUN(cnd_vcc_valu_once, 4) RUN(cnd_vcc_salu_once, 4) RUN(cnd_sgpr_salu_once, 4) RUN(cnd_vcc_valu_each, 5) RUN(cnd_vcc_salu_each, 6) RUN(bfi_mask_data, 4)
  RUN(cnd_vcc_block8, 8) RUN(cnd_vcc_salu_between, 8) RUN(add_salu_between, 8) RUN(cnd_e64_vcc, 4) RUN(cnd_vcc_vadd_between, 8) RUN(cnd_vcc_independent, 4)
  return 0;
}
