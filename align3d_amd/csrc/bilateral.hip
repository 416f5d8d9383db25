// BilateralFilter<u16>::filter (src/bilateral/edge_aware_filter.rs:126-135): bilateral-grid splat,
// 3 axes x 2 passes of the [1 2 1]/4 blur, normalise, trilinear slice — all in f64 like the reference,
// compiled without contraction, so the u16 result is bit-identical to the CPU path.
//
// Why the splat may use atomics: every summand is an integer (a depth value, or 1.0) and the totals
// stay far below 2^53, so f64 addition is exact and the order of the adds cannot matter.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "bilateral.hpp"

using namespace a3d;

namespace {

// min / max over ALL pixels, zeros included (src/bilateral/grid.rs:41-49)
__global__ void __launch_bounds__(256)
    minmax_u16_kernel(const uint16_t* __restrict__ img, uint32_t n, uint32_t* __restrict__ out_minmax,
                      uint32_t* __restrict__ partials) {
  img += (size_t)blockIdx.y * n;  // blockIdx.y = frame of a batch ([frames][n] pixels)
  uint32_t mi = 0xFFFFu, ma = 0u;
  // eight pixels per 16-byte load when the image is 16-byte aligned (it is when it comes from the library's own
  // allocations), the remainder and unaligned images one by one
  const bool aligned = ((uintptr_t)img & 15u) == 0;
  const uint32_t n8 = aligned ? n / 8 : 0;
  const uint4* img8 = (const uint4*)img;
  // (four loads of a thread in flight at a time: a 640x480 frame is 2.3 loads per thread of its 64 blocks, and one at a
  // time the kernel ran at a seventh of the memory rate)
  const uint32_t step = gridDim.x * blockDim.x;
  for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n8; i0 += 4 * step) {
    uint4 q[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) q[u] = img8[min(i0 + u * step, n8 - 1)];  // (a clamped repeat changes neither result)
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t lo = w[k] & 0xFFFFu, hi = w[k] >> 16;
        mi = min(mi, min(lo, hi));
        ma = max(ma, max(lo, hi));
      }
    }
  }
  for (uint32_t i = n8 * 8 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t v = img[i];
    mi = min(mi, v);
    ma = max(ma, v);
  }
  for (int off = 32; off >= 1; off >>= 1) {
    mi = min(mi, (uint32_t)__shfl_xor((int)mi, off, 64));
    ma = max(ma, (uint32_t)__shfl_xor((int)ma, off, 64));
  }
  __shared__ uint32_t s_mi[4], s_ma[4];
  if ((threadIdx.x & 63) == 0) s_mi[threadIdx.x >> 6] = mi, s_ma[threadIdx.x >> 6] = ma;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t bmin = min(min(s_mi[0], s_mi[1]), min(s_mi[2], s_mi[3]));
    const uint32_t bmax = max(max(s_ma[0], s_ma[1]), max(s_ma[2], s_ma[3]));
    if (partials) {  // batches: one (min, max) per block, reduced by the frame's last block — no atomics on the values
      partials[(blockIdx.y * gridDim.x + blockIdx.x) * 2 + 0] = bmin;
      partials[(blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = bmax;
    } else {         // one atomic pair per block: 64 blocks, not thousands of waves, meet on the two words
      atomicMin(&out_minmax[0], bmin);
      atomicMax(&out_minmax[1], bmax);
    }
  }
}

__device__ __forceinline__ void dims_table_body(uint32_t cmin, uint32_t cmax, uint32_t part, uint32_t parts, uint32_t* __restrict__ sc,
                                                uint32_t w, uint32_t h, double sigma_space, double sigma_color,
                                                unsigned long long capacity_cells, uint32_t* __restrict__ channel_of, double inv_sc);
#ifdef A3D_DIAGNOSTICS  // (A3D_BILATERAL_MINMAX=fused: measured no faster than the two launches, kept as a cross-check)
// The batch form, round 6: min / max AND what dims_table_kernel did, in one launch.  The frame's blocks fold their
// (min, max) into two accumulator words and take a ticket (scal[SC_ACC_NMIN], [SC_ACC_MAX], [SC_TICKET], zero between
// enqueues: the last block puts the zeros back with the rest of the scalar block); whoever draws the last ticket sizes the
// grid and fills the colour -> channel table alone (a few thousand entries; 16 blocks did it before) — one dependent launch and its boundary less in front of
// the splat (4.8 + ~2 us per launch sequence).
__global__ void __launch_bounds__(256)
    minmax_dims_kernel(const uint16_t* __restrict__ img, uint32_t n, uint32_t* __restrict__ scal,
                       uint32_t w, uint32_t h, double sigma_space, double sigma_color, unsigned long long capacity_cells,
                       uint32_t* __restrict__ channel_of, double inv_sc) {
  img += (size_t)blockIdx.y * n;  // blockIdx.y = frame of a batch ([frames][n] pixels)
  uint32_t mi = 0xFFFFu, ma = 0u;
  const bool aligned = ((uintptr_t)img & 15u) == 0;
  const uint32_t n8 = aligned ? n / 8 : 0;
  const uint4* img8 = (const uint4*)img;
  const uint32_t step = gridDim.x * blockDim.x;
  for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n8; i0 += 4 * step) {  // (as minmax_u16_kernel)
    uint4 q[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) q[u] = img8[min(i0 + u * step, n8 - 1)];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t wd[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t lo = wd[k] & 0xFFFFu, hi = wd[k] >> 16;
        mi = min(mi, min(lo, hi));
        ma = max(ma, max(lo, hi));
      }
    }
  }
  for (uint32_t i = n8 * 8 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t v = img[i];
    mi = min(mi, v);
    ma = max(ma, v);
  }
  for (int off = 32; off >= 1; off >>= 1) {
    mi = min(mi, (uint32_t)__shfl_xor((int)mi, off, 64));
    ma = max(ma, (uint32_t)__shfl_xor((int)ma, off, 64));
  }
  __shared__ uint32_t s_mi[4], s_ma[4], s_last;
  if ((threadIdx.x & 63) == 0) s_mi[threadIdx.x >> 6] = mi, s_ma[threadIdx.x >> 6] = ma;
  __syncthreads();
  uint32_t* sc = scal + blockIdx.y * SC_STRIDE;
  if (threadIdx.x == 0) {
    // Relaxed agent-scope atomics only: a release / acquire pair at agent scope is a write-back and an invalidate of the
    // XCD's whole L2 on this part, per block (the first form of this kernel took 66 us instead of 11).  Two running maxima
    // (of 0xFFFF - min and of max: both start at the zero the scalar block holds between enqueues), each WITH its return
    // value, and the ticket's increment made to depend on those values: an atomic has been performed at the memory side
    // when its value comes back, so whoever draws the last ticket sees every block's contribution.
    const uint32_t bmin = min(min(s_mi[0], s_mi[1]), min(s_mi[2], s_mi[3]));
    const uint32_t bmax = max(max(s_ma[0], s_ma[1]), max(s_ma[2], s_ma[3]));
    const uint32_t a = __hip_atomic_fetch_max(&sc[SC_ACC_NMIN], 0xFFFFu - bmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t b = __hip_atomic_fetch_max(&sc[SC_ACC_MAX], bmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t one = 1u + (((a | b) >> 31) & 1u);  // (= 1: the accumulators stay below 2^16)
    s_last = __hip_atomic_fetch_add(&sc[SC_TICKET], one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) {
    s_mi[0] = 0xFFFFu - __hip_atomic_load(&sc[SC_ACC_NMIN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ma[0] = __hip_atomic_load(&sc[SC_ACC_MAX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // (the body rewrites the whole scalar block: ticket and accumulators are zero again for the next enqueue)
  dims_table_body(s_mi[0], s_ma[0], 0u, 1u, sc, w, h, sigma_space, sigma_color, capacity_cells, channel_of, inv_sc);
}
#endif  // A3D_DIAGNOSTICS

// BilateralGrid::from_image splat (src/bilateral/grid.rs:60-78); grid cell = {value sum, count}
__global__ void __launch_bounds__(256)
    splat_kernel(const uint16_t* __restrict__ img, uint32_t w, uint32_t h, double inv_ss, double inv_sc,
                 uint32_t color_min, GridDims g, double* __restrict__ grid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const uint32_t color = img[i];
  if (color == 0) return;  // `color <= I::min_value()` (:67)
  const uint32_t row = i / w, col = i % w;
  const uint32_t grow = f64_as_usize((double)row * inv_ss + 0.5) + 2;
  const uint32_t gcol = f64_as_usize((double)col * inv_ss + 0.5) + 2;
  const uint32_t ch = f64_as_usize((double)(color - color_min) * inv_sc + 0.5) + 2;
  const size_t cell = (((size_t)grow * g.gw + gcol) * g.gd + ch) * 2;
  atomicAdd(&grid[cell], (double)color);
  atomicAdd(&grid[cell + 1], 1.0);
}

// One pass of BilateralFilter::convolution (src/bilateral/edge_aware_filter.rs:68-114) along `axis`
// (0 = row, 1 = col, 2 = channel): out = (prev + next + 2 cur) * 0.25 on value and weight, written for
// rows 1..gh-2, cols 1..gw-2, channels 0..gd-2; every other cell is never written and stays zero in both
// buffers.  The reference's channel loop starts at an un-offset pointer, so at channel 0 its "previous"
// read aliases cell (row, col-1, gd-1), which is always zero: read as 0.0 here.
__global__ void __launch_bounds__(256)
    blur_axis_kernel(const double2* __restrict__ in, double2* __restrict__ out, GridDims g, int axis) {
  const uint32_t zc = g.gd - 1;  // channels per (row, col) that are written
  const uint64_t interior = (uint64_t)(g.gh - 2) * (g.gw - 2) * zc;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= interior) return;
  const uint32_t z = (uint32_t)(t % zc);
  const uint64_t rc = t / zc;
  const uint32_t col = 1 + (uint32_t)(rc % (g.gw - 2));
  const uint32_t row = 1 + (uint32_t)(rc / (g.gw - 2));
  const size_t cell = ((size_t)row * g.gw + col) * g.gd + z;
  const size_t stride = axis == 0 ? (size_t)g.gw * g.gd : (axis == 1 ? (size_t)g.gd : 1);
  const double2 cur = in[cell];
  const double2 next = in[cell + stride];
  double2 prev = make_double2(0.0, 0.0);
  if (!(axis == 2 && z == 0)) prev = in[cell - stride];
  double2 o;
  o.x = (prev.x + next.x + 2.0 * cur.x) * 0.25;
  o.y = (prev.y + next.y + 2.0 * cur.y) * 0.25;
  out[cell] = o;
}

// ---- packed splat + fused blur (images below 2^24 pixels) ------------------------------------------------
// Both sums of a cell are integers: the value sum is < pixels x 65535 < 2^40 and the count < 2^24, so a cell is
// ONE u64 (value sum << 24 | count) and a pixel is ONE integer add (in the gather-splat: an LDS add into the owning
// thread's slot) instead of two f64 atomics; integer adds are exact, so the cell holds exactly the reference's f64 sums.
constexpr int PACK_SHIFT = 24;
// When at most 255 pixels can splat into one (row, column) of the grid — sigma_space below ~14 — the count of a cell
// is < 2^8 and its value sum < 255 x 65535 < 2^24: the cell is ONE u32 (value sum << 8 | count), half the bytes to
// write and to load in the blur.
template <typename CELL>
struct Pack;
// Weight: the type the blur carries a cell's COUNT in.  Six unnormalised [1 2 1] passes multiply a count by at most
// 4^6, so a count below 2^8 stays below 2^20: every intermediate is an integer that f32 holds exactly, and the weight
// channel of the narrow cells runs on the full-rate f32 pipe (one DPP move per shift instead of two) while the value
// channel (sums up to 2^24 x 4^6 = 2^36) stays in f64.  Wide cells count up to 2^24: f64 for both.
// Narrow cells go further: a value sum is below 255 x 65535 < 2^24, so after FOUR unnormalised passes (x 4^4) it is
// still below 2^32: the first two axes of the blur (rows, channels) run on u32 integers — full-rate adds, one DPP move
// per shift, 4-byte LDS cells — and only the last axis, whose results reach 2^36, runs in f64 (value) / f32 (weight).
// The conversions u32 -> f64 the tile needs anyway simply happen after four passes instead of before the first.
template <>
struct Pack<unsigned long long> {
  static constexpr int SHIFT = PACK_SHIFT;
  typedef double EarlyValue;   // first two axes
  typedef double EarlyWeight;
  typedef double Weight;       // last axis
};
template <>
struct Pack<uint32_t> {
  static constexpr int SHIFT = 8;
  typedef uint32_t EarlyValue;
  typedef uint32_t EarlyWeight;
  typedef float Weight;
};
// 2 c + (p + n) in the line's own type: exact in every type the blur uses it with (see above)
__device__ __forceinline__ double blur3(double p, double c, double n) { return __builtin_fma(2.0, c, p + n); }
__device__ __forceinline__ float blur3(float p, float c, float n) { return 2.0f * c + (p + n); }
__device__ __forceinline__ uint32_t blur3(uint32_t p, uint32_t c, uint32_t n) { return 2u * c + (p + n); }
// the value of the lane one below (from_lower_lane) / above in the same 16-lane DPP row; 0 at the row's ends
template <typename T>
__device__ __forceinline__ T row_neighbour(T x, bool from_lower_lane) {
  if constexpr (sizeof(T) == 8) {
    unsigned long long b;
    __builtin_memcpy(&b, &x, 8);
    int lo32 = (int)(uint32_t)b, hi32 = (int)(uint32_t)(b >> 32);
    if (from_lower_lane) {  // row_shr:1 = 0x111, bound_ctrl: lanes without a source get 0
      lo32 = __builtin_amdgcn_update_dpp(0, lo32, 0x111, 0xF, 0xF, true);
      hi32 = __builtin_amdgcn_update_dpp(0, hi32, 0x111, 0xF, 0xF, true);
    } else {                // row_shl:1 = 0x101
      lo32 = __builtin_amdgcn_update_dpp(0, lo32, 0x101, 0xF, 0xF, true);
      hi32 = __builtin_amdgcn_update_dpp(0, hi32, 0x101, 0xF, 0xF, true);
    }
    b = ((unsigned long long)(uint32_t)hi32 << 32) | (uint32_t)lo32;
    T r;
    __builtin_memcpy(&r, &b, 8);
    return r;
  } else {
    int b;
    __builtin_memcpy(&b, &x, 4);
    b = from_lower_lane ? __builtin_amdgcn_update_dpp(0, b, 0x111, 0xF, 0xF, true)
                        : __builtin_amdgcn_update_dpp(0, b, 0x101, 0xF, 0xF, true);
    T r;
    __builtin_memcpy(&r, &b, 4);
    return r;
  }
}
// [1 2 1] along the 16 lanes of a DPP row.  Integers as (prev + c) + (next + c): each addition takes its shifted operand
// straight through the DPP path of v_add_u32 (three instructions instead of two moves, a shift and an add3).
// (The empty asm statements keep the two sums apart: left alone the compiler regroups them into prev + next + 2 c.)
__device__ __forceinline__ uint32_t blur3_across_lanes(uint32_t c) {
  uint32_t below = row_neighbour(c, true) + c, above = row_neighbour(c, false) + c;
  asm("" : "+v"(below));
  asm("" : "+v"(above));
  return below + above;
}
__device__ __forceinline__ double blur3_across_lanes(double c) { return blur3(row_neighbour(c, true), c, row_neighbour(c, false)); }
__device__ __forceinline__ float blur3_across_lanes(float c) { return blur3(row_neighbour(c, true), c, row_neighbour(c, false)); }
constexpr int BT = 12, BR = BT + 4;  // blur tiles: 16^3 cells per tile, 12^3 of them final
#if defined(A3D_DIAGNOSTICS) && defined(A3D_BLUR_STAMPS)
// s_memtime stamps (shader clock) of wave 0 of block 0 of frame 0 in blur_fused_kernel's walk, tile by tile (8 stamps per
// tile, the first 8 tiles): scripts/blur_stamps.py via a3d_debug_blur_stamps.
__device__ unsigned long long g_blur_stamps[80];
__device__ int g_blur_tile;
#define A3D_BSTAMP(k)                                                                                                      \
  do {                                                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && g_blur_tile < 8) g_blur_stamps[8 * g_blur_tile + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define A3D_BSTAMP(k) do { } while (0)
#endif

// value / weight of a blurred cell: both are integers times 4^6 — 0 <= value < 2^48, 1 <= weight < 2^32 — so neither the
// operands nor the quotient come near the ends of the f64 range, and the compiler's correctly rounded division
// (v_div_scale x 2, v_rcp, four fma, v_mul, fma, v_div_fmas, v_div_fixup: twelve instructions, a quarter of the blur's
// last phase) reduces to its arithmetic core: the two Newton steps on the reciprocal, the quotient and its one fma
// correction — the same operations on the same (unscaled, since the scale factors are 1 in this range) operands, eight
// instructions, the same bits.  The selections the fix-up makes (zero, infinite or NaN operands) do not occur.
__device__ __forceinline__ double quotient_of_counts(double value, double weight) {
  const double r0 = __builtin_amdgcn_rcp(weight);
  const double e0 = __builtin_fma(-weight, r0, 1.0);
  const double r1 = __builtin_fma(r0, e0, r0);
  const double e1 = __builtin_fma(-weight, r1, 1.0);
  const double r2 = __builtin_fma(r1, e1, r1);
  const double q = value * r2;
  const double rem = __builtin_fma(-weight, q, value);
  return __builtin_fma(rem, r2, q);
}

// grid.rs:37-56 on the device (same f64 arithmetic as the host path), plus the capacity check, for a batch:
// grid = (blocks, frames).  Every block reduces its frame's (min, max) partials and sizes the grid; block 0 writes the
// frame's scalar block (all of it: nothing is cleared beforehand); all blocks fill the frame's colour -> channel table
// (grid.rs:74: floor((v - min) / sigma_color + 0.5) + 2 for each of the 65 536 values of v, in the splat's own f64
// arithmetic): the splat then looks a channel up instead of converting u16 -> f64 -> usize per pixel (conversions run at
// a quarter of the f64 rate).
// `part` of `parts` blocks (or the only one) of frame blockIdx.y: the shared body of dims_table_kernel and of the fused tail
// of minmax_dims_kernel.
__device__ __forceinline__ void dims_table_body(uint32_t cmin, uint32_t cmax, uint32_t part, uint32_t parts, uint32_t* __restrict__ sc,
                                                uint32_t w, uint32_t h, double sigma_space, double sigma_color,
                                                unsigned long long capacity_cells, uint32_t* __restrict__ channel_of, double inv_sc) {
  const uint32_t gh = (uint32_t)((double)(h - 1) / sigma_space) + 1 + 4;
  const uint32_t gw = (uint32_t)((double)(w - 1) / sigma_space) + 1 + 4;
  const uint32_t gd = (uint32_t)((double)(cmax - cmin) / sigma_color) + 1 + 4;
  // (a grid beyond 32-bit cell offsets is refused like one beyond the capacity: the host reports it, frame.hip)
  const bool too_big = (unsigned long long)gh * gw * gd > capacity_cells || !grid_fits_idx32(gh, gw, gd);
  if (part == 0 && threadIdx.x < SC_STRIDE) {
    uint32_t v = 0;  // SC_OVERFLOW, SC_TICKET, the accumulators, SC_NLIST, SC_NZERO and the unused words start at zero
    switch (threadIdx.x) {
      case SC_MIN: v = cmin; break;
      case SC_MAX: v = cmax; break;
      case SC_GH: v = gh; break;
      case SC_GW: v = gw; break;
      case SC_GD: v = gd; break;
      case SC_TOO_BIG: v = too_big ? 1u : 0u; break;
      default: break;
    }
    sc[threadIdx.x] = v;
  }
  if (too_big) return;
  // (only the colours the frame holds, cmin .. cmax, and 0 — an invalid pixel, looked up but not used — are ever read)
  if (part == 0 && threadIdx.x == 0 && cmin > 0) channel_of[blockIdx.y * 65536u] = 0u;
  for (uint32_t v = cmin + part * 256u + threadIdx.x; v <= cmax; v += parts * 256u)
    channel_of[blockIdx.y * 65536u + v] = f64_as_usize((double)(v - cmin) * inv_sc + 0.5) + 2;
}
__global__ void __launch_bounds__(256)
    dims_table_kernel(const uint32_t* __restrict__ partials, uint32_t n_partials, uint32_t* __restrict__ scal, uint32_t w,
                      uint32_t h, double sigma_space, double sigma_color, unsigned long long capacity_cells,
                      uint32_t* __restrict__ channel_of, double inv_sc) {
  __shared__ uint32_t s_mm[2];
  uint32_t* sc = scal + blockIdx.y * SC_STRIDE;
  if (threadIdx.x < 64) {  // n_partials <= 64
    uint32_t mi = 0xFFFFu, ma = 0u;
    if (threadIdx.x < n_partials)
      mi = partials[(blockIdx.y * n_partials + threadIdx.x) * 2], ma = partials[(blockIdx.y * n_partials + threadIdx.x) * 2 + 1];
    for (int off = 32; off >= 1; off >>= 1) {
      mi = min(mi, (uint32_t)__shfl_xor((int)mi, off, 64));
      ma = max(ma, (uint32_t)__shfl_xor((int)ma, off, 64));
    }
    if (threadIdx.x == 0) s_mm[0] = mi, s_mm[1] = ma;
  }
  __syncthreads();
  dims_table_body(s_mm[0], s_mm[1], blockIdx.x, gridDim.x, sc, w, h, sigma_space, sigma_color, capacity_cells, channel_of, inv_sc);
}

// The last kernel of a filter enqueue: puts back the zeros.  The splat left, per (row, column) of the grid, the range of
// channels it added into; those cells and the frame's tile flags are the only non-zero bytes of the packed grid and the
// flag array, so zeroing them restores "everything is zero" for the next enqueue — 60 KB of extents read and ~100 KB
// written per frame instead of clearing 9.6 MB of cells of which 1 % are used.  Thread = the splat's (row, column).
template <typename CELL>
__global__ void __launch_bounds__(256)
    unsplat_kernel(CELL* __restrict__ grid, const uint32_t* __restrict__ dyn, unsigned long long capacity,
                   const uint2* __restrict__ extent, uint32_t columns, uint32_t* __restrict__ tile_flags, uint32_t flags_stride) {
  dyn += blockIdx.y * SC_STRIDE;
  grid += blockIdx.y * capacity;
  GridDims g;
  if (!dyn_dims(dyn, &g, nullptr)) return;  // (its splat returned as well: nothing to put back)
  const uint32_t id = blockIdx.x * 256u + threadIdx.x;
  for (uint32_t k = id; k < flags_stride / 4; k += gridDim.x * 256u) tile_flags[blockIdx.y * (flags_stride / 4) + k] = 0u;
  const uint32_t cols = g.gw - 3, total = (g.gh - 3) * cols;
  if (id >= total) return;
  const uint2 e = extent[(size_t)blockIdx.y * columns + id];
  const uint32_t column = __umul24(__umul24(id / cols + 2, g.gw) + id % cols + 2, g.gd);
  for (uint32_t ch = e.x; ch <= e.y; ++ch)  // (an untouched column has e.x > e.y)
    *(CELL __attribute__((address_space(1)))*)((a3d_gptr)grid + (column + ch) * (uint32_t)sizeof(CELL)) = (CELL)0;
}

// The splat as a GATHER: a thread owns one (row, column) of the grid — all its channels — and visits the pixels that
// splat into it: the 4-5 image rows r with floor(r / sigma + 0.5) + 2 == its grid row (grid.rs:60-78) times the 4-5
// columns likewise, found with the splat's own f64 expression.  Nobody else touches its cells, so the sums need no
// global atomics (no same-address chains at the L2), and neighbouring threads read neighbouring pixels (coalesced).
// Integer adds: any grouping is exact.  Also marks the blur tiles whose window holds a cell it wrote, and leaves the
// range of channels it wrote for unsplat_kernel.
//
// Per patch of UR x UC pixels: every pixel load is issued before the first is used, then every colour -> channel lookup
// (a dependent L2 round trip each: one at a time they were most of this kernel), then the sums are collected in the
// thread's own SLOTS LDS words — slot = channel - lowest channel of the patch, one ds_add per pixel, conflict-free
// (word q * 256 + thread) — and written out once.  A patch whose channels span more than SLOTS (a depth edge) takes
// another round from the lowest channel not yet written; rounds are wave-uniform loops.  The earlier form kept four
// (channel, sum) pairs in registers and flushed them to the grid whenever a fifth channel appeared: a global
// read-modify-write round trip in the middle of the pixel loop for the whole wave, several times per wave on real
// frames — 35 us per 16 frames against ~10 for this one.
template <typename CELL>
__global__ void __launch_bounds__(256)
    splat_packed_kernel(const uint16_t* __restrict__ img, uint32_t w, uint32_t h, const uint32_t* __restrict__ row_starts,
                        const uint32_t* __restrict__ col_starts, double inv_sc, uint32_t color_min, GridDims g,
                        CELL* __restrict__ grid, const uint32_t* __restrict__ dyn, unsigned long long capacity,
                        uint8_t* __restrict__ tile_flags, uint32_t flags_stride, const uint32_t* __restrict__ channel_of,
                        uint2* __restrict__ extent, uint32_t columns) {
  if (dyn) dyn += blockIdx.y * SC_STRIDE;  // blockIdx.y = frame of a batch
  img += (size_t)blockIdx.y * w * h;
  grid += blockIdx.y * capacity;
  if (tile_flags) tile_flags += (size_t)blockIdx.y * flags_stride;
  if (channel_of) channel_of += blockIdx.y * 65536u;
  if (!dyn_dims(dyn, &g, &color_min)) return;
  const uint32_t ty = (g.gw + BT - 1) / BT, tz = (g.gd + BT - 1) / BT, tx = (g.gh + BT - 1) / BT;
  // pixels splat into grid rows 2 .. gh - 2 and columns 2 .. gw - 2 (floor(x / sigma + 0.5) <= floor(x / sigma) + 1):
  // one thread per such (row, column)
  const uint32_t cols = g.gw - 3, total = (g.gh - 3) * cols, first_id = blockIdx.x * 256u, id = first_id + threadIdx.x;
  if (first_id >= total) return;
  // Tile marks go through LDS first: the threads of a frame write ~500 000 marks to the ~1 700 flag bytes of its
  // tiles, and that many stores to a dozen cache lines queue up at the L2 (they, not the sums, bounded this kernel).
  // The block's marks fall into a window of tile rows [ia, ib] x all tile columns x all channel tiles = a contiguous
  // range of the flag array; each set entry is stored once per block.  (Windows beyond MARKS entries: direct stores.)
  constexpr uint32_t MARKS = 8192;
  __shared__ uint8_t s_marks[MARKS];
  constexpr uint32_t SLOTS = 8;
  __shared__ CELL s_slot[SLOTS][256];
  const uint32_t last_id = min(first_id + 255u, total - 1u);
  const uint32_t ra = (first_id / cols + 2) / BT, rb = (last_id / cols + 2) / BT;
  const uint32_t ia = ra > 0 ? ra - 1 : 0u, ib = min(rb + 1, tx - 1), entries = (ib - ia + 1) * ty * tz;
  const bool marks_in_lds = tile_flags && entries <= MARKS;
  if (marks_in_lds) {
    for (uint32_t e = threadIdx.x; e < entries; e += 256) s_marks[e] = 0;
    __syncthreads();
  }
#pragma unroll
  for (uint32_t q = 0; q < SLOTS; ++q) s_slot[q][threadIdx.x] = 0;  // (a thread's own words: no barrier)
  const bool active = id < total;
  const uint32_t t_row = active ? id / cols : 0u, t_col = active ? id % cols : 0u, gr = t_row + 2, gc = t_col + 2;
  // the image rows / columns that splat into this grid row / column: [starts[t], starts[t + 1]) (splat_starts, host);
  // threads past the end get an empty footprint
  const uint32_t r_lo = row_starts[t_row], r_hi = active ? row_starts[t_row + 1] : r_lo;
  const uint32_t c_lo = col_starts[t_col], c_hi = col_starts[t_col + 1];
  // (32-bit cell and pixel offsets off block-uniform base pointers: the grid is below 2^29 cells — grid_fits_idx32 — and
  // the image below 2^24 pixels; 64-bit index arithmetic was a quarter of this kernel's instructions)
  const uint32_t column = __umul24(__umul24(gr, g.gw) + gc, g.gd);
  auto cell_at = [&](uint32_t ch) { return (CELL __attribute__((address_space(1)))*)((a3d_gptr)grid + (column + ch) * (uint32_t)sizeof(CELL)); };
  constexpr uint32_t NONE = 0xFFFFFFFFu;
  uint32_t ch_lo = NONE, ch_hi = 0u;  // the channels this (row, column) adds into: what unsplat_kernel zeroes again
  // The blur tiles whose 16^3 window (12^3 tile + 2 cells of halo) contains a cell of this column: rows and columns of
  // tiles are the thread's own (a0..a1 x b0..b1); the channel tiles depend on the cells and are collected as bits of
  // `ztiles` (grids of up to 64 channel tiles, i.e. 768 channels; deeper grids mark as they write), marked once at the end.
  // WHICH tiles (round 6).  The slice reads, for a pixel, the eight cells around its grid position: each within ONE cell
  // of the pixel's own splat cell on every axis (grid.rs:60-78 against :132-146: floor(t + 0.5) against floor(t), floor(t)
  // + 1).  So a blurred cell is only ever read if a splat cell lies within one cell of it, and a tile's 12^3 cells only if
  // a splat cell lies in the 14^3 box around them (MARGIN 1) — not in the whole 16^3 window the tile loads (margin 2, what
  // rounds 3-5 marked: 340 instead of 293 tiles on the benchmark's frames).  A marked tile still loads its whole window, so
  // what it writes is complete.  The exception are the zero pixels, which are sliced but not splatted: they read channels
  // 2 and 3 (colour minimum 0) of the FIRST channel tile, where a splat cell two cells away still leaves a non-zero value:
  // first-channel tiles keep margin 2 (and are written as zeros when unmarked, blur_fused_kernel).
  const uint32_t ta = gr / BT, tb = gc / BT, la = gr % BT, lb = gc % BT;
  const uint32_t a0 = (la < 1 && ta > 0) ? ta - 1 : ta, a1 = (la >= BT - 1 && ta + 1 < tx) ? ta + 1 : ta;
  const uint32_t b0 = (lb < 1 && tb > 0) ? tb - 1 : tb, b1 = (lb >= BT - 1 && tb + 1 < ty) ? tb + 1 : tb;
  const uint32_t a0w = (la < 2 && ta > 0) ? ta - 1 : ta, a1w = (la >= BT - 2 && ta + 1 < tx) ? ta + 1 : ta;
  const uint32_t b0w = (lb < 2 && tb > 0) ? tb - 1 : tb, b1w = (lb >= BT - 2 && tb + 1 < ty) ? tb + 1 : tb;
  const bool ztiles_fit = tz <= 64;
  unsigned long long ztiles = 0;
  auto mark = [&](uint32_t z) {
    const uint32_t i0 = z == 0 ? a0w : a0, i1 = z == 0 ? a1w : a1, j0 = z == 0 ? b0w : b0, j1 = z == 0 ? b1w : b1;
    for (uint32_t i = i0; i <= i1; ++i)
      for (uint32_t j = j0; j <= j1; ++j) {
        if (marks_in_lds) s_marks[((i - ia) * ty + j) * tz + z] = 1;
        else tile_flags[(i * ty + j) * tz + z] = 1;
      }
  };
  // (unconditional lookups — a conditional load is compiled into a branch with its own wait, one lookup at a time:
  // without a table the loads read word 0 of the row table and the channel is computed instead)
  const uint32_t* table = channel_of ? channel_of : row_starts;
  const uint32_t table_mask = channel_of ? 0xFFFFu : 0u;
  constexpr uint32_t UR = 5, UC = 6;  // the default sigma's footprint, 4-5 x 4-5 pixels, is one patch
  bool wrote_before = false;          // an earlier patch of this thread may have written the same channels
  const uint32_t own_word = threadIdx.x * (uint32_t)sizeof(CELL);
  for (uint32_t r0 = r_lo; r0 < r_hi; r0 += UR)
    for (uint32_t c0 = c_lo; c0 < c_hi; c0 += UC) {
      uint32_t v[UR * UC], chv[UR * UC];
#pragma unroll
      for (uint32_t i = 0; i < UR; ++i)
#pragma unroll
        for (uint32_t k = 0; k < UC; ++k) {
          const bool in = r0 + i < r_hi && c0 + k < c_hi;
          const uint32_t px = in ? __umul24(r0 + i, w) + (c0 + k) : __umul24(r_lo, w) + c_lo;  // (unconditional load)
          const uint32_t got = *(const uint16_t __attribute__((address_space(1)))*)((a3d_gptr_c)img + px * 2u);
          v[i * UC + k] = in ? got : 0u;  // `color <= I::min_value()` (:67) is skipped: pixels outside the footprint count as 0
        }
#pragma unroll
      for (uint32_t j = 0; j < UR * UC; ++j)
        chv[j] = *(const uint32_t __attribute__((address_space(1)))*)((a3d_gptr_c)table + (v[j] & table_mask) * 4u);
      uint32_t base = NONE;
#pragma unroll
      for (uint32_t j = 0; j < UR * UC; ++j) {
        if (!channel_of) chv[j] = f64_as_usize((double)(v[j] - color_min) * inv_sc + 0.5) + 2;  // (uniform branch)
        chv[j] = v[j] != 0 ? chv[j] : NONE;  // a skipped pixel: in no round, never a candidate for the next one
        base = min(base, chv[j]);
        ch_hi = v[j] != 0 ? max(ch_hi, chv[j]) : ch_hi;
      }
      ch_lo = min(ch_lo, base);
      while (__builtin_amdgcn_ballot_w64(base != NONE)) {  // rounds of SLOTS channels from `base` up
        const uint32_t lim = base == NONE ? NONE : base + SLOTS;  // (a lane that is done takes part with nothing)
        uint32_t next = NONE;
#pragma unroll
        for (uint32_t j = 0; j < UR * UC; ++j) {
          const uint32_t q = chv[j] - base;
          const bool mine = q < SLOTS && chv[j] != NONE && lim != NONE;  // (q wraps for channels below base: written in an earlier round)
          const CELL add = mine ? ((CELL)v[j] << Pack<CELL>::SHIFT) + (CELL)1 : (CELL)0;
          // ds_add without return into the thread's own word of slot q; nobody else uses it
          atomicAdd((CELL*)((char*)&s_slot[0][0] + ((mine ? q : 0u) * (256u * (uint32_t)sizeof(CELL)) + own_word)), add);
          next = min(next, chv[j] >= lim ? chv[j] : NONE);
        }
        CELL sum[SLOTS], old[SLOTS];
#pragma unroll
        for (uint32_t q = 0; q < SLOTS; ++q) sum[q] = s_slot[q][threadIdx.x];
#pragma unroll
        for (uint32_t q = 0; q < SLOTS; ++q) old[q] = (wrote_before && sum[q] != 0) ? *cell_at(base + q) : (CELL)0;
#pragma unroll
        for (uint32_t q = 0; q < SLOTS; ++q) {
          if (sum[q] != 0) {  // this thread's cell: nobody else reads or writes it (zero before the splat: a3d_context::grid_clean)
            const uint32_t ch = base + q;
            *cell_at(ch) = old[q] + sum[q];
            s_slot[q][threadIdx.x] = 0;
            if (tile_flags) {
              const uint32_t tc = ch / BT, lc = ch - tc * BT;
              // (margin 1 along the channels as well; the first channel tile from two cells away: channels 12 and 13)
              const uint32_t z0 = ((lc < 1 || (tc == 1 && lc < 2)) && tc > 0) ? tc - 1 : tc,
                             z1 = (lc >= BT - 1 && tc + 1 < tz) ? tc + 1 : tc;
              if (ztiles_fit) ztiles |= (1ull << z0) | (1ull << z1);  // (one of the two is tc itself)
              else mark(z0), mark(z1);
            }
          }
        }
        base = next;
      }
      wrote_before = true;
    }
  while (ztiles) {  // (a thread's cells lie in two or three channel tiles)
    const uint32_t z = (uint32_t)__builtin_ctzll(ztiles);
    ztiles &= ztiles - 1;
    mark(z);
  }
  if (extent && active) extent[(size_t)blockIdx.y * columns + id] = ch_lo <= ch_hi ? make_uint2(ch_lo, ch_hi) : make_uint2(1u, 0u);
  if (marks_in_lds) {
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < entries; e += 256)
      if (s_marks[e]) tile_flags[ia * ty * tz + e] = 1;
  }
}

// BilateralGrid::normalize (grid.rs:90-104) after the pass-per-launch blur: (value, weight) cells -> normalised values.
__global__ void __launch_bounds__(256)
    normalize_kernel(const double2* __restrict__ in, double* __restrict__ out, unsigned long long cells) {
  const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
  if (i < cells) out[i] = normalized_cell(in[i].x, in[i].y);
}

// The six blur passes (axis 0 twice, axis 1 twice, axis 2 twice; edge_aware_filter.rs:68-114) on one
// (T+4)^3 tile held in LDS: each pass needs its neighbours along one axis only, so after the two passes of an axis
// the two outermost layers of the tile along that axis are stale and the central T^3 cells are exactly what six
// full-grid passes produce.  A pass writes rows 1..gh-2, cols 1..gw-2, channels 0..gd-2 and nothing else; every
// other cell is zero in both of the reference's buffers forever, which is what `interior ? blur : 0` reproduces
// (cells outside the grid count as such zeros).  One read of the packed grid, one write of the blurred grid:
// 24 B of HBM traffic per cell instead of 6 x 32 B.
constexpr int BZP = BR;               // z pitch: every LDS access of the tile has the channel axis across lanes
constexpr int BCELLS = BR * BR * BZP;  // x 16 B = 64 KiB of LDS

// Two passes of the [1 2 1] blur along one 16-cell line held in registers, WITHOUT the reference's division by four:
// the grid starts as integers below 2^40 (u16 sums of fewer than 2^24 pixels), so every value the reference's six
// passes produce is an integer multiple of 4^-k below 2^40 — exact in f64 — and (prev + next + 2 cur) * 0.25 never
// rounds.  The kernel therefore carries 4^k times the reference's values (integers below 2^52, equally exact) and
// scales by 2^-12 once, when it writes the cell: bit-identical, at half the f64 instructions.
// `ok(i)`: cell i is one the reference writes (otherwise it is zero after every pass).  MASKED = false: the tile
// lies inside the written box on this axis and the other two, ok is true everywhere (no selects).
template <bool MASKED, typename V, typename W, typename OK>
__device__ __forceinline__ void blur_line_twice(V (&vx)[BR], W (&vw)[BR], OK ok) {
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    // left of the tile: stale layers, never part of the final 12^3 (at grid channel 0 the reference's "previous" read
    // aliases an always-zero cell: also zero)
    V prev_x = (V)0;
    W prev_w = (W)0;
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const V cur_x = vx[i], next_x = i + 1 < BR ? vx[i + 1] : (V)0;
      const W cur_w = vw[i], next_w = i + 1 < BR ? vw[i + 1] : (W)0;
      V ox = blur3(prev_x, cur_x, next_x);  // exact: same as (p + n) + 2 c in the reference's f64
      W ow = blur3(prev_w, cur_w, next_w);
      if (MASKED && !ok(i)) ox = (V)0, ow = (W)0;
      vx[i] = ox, vw[i] = ow;
      prev_x = cur_x, prev_w = cur_w;
    }
  }
}

// The 16 cells a thread owns in a tile's first pass — the rows of (column hi, channel lo) of the 16^3 window — straight
// from the packed grid (zero outside it).  Apart from blur_tile so that a block walking a list of tiles can have the
// NEXT tile's sixteen loads in flight while it computes the current one: with four blocks of four waves per CU there is
// one wave per SIMD and block to hide a tile's initial load latency behind, i.e. nothing.
template <typename CELL>
__device__ __forceinline__ void load_window(uint32_t tile_id, const CELL* __restrict__ packed, const GridDims g, CELL (&u)[BR]) {
  const uint32_t tz = (g.gd + BT - 1) / BT, ty = (g.gw + BT - 1) / BT;
  const int r0 = (int)(tile_id / (ty * tz)) * BT - 2, c0 = (int)((tile_id / tz) % ty) * BT - 2,
            z0 = (int)(tile_id % tz) * BT - 2;
  const int gh = (int)g.gh, gw = (int)g.gw, gd = (int)g.gd;
  const int t = (int)threadIdx.x, hi = t >> 4, lo = t & 15;
  // Sixteen UNCONDITIONAL loads at clamped coordinates (round 6; a load under `if (inside the grid)` is a branch of its
  // own, sixteen of them per tile).  A cell outside the grid counts as zero — and the cell the clamp lands on IS zero:
  // pixels splat into rows 2 .. gh - 2, columns 2 .. gw - 2 and channels 2 .. gd - 2 only (grid.rs:60-78: floor(t + 0.5) + 2
  // against gd = floor(t_max) + 5), the rest of the packed grid keeps the zeros it starts with (a3d_context::grid_clean),
  // so row 0 / gh - 1, column 0 / gw - 1 and channel 0 / gd - 1 are what "outside" reads.
  const int gc = min(max(c0 + hi, 0), gw - 1), gz = min(max(z0 + lo, 0), gd - 1);
  // cell (gr, gc, gz) = row part (block-uniform) + lane part: 32-bit arithmetic off the frame's (block-uniform) grid pointer —
  // the fused kernels only see grids below 2^29 cells (grid_fits_idx32) — instead of a 64-bit multiply-add per load
  const uint32_t row_stride = (uint32_t)(gw * gd) * (uint32_t)sizeof(CELL), lane_part = (uint32_t)(gc * gd + gz) * (uint32_t)sizeof(CELL);
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const uint32_t gr = (uint32_t)min(max(r0 + i, 0), gh - 1);
    u[i] = *(const CELL __attribute__((address_space(1)))*)((a3d_gptr_c)packed + (gr * row_stride + lane_part));
  }
}

// One tile.  `known_occupied`: the tile comes from the splat's list of marked windows; otherwise emptiness is decided
// from the loaded window `u` (load_window).  `zeros_only`: an unmarked first-channel tile, written as zeros (see below).
// `nx` (TAKE_NEXT): the NEXT tile's window, in flight since before this tile's first pass (load_window); it is moved into
// `u` between this tile's last arithmetic and its stores — see "the order of waits" at the stores.
template <typename CELL, bool TAKE_NEXT = false>
__device__ __forceinline__ void blur_tile(typename Pack<CELL>::EarlyValue* tile_x, typename Pack<CELL>::EarlyWeight* tile_w,
                                          uint32_t tile_id, CELL (&u)[BR], const CELL (&nx)[TAKE_NEXT ? BR : 1],
                                          const GridDims g, double* __restrict__ out, bool known_occupied,
                                          bool zeros_only) {
  const uint32_t tz = (g.gd + BT - 1) / BT, ty = (g.gw + BT - 1) / BT;
  const int r0 = (int)(tile_id / (ty * tz)) * BT - 2, c0 = (int)((tile_id / tz) % ty) * BT - 2,
            z0 = (int)(tile_id % tz) * BT - 2;
  const int gh = (int)g.gh, gw = (int)g.gw, gd = (int)g.gd;
  const int t = (int)threadIdx.x, hi = t >> 4, lo = t & 15;
  // the first channel tile of an empty window is written as zeros (see below), other empty tiles are left alone
  auto write_zeros = [&]() {
    for (int l = t; l < BT * BT * BT; l += 256) {
      const int gr = r0 + 2 + l / (BT * BT), gc2 = c0 + 2 + (l / BT) % BT, gz2 = l % BT;
      if (gr < gh && gc2 < gw && gz2 < gd) out[((size_t)gr * gw + gc2) * gd + gz2] = 0.0;
    }
  };
  if (zeros_only) {
    write_zeros();
    return;
  }
  static_assert(!TAKE_NEXT || sizeof(CELL) == 4, "the prefetching walk is the narrow cells' (registers)");
  A3D_BSTAMP(0);  // tile entered (the next window's loads are out)
  auto at = [](int lr, int lc, int lz) { return (lr * BR + lc) * BZP + lz; };
  auto row_ok = [&](int gr) { return gr >= 1 && gr <= gh - 2; };
  auto col_ok = [&](int gc) { return gc >= 1 && gc <= gw - 2; };
  auto chan_ok = [&](int gz) { return gz >= 0 && gz <= gd - 2; };
  // a tile whose 16^3 window lies inside the box of cells the reference writes needs no masks (block-uniform)
  const bool inside = r0 >= 1 && r0 + BR - 1 <= gh - 2 && c0 >= 1 && c0 + BR - 1 <= gw - 2 && z0 >= 0 && z0 + BR - 1 <= gd - 2;
  typedef typename Pack<CELL>::EarlyValue EV;
  typedef typename Pack<CELL>::EarlyWeight EW;
  typedef typename Pack<CELL>::Weight W;
  // ---- axis 0: thread = (column hi, channel lo) owns the 16 rows; read straight from the packed grid -------
  {
    EV vx[BR];
    EW vw[BR];
    const int gc = c0 + hi, gz = z0 + lo;
#pragma unroll
    for (int i = 0; i < BR; ++i)
      vx[i] = (EV)(u[i] >> Pack<CELL>::SHIFT), vw[i] = (EW)(u[i] & (((CELL)1 << Pack<CELL>::SHIFT) - 1));
    // Empty windows: a depth image occupies ~1 % of its grid's cells and 20-30 % of its tiles.  A tile whose whole
    // 16^3 window holds no splat blurs to zero, and the slice never reads it: a pixel's eight cells lie within one cell
    // of its own splat cell on every axis (grid.rs:60-78 vs :132-146: floor(t + 0.5) against floor(t), floor(t) + 1),
    // hence inside a tile whose window contains that splat.  The exception are the zero pixels, which are sliced but
    // not splatted; they read channels 2 and 3 (colour minimum = 0), i.e. the first channel tile: those tiles are
    // always written (zeros when empty).  Skipped tiles keep stale cells that nothing reads.
    bool mine = false;
#pragma unroll
    for (int i = 0; i < BR; ++i) mine |= (vx[i] != (EV)0) | (vw[i] != (EW)0);
    if (!known_occupied && !__syncthreads_or(mine ? 1 : 0)) {  // (no list from the splat: decided from the loaded window)
      if (z0 == -2) write_zeros();
      return;
    }
    const bool line_ok = col_ok(gc) && chan_ok(gz) && gz < gd;
    if (inside) blur_line_twice<false>(vx, vw, [](int) { return true; });
    else blur_line_twice<true>(vx, vw, [&](int i) { return line_ok && row_ok(r0 + i); });
    // ---- axis 2 (channels) in the same layout: the 16 channels of a (row, column) line sit in the 16 lanes of one
    // DPP row, so "previous" and "next" are row shifts by one lane (zero shifted in at the ends of the window: stale
    // layers there, and the reference's aliased always-zero "previous" at grid channel 0).  The axes commute exactly:
    // every intermediate grid is zero outside the box of written cells, and the arithmetic is exact.
    // (round 6: only the 12 central ROWS go on — the two outermost on either side are stale after the row passes and nothing
    // reads them; the row passes above are straight-line code, so what only feeds those four rows is not computed either)
    {
      const bool lane_ok = col_ok(gc) && chan_ok(gz);
#pragma unroll
      for (int i = 2; i < BR - 2; ++i) {
        const bool ok = inside || (lane_ok && row_ok(r0 + i));
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const EV cx_ = vx[i];
          const EW cw_ = vw[i];
          const EV ox = blur3_across_lanes(cx_);
          const EW ow = blur3_across_lanes(cw_);
          vx[i] = ok ? ox : (EV)0, vw[i] = ok ? ow : (EW)0;
        }
      }
    }
    A3D_BSTAMP(1);  // row and channel passes done
#pragma unroll
    for (int i = 2; i < BR - 2; ++i) tile_x[at(i, hi, lo)] = vx[i], tile_w[at(i, hi, lo)] = vw[i];
  }
  __syncthreads();
  A3D_BSTAMP(2);  // window in LDS, barrier passed
  // ---- axis 1: thread = (row, channel lo) owns the 16 columns; the central 12^3 cells go straight to the grid.  Round 6:
  // the 12 central rows on the block's first 192 threads (row = 2 + t / 16): three waves do what four did with half of the
  // first and of the last wave's lanes on stale rows; the fourth wave goes on to the tile loop's barrier. ----
#ifndef A3D_BLUR_3WAVES
#define A3D_BLUR_3WAVES 1
#endif
  double o[BT];  // the thread's twelve normalised cells
  bool stores = false;
  const int gr2 = r0 + hi + (A3D_BLUR_3WAVES ? 2 : 0), gz2 = z0 + lo;
  if (A3D_BLUR_3WAVES ? hi < BR - 4 : (hi >= 2 && hi < BR - 2)) {
    constexpr int SHIFT = A3D_BLUR_3WAVES ? 2 : 0;  // window row of thread row `hi`
    double vx[BR];  // (results up to 2^36: f64 from here on; the counts, below 2^20, in Pack<CELL>::Weight)
    W vw[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) vx[i] = (double)tile_x[at(hi + SHIFT, i, lo)], vw[i] = (W)tile_w[at(hi + SHIFT, i, lo)];
    A3D_BSTAMP(3);  // the thread's 16 columns read back and converted
    const bool line_ok = row_ok(gr2) && chan_ok(gz2);
    if (inside) blur_line_twice<false>(vx, vw, [](int) { return true; });
    else blur_line_twice<true>(vx, vw, [&](int i) { return line_ok && col_ok(c0 + i); });
    A3D_BSTAMP(4);  // column passes done
    stores = lo >= 2 && lo < BR - 2 && gr2 < gh && gz2 < gd;
    // normalised (grid.rs:90-104): value / weight — the common factor 4^6 cancels exactly — or, where the weight is zero,
    // the value itself (x 4^-6: the six divisions by four)
    // Four quotients at a time, stage by stage and without a branch (round 6): written as `w > 0 ? value / w : value x 2^-12`
    // per cell, each of the twelve became its own divergent if / else around a chain of nine dependent f64 operations —
    // 2 700 cycles of a tile's 26 000 (profiles/round6_blur_tile_stamps.txt).  The quotient is taken for every lane over a
    // denominator that is never zero and the result selected: the same values, the chains of four cells interleaved.
#pragma unroll
    for (int i0 = 2; i0 < BR - 2; i0 += 4) {
      double wd[4], r[4], e[4], q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) wd[k] = (double)(vw[i0 + k] > (W)0 ? vw[i0 + k] : (W)1), r[k] = __builtin_amdgcn_rcp(wd[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = __builtin_fma(-wd[k], r[k], 1.0);
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = __builtin_fma(r[k], e[k], r[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = __builtin_fma(-wd[k], r[k], 1.0);
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = __builtin_fma(r[k], e[k], r[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = vx[i0 + k] * r[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = __builtin_fma(-wd[k], q[k], vx[i0 + k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = __builtin_fma(e[k], r[k], q[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[i0 + k - 2] = vw[i0 + k] > (W)0 ? q[k] : vx[i0 + k] * 0x1p-12;
    }
  }
  // THE ORDER OF WAITS (round 6).  Loads and stores share one in-order counter (vmcnt), and the compiler cannot know how many
  // of this tile's stores a wave issues (grid borders, the idle fourth wave), so a wait for the next window's loads placed
  // AFTER the stores is a wait for the stores' completion as well: every tile ended with a store round trip to memory, and
  // the "prefetch" was waited for before the first pass (the ISA had s_waitcnt vmcnt(0) at both places; the kernel kept the
  // VALU pipe under half busy).  Taking the next window BEFORE the stores waits for loads issued a whole tile ago and for
  // the PREVIOUS tile's stores, both long complete; this tile's stores then drain under the next tile's passes.
  A3D_BSTAMP(5);  // twelve cells normalised
  if constexpr (TAKE_NEXT) {
    // (an empty asm that reads and writes every word of the next window: the loads must have landed HERE — a plain copy the
    // compiler sinks to the loop's latch, behind the stores — and no memory operation moves across)
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      CELL v = nx[i];
      asm volatile("" : "+v"(v) : : "memory");
      u[i] = v;
    }
  }
  A3D_BSTAMP(6);  // next window taken (its loads had landed)
  // (a wave stores 4 rows x 12 channels = four 192-byte runs per column)
  if (stores) {
    const int col_stride = gd * 8;
    int at_col = ((gr2 * gw + c0 + 2) * gd + gz2) * 8;  // byte offset of cell (gr, c0 + i, gz) in the f64 grid
#pragma unroll
    for (int i = 2; i < BR - 2; ++i, at_col += col_stride)
      if (c0 + i < gw) *(double __attribute__((address_space(1)))*)((a3d_gptr)out + (uint32_t)at_col) = o[i - 2];
  }
  A3D_BSTAMP(7);  // stores issued
}

// (Grids of more than BLUR_LIST_MAX tiles only: otherwise the blur's blocks compact the flags themselves.)
// Turns the tile flags the splat wrote into two lists per frame: the marked windows (the blur's work) and the
// unmarked first-channel tiles (written as zeros).  A depth image marks 20-30 % of its tiles; launching one block
// per TILE made the blur a block-dispatch benchmark (each block holds 68 KiB of LDS, two fit a CU, and an empty one
// still waits for its slot): 180 us per 16 frames, of which the arithmetic was a third.
__global__ void __launch_bounds__(256)
    tile_list_kernel(const uint8_t* __restrict__ tile_flags, uint32_t flags_stride, uint32_t* __restrict__ scal,
                     uint32_t* __restrict__ lists) {
  uint32_t* sc = scal + blockIdx.y * SC_STRIDE;
  GridDims g;
  if (!dyn_dims(sc, &g, nullptr)) return;
  const uint32_t tz = (g.gd + BT - 1) / BT, ty = (g.gw + BT - 1) / BT, tx = (g.gh + BT - 1) / BT;
  const uint32_t id = blockIdx.x * 256u + threadIdx.x;
  const bool in = id < tx * ty * tz;
  const bool marked = in && tile_flags[(size_t)blockIdx.y * flags_stride + id] != 0;
  const bool zero = in && !marked && id % tz == 0;
  uint32_t* work = lists + (size_t)blockIdx.y * 2 * flags_stride;
  auto append = [&](bool mine, uint32_t* counter, uint32_t* list) {  // one atomic per wave
    const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
    if (!m) return;
    const uint32_t lane = __lane_id(), leader = (uint32_t)__builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    base = __shfl(base, leader);
    if (mine) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = id;
  };
  append(marked, &sc[SC_NLIST], work);
  append(zero, &sc[SC_NZERO], work + flags_stride);
}

// grid = (blocks, frames).  With `lists` or `tile_flags`: every block walks its frame's lists with stride gridDim.x (the
// launch is sized to the blocks the chip holds at once).  With neither: one block per tile, blockIdx.x = tile.
// `tile_flags` (frames with at most BLUR_LIST_MAX tiles): every block compacts the frame's flags into the two lists
// itself, in LDS — marked tiles in tile order from the front, unmarked first-channel tiles from the back; all blocks of a
// frame derive the same lists (ordered scan, no atomics), so the strided walk covers every entry once.  That is a few
// hundred instructions per block instead of a launch between splat and blur (tile_list_kernel: 4.8 us per 16 frames).
constexpr uint32_t BLUR_LIST_MAX = 3072;
template <typename CELL>
__global__ void __launch_bounds__(256, sizeof(CELL) == 4 ? 4 : 2)  // (waves per SIMD: 128 / 256 registers a thread)
    blur_fused_kernel(const CELL* __restrict__ packed, GridDims g, double* __restrict__ out,
                      uint32_t* __restrict__ dyn, unsigned long long capacity,
                      const uint32_t* __restrict__ lists, uint32_t flags_stride, const uint32_t* __restrict__ tile_flags,
                      uint32_t per_frame) {
  __shared__ typename Pack<CELL>::EarlyValue tile_x[BCELLS];   // 16 KiB (narrow cells) / 32 KiB
  __shared__ typename Pack<CELL>::EarlyWeight tile_w[BCELLS];  // 16 KiB / 32 KiB
  // (frame, rank of this block among the frame's `blocks`).  per_frame != 0: a 1-D launch of frames x per_frame blocks with
  // the frames placed XCD by XCD (round 6, below); otherwise blockIdx.y = frame, blockIdx.x = rank.
#ifndef A3D_BLUR_XCD  // 1: a frame's blocks share an XCD (its L2), 0: frames spread over all eight (the round-3..5 placement)
#define A3D_BLUR_XCD 1
#endif
#ifndef A3D_BLUR_CHUNKED  // 1: a block walks a contiguous piece of the frame's tile list, 0: every blocks-th entry
#define A3D_BLUR_CHUNKED 0
#endif
  uint32_t frame = blockIdx.y, rank = blockIdx.x, blocks = gridDim.x;
  if (per_frame) {
    const uint32_t v = A3D_BLUR_XCD ? xcd_contiguous_index(blockIdx.x, gridDim.x) : blockIdx.x;
    frame = v / per_frame, rank = v - frame * per_frame, blocks = per_frame;
  }
  if (dyn) dyn += frame * SC_STRIDE;
  packed += frame * capacity;
  out += frame * capacity;
  if (!dyn_dims(dyn, &g, nullptr)) return;
  const uint32_t tz = (g.gd + BT - 1) / BT, ty = (g.gw + BT - 1) / BT, tx = (g.gh + BT - 1) / BT;
  if (tile_flags) {
    __shared__ uint16_t s_list[BLUR_LIST_MAX];
    __shared__ uint32_t s_wave[2][4];
    tile_flags += frame * (flags_stride / 4);
    const uint32_t tiles = tx * ty * tz, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t n_work = 0, n_zero = 0;
    for (uint32_t base = 0; base < tiles; base += 1024u) {  // 256 threads x one flag word (four tiles) per round
      const uint32_t first = base + threadIdx.x * 4u;
      const uint32_t word = first < tiles ? tile_flags[first / 4] : 0u;
      uint32_t marked = 0, zero = 0;  // bit b: tile first + b
#pragma unroll
      for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t id = first + b;
        const bool m = ((word >> (8 * b)) & 0xFFu) != 0;
        marked |= (m ? 1u : 0u) << b;
        zero |= ((id < tiles && !m && id % tz == 0) ? 1u : 0u) << b;
      }
      const uint32_t cm = __builtin_popcount(marked), cz = __builtin_popcount(zero);
      uint32_t im = cm, iz = cz;  // inclusive scans over the wave
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = (uint32_t)__shfl_up((int)im, off, 64), c = (uint32_t)__shfl_up((int)iz, off, 64);
        if (lane >= (uint32_t)off) im += a, iz += c;
      }
      if (lane == 63) s_wave[0][wave] = im, s_wave[1][wave] = iz;
      __syncthreads();
      uint32_t at_m = n_work + im - cm, at_z = n_zero + iz - cz;
      for (uint32_t k = 0; k < wave; ++k) at_m += s_wave[0][k], at_z += s_wave[1][k];
#pragma unroll
      for (uint32_t b = 0; b < 4; ++b) {
        if (marked >> b & 1u) s_list[at_m++] = (uint16_t)(first + b);
        if (zero >> b & 1u) s_list[BLUR_LIST_MAX - 1u - at_z++] = (uint16_t)(first + b);
      }
      n_work += s_wave[0][0] + s_wave[0][1] + s_wave[0][2] + s_wave[0][3];
      n_zero += s_wave[1][0] + s_wave[1][1] + s_wave[1][2] + s_wave[1][3];
      __syncthreads();  // the wave totals are read before the next round replaces them; the lists before they are used
    }
    if (rank == 0 && threadIdx.x == 0) dyn[SC_NLIST] = n_work, dyn[SC_NZERO] = n_zero;  // (statistics for the host)
    auto entry = [&](uint32_t j) { return (uint32_t)(j < n_work ? s_list[j] : s_list[BLUR_LIST_MAX - 1u - (j - n_work)]); };
    constexpr bool PREFETCH = sizeof(CELL) == 4;  // (the wide cells' kernel has no registers to spare for a second window)
    if constexpr (PREFETCH) {
      // marked tiles first, the next tile's sixteen loads in flight under this tile; then the zero tiles (their own loop: a
      // loop of stores inside the walk would leave the compiler no count of what is outstanding)
      // WHICH tiles, WHERE (round 6).  A window's lines are shared: consecutive list entries are channel neighbours (tile
      // id = ((row tile, column tile), channel tile), channel fastest) whose 16-channel windows overlap by four channels
      // and split 128-byte lines between them, the next column tile shares four of sixteen columns, the next row tile four
      // rows.  With a frame's blocks on all eight XCDs no L2 ever saw a line twice: the kernel fetched 16.4 MB per frame for
      // 4.8 MB of windows, and its loads and stores queued at issue (250 / 130 cycles each of a tile's 27 000,
      // profiles/round6_blur_tile_stamps.txt).  Now the frame's blocks sit on ONE XCD (the 1-D launch above) and walk the
      // list every blocks-th entry, i.e. at any moment they hold neighbouring tiles: what neighbours share is fetched into
      // that L2 once (109 -> 100.5 us per 32 frames).  A block walking a CONTIGUOUS piece of the list instead — reuse in
      // time rather than between concurrent blocks — was slower than either (112-114 us: A3D_BLUR_CHUNKED).
      CELL cur[BR] = {}, nxt[BR] = {};
      const uint32_t step = A3D_BLUR_CHUNKED ? 1u : blocks;
      uint32_t j = A3D_BLUR_CHUNKED ? (uint32_t)(((unsigned long long)rank * n_work) / blocks) : rank;
      const uint32_t j_end = A3D_BLUR_CHUNKED ? (uint32_t)(((unsigned long long)(rank + 1) * n_work) / blocks) : n_work;
      if (j < j_end) load_window<CELL>(entry(j), packed, g, cur);
      // (the first window has landed before the walk starts: entered with these loads outstanding, the loop's first use of
      // `cur` gets a wait that every later trip pays too — right behind the next window's loads)
#pragma unroll
      for (int i = 0; i < BR; ++i) asm volatile("" : "+v"(cur[i]) : : "memory");
      for (; j < j_end; j += step) {
        if (j + step < j_end) load_window<CELL>(entry(j + step), packed, g, nxt);
        blur_tile<CELL, true>(tile_x, tile_w, entry(j), cur, nxt, g, out, true, false);
        __syncthreads();  // the tile's last reads of the LDS window are done before the next tile overwrites it
#if defined(A3D_DIAGNOSTICS) && defined(A3D_BLUR_STAMPS)
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
          if (g_blur_tile < 8) g_blur_stamps[64 + g_blur_tile] = __builtin_amdgcn_s_memtime();  // behind the tile's last barrier
          ++g_blur_tile;
        }
#endif
      }
      // the unmarked first-channel tiles (written as zeros), shared out the same way
      uint32_t z = A3D_BLUR_CHUNKED ? (uint32_t)(((unsigned long long)rank * n_zero) / blocks) : rank;
      const uint32_t z_end = A3D_BLUR_CHUNKED ? (uint32_t)(((unsigned long long)(rank + 1) * n_zero) / blocks) : n_zero;
      for (; z < z_end; z += step) blur_tile<CELL>(tile_x, tile_w, entry(n_work + z), cur, {}, g, out, true, true);
    } else {
      CELL cur[BR] = {};
      for (uint32_t j = rank; j < n_work + n_zero; j += blocks) {
        const bool work = j < n_work;
        if (work) load_window<CELL>(entry(j), packed, g, cur);
        blur_tile<CELL>(tile_x, tile_w, entry(j), cur, {}, g, out, true, !work);
        __syncthreads();  // the tile's last reads of the LDS window are done before the next tile overwrites it
      }
    }
    return;
  }
  if (!lists) {
    // 1-D launch (the host may not know the dimensions): block -> tile (row, column, channel), channel fastest
    if (blockIdx.x < tx * ty * tz) {
      CELL u[BR];
      load_window<CELL>(blockIdx.x, packed, g, u);
      blur_tile<CELL>(tile_x, tile_w, blockIdx.x, u, {}, g, out, false, false);
    }
    return;
  }
  lists += (size_t)blockIdx.y * 2 * flags_stride;
  const uint32_t n_work = dyn[SC_NLIST], n_zero = dyn[SC_NZERO];
  for (uint32_t j = blockIdx.x; j < n_work + n_zero; j += gridDim.x) {  // (grids of more than BLUR_LIST_MAX tiles: no prefetch)
    const bool work = j < n_work;
    const uint32_t tile = work ? lists[j] : lists[flags_stride + (j - n_work)];
    CELL u[BR] = {};
    if (work) load_window<CELL>(tile, packed, g, u);
    blur_tile<CELL>(tile_x, tile_w, tile, u, {}, g, out, true, !work);
    __syncthreads();  // the tile's last reads of the LDS window are done before the next tile overwrites it
  }
}

// BilateralGrid::slice (grid.rs:106-130): every pixel, zeros included; num::cast::<f64,u16>
// (blockIdx.y = image of a batch — [images][h][w] in and out, grids `capacity` cells apart, scalar blocks SC_STRIDE words
// apart, each image's overflow flag in its own scalar block: a3d_bilateral_filter_u16_device; one image otherwise)
__global__ void __launch_bounds__(256)
    slice_kernel(const uint16_t* __restrict__ img, uint32_t w, uint32_t h, double inv_ss, double inv_sc,
                 uint32_t color_min, GridDims g, const double* __restrict__ grid, uint16_t* __restrict__ out,
                 uint32_t* __restrict__ overflow_flag, const uint32_t* __restrict__ dyn, unsigned long long capacity) {
  if (blockIdx.y) {
    img += (size_t)blockIdx.y * w * h, out += (size_t)blockIdx.y * w * h, grid += blockIdx.y * capacity;
    dyn += blockIdx.y * SC_STRIDE, overflow_flag += blockIdx.y * SC_STRIDE;
  }
  if (!dyn_dims(dyn, &g, &color_min)) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  uint16_t v;
  if (!slice_pixel(img[i], i / w, i % w, inv_ss, inv_sc, color_min, g, grid, &v))
    atomicOr(overflow_flag, 1u);  // the reference's .unwrap() would panic
  out[i] = v;
}

}  // namespace

namespace a3d {

// The whole filter on device-resident images.  One small host round trip (min / max) sizes the grid.
// starts[t] = the first image coordinate x <= size with floor(x / sigma + 0.5) >= t (grid.rs:60-78, in the splat's own
// f64 arithmetic: this file is compiled without contraction for host and device alike), t = 0 .. cells: the pixels
// that splat into grid row / column t + 2 are [starts[t], starts[t + 1]).  Cached per (size, sigma) on the context.
a3d_status splat_starts(a3d_context* ctx, uint32_t size, double sigma_space, uint32_t cells, const uint32_t** out) {
  uint32_t key[4] = {0x53504C54u /* 'SPLT' */, size, 0, 0};
  memcpy(&key[2], &sigma_space, 8);
  for (const auto& t : ctx->tables)
    if (!memcmp(t.key, key, sizeof(key))) {
      *out = (const uint32_t*)t.d;
      return A3D_OK;
    }
  const double inv_ss = 1.0 / sigma_space;
  std::vector<uint32_t> starts(cells + 1);
  uint32_t x = 0;
  for (uint32_t t = 0; t <= cells; ++t) {
    while (x < size && f64_as_usize((double)x * inv_ss + 0.5) < t) ++x;
    starts[t] = x;
  }
  return ctx_cached_table(ctx, key, starts.data(), starts.size() * 4, (void**)out);
}

a3d_status bilateral_filter_device(a3d_context* ctx, const uint16_t* d_img, uint16_t* d_out, uint32_t w, uint32_t h,
                                   double sigma_space, double sigma_color, uint64_t out_grid_dims[3]) {
  hipStream_t s = ctx->stream;
  const uint32_t n = w * h;
  uint32_t* d_scal = nullptr;  // [0] min, [1] max, [2] overflow flag
  double2 *d_a = nullptr, *d_b = nullptr;
  a3d_status st = A3D_OK;
  auto fail = [&](const char* what) {
    set_error("bilateral filter: %s: %s", what, hipGetErrorString(hipGetLastError()));
    st = A3D_HIP_ERROR;
  };
  uint32_t h_scal[3] = {0xFFFFu, 0u, 0u};
  void* scratch = nullptr;
  // the three scalar words sit at the start of the grid scratch region; the grids follow once their size is known
  if (ctx_scratch(ctx, 1, 256, &scratch) != A3D_OK) return A3D_HIP_ERROR;
  d_scal = (uint32_t*)scratch;
  if (hipMemcpyAsync(d_scal, h_scal, 12, hipMemcpyHostToDevice, s) != hipSuccess) fail("upload");
  if (st == A3D_OK) {
    ctx->grid_clean = a3d_context::GridLayoutKey{};  // this path uses the grid scratch region its own way
    hipLaunchKernelGGL(minmax_u16_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 64)), dim3(256), 0, s, d_img, n,
                       d_scal, (uint32_t*)nullptr);
    if (hipMemcpyAsync(h_scal, d_scal, 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      fail("min/max");
  }
  GridDims g{0, 0, 0};
  if (st == A3D_OK) {
    // grid.rs:37-56 (host arithmetic: the grid size is needed to allocate)
    const uint32_t cmin = h_scal[0], cmax = h_scal[1];
    g.gh = (uint32_t)((double)(h - 1) / sigma_space) + 1 + 4;
    g.gw = (uint32_t)((double)(w - 1) / sigma_space) + 1 + 4;
    g.gd = (uint32_t)((double)(cmax - cmin) / sigma_color) + 1 + 4;
    if (out_grid_dims) out_grid_dims[0] = g.gh, out_grid_dims[1] = g.gw, out_grid_dims[2] = g.gd;
    const size_t cells = (size_t)g.gh * g.gw * g.gd;
    const size_t grid_bytes = ((cells * sizeof(double2) + 255) / 256) * 256;
    // images below 2^24 pixels: packed integer splat + all six blur passes in one LDS-tiled kernel
    const char* mode = A3D_DIAG_ENV("A3D_BILATERAL");  // diagnostics build: force the pass-per-launch path
    // (and grids whose cells 32-bit offsets can address: the fused kernels index them so)
    const bool fused = n < (1u << PACK_SHIFT) && grid_fits_idx32(g.gh, g.gw, g.gd) && !(mode && !strcmp(mode, "unfused"));
    // (growing the region synchronises; the stream is idle here anyway after the min/max read-back)
    if (ctx_scratch(ctx, 1, 256 + 2 * grid_bytes, &scratch) != A3D_OK) {
      st = A3D_HIP_ERROR;
    } else {
      if ((uint32_t*)scratch != d_scal) {  // the region moved: restore the flag word
        d_scal = (uint32_t*)scratch;
        if (hipMemsetAsync(d_scal, 0, 12, s) != hipSuccess) fail("flag reset");
      }
      d_a = (double2*)((char*)scratch + 256);
      d_b = (double2*)((char*)scratch + 256 + grid_bytes);
      // fused: d_b holds the packed u64 cells (cleared), d_a receives the blurred grid (every cell written)
      if (fused ? hipMemsetAsync(d_b, 0, cells * 8, s) != hipSuccess : hipMemsetAsync(d_a, 0, 2 * grid_bytes, s) != hipSuccess)
        fail("grid clear");
    }
    if (st == A3D_OK) {
      const double inv_ss = 1.0 / sigma_space, inv_sc = 1.0 / sigma_color;
      double2* src = d_a;
      if (fused) {
        const uint32_t *row_starts = nullptr, *col_starts = nullptr;
        if (splat_starts(ctx, h, sigma_space, g.gh - 3, &row_starts) != A3D_OK ||
            splat_starts(ctx, w, sigma_space, g.gw - 3, &col_starts) != A3D_OK)
          return A3D_HIP_ERROR;
        hipLaunchKernelGGL(splat_packed_kernel<unsigned long long>, dim3(((g.gh - 3) * (g.gw - 3) + 255) / 256), dim3(256), 0, s, d_img, w, h,
                           row_starts, col_starts, inv_sc, cmin, g, (unsigned long long*)d_b, (const uint32_t*)nullptr, 0ull,
                           (uint8_t*)nullptr, 0u, (const uint32_t*)nullptr, (uint2*)nullptr, 0u);
        hipLaunchKernelGGL(blur_fused_kernel<unsigned long long>,
                           dim3(((g.gd + BT - 1) / BT) * ((g.gw + BT - 1) / BT) * ((g.gh + BT - 1) / BT)), dim3(256), 0, s,
                           (const unsigned long long*)d_b, g, (double*)d_a, (uint32_t*)nullptr, 0ull,
                           (const uint32_t*)nullptr, 0u, (const uint32_t*)nullptr, 0u);
      } else {
        hipLaunchKernelGGL(splat_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_img, w, h, inv_ss, inv_sc, cmin, g,
                           (double*)d_a);
        const uint64_t interior = (uint64_t)(g.gh - 2) * (g.gw - 2) * (g.gd - 1);
        const uint32_t blocks = (uint32_t)((interior + 255) / 256);
        double2* dst = d_b;
        for (int axis = 0; axis < 3; ++axis)
          for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(blur_axis_kernel, dim3(blocks), dim3(256), 0, s, src, dst, g, axis);
            std::swap(src, dst);
          }
        // six passes: the result is back in d_a (== src after the final swap); normalised into the other buffer
        hipLaunchKernelGGL(normalize_kernel, dim3((uint32_t)((cells + 255) / 256)), dim3(256), 0, s, (const double2*)src,
                           (double*)dst, (unsigned long long)cells);
        src = dst;
      }
      hipLaunchKernelGGL(slice_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_img, w, h, inv_ss, inv_sc, cmin, g,
                         (const double*)src, d_out, d_scal + 2, (const uint32_t*)nullptr, 0ull);
      if (hipGetLastError() != hipSuccess) fail("kernel launch");
    }
    if (st == A3D_OK && (hipMemcpyAsync(h_scal + 2, d_scal + 2, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                         hipStreamSynchronize(s) != hipSuccess))
      fail("flag download");
  }
  if (st != A3D_OK) return st;
  if (h_scal[2]) {
    set_error("bilateral slice produced a value outside u16 (the reference panics in num::cast().unwrap())");
    return A3D_CAST_OVERFLOW;
  }
  return A3D_OK;
}

unsigned long long bilateral_grid_cells(uint32_t w, uint32_t h, double sigma_space, double sigma_color, uint32_t depth_span) {
  const unsigned long long gh = (uint32_t)((double)(h - 1) / sigma_space) + 1 + 4, gw = (uint32_t)((double)(w - 1) / sigma_space) + 1 + 4;
  const unsigned long long gd = (uint32_t)((double)depth_span / sigma_color) + 1 + 4;
  return gh * gw * gd;
}

// The grids of a batch of frames enqueued WITHOUT a host round trip: min/max, the grid dimensions and the capacity
// check stay on the device, the kernels read them from each frame's scalar block, launch sizes come from the capacity.
// Five launches: minmax (per-block partials) -> dims_table -> splat -> blur (compacts its own tile lists) -> unsplat;
// nothing is cleared per enqueue (a3d_context::grid_clean).
template <typename CELL>
static void enqueue_splat_blur(hipStream_t s, const uint16_t* d_depth, uint32_t n_frames, uint32_t w, uint32_t h,
                               const uint32_t* row_starts, const uint32_t* col_starts, double inv_sc, dim3 splat_grid,
                               dim3 blur_grid, dim3 list_grid, GridBatch* out, unsigned long long capacity, uint8_t* flags,
                               uint32_t flags_stride, uint32_t* lists, bool lists_in_blur, const uint32_t* channel_of,
                               uint2* extent, uint32_t columns, bool defer_unsplat) {
  const GridDims none{0, 0, 0};
  hipLaunchKernelGGL(splat_packed_kernel<CELL>, splat_grid, dim3(256), 0, s, d_depth, w, h, row_starts, col_starts, inv_sc, 0u,
                     none, (CELL*)out->packed, (const uint32_t*)out->scal, capacity, flags, flags_stride, channel_of, extent, columns);
  if (!lists_in_blur)
    hipLaunchKernelGGL(tile_list_kernel, list_grid, dim3(256), 0, s, (const uint8_t*)flags, flags_stride, out->scal, lists);
  if (lists_in_blur)  // 1-D: the kernel places the frames XCD by XCD (blur_grid.x blocks per frame)
    hipLaunchKernelGGL(blur_fused_kernel<CELL>, dim3(blur_grid.x * blur_grid.y), dim3(256), 0, s, (const CELL*)out->packed, none,
                       out->blurred, out->scal, capacity, (const uint32_t*)nullptr, flags_stride, (const uint32_t*)flags, blur_grid.x);
  else
    hipLaunchKernelGGL(blur_fused_kernel<CELL>, blur_grid, dim3(256), 0, s, (const CELL*)out->packed, none, out->blurred, out->scal,
                       capacity, (const uint32_t*)lists, flags_stride, (const uint32_t*)nullptr, 0u);
  if (!defer_unsplat)
    hipLaunchKernelGGL(unsplat_kernel<CELL>, splat_grid, dim3(256), 0, s, (CELL*)out->packed, (const uint32_t*)out->scal, capacity,
                       (const uint2*)extent, columns, (uint32_t*)flags, flags_stride);
  (void)n_frames;
}

a3d_status bilateral_grids_enqueue(a3d_context* ctx, const uint16_t* d_depth, uint32_t n_frames, uint32_t w, uint32_t h,
                                   double sigma_space, double sigma_color, unsigned long long capacity, GridBatch* out,
                                   bool defer_unsplat) {
  const uint32_t n = w * h;
  A3D_REQUIRE(n < (1u << PACK_SHIFT), A3D_INVALID_PARAMETER,
              "the device frame builder's bilateral filter handles images below 2^24 pixels");
  hipStream_t s = ctx->stream;
  // The region is laid out for the most frames any enqueue on this context has asked for, so that a short last chunk of
  // a batch finds its arrays where the full chunks left them (and left them zeroed).
  ctx->grid_layout_frames = std::max(ctx->grid_layout_frames, n_frames);
  const uint32_t lf = ctx->grid_layout_frames;
  // any grid of `capacity` cells with this image's row / column extents has at most this many 12^3 tiles
  const uint32_t gh = (uint32_t)((double)(h - 1) / sigma_space) + 1 + 4, gw = (uint32_t)((double)(w - 1) / sigma_space) + 1 + 4;
  const unsigned long long plane_tiles = (unsigned long long)((gh + BT - 1) / BT) * ((gw + BT - 1) / BT);
  const unsigned long long max_gd = capacity / ((unsigned long long)gh * gw) + 1;
  const uint32_t tiles = (uint32_t)std::min<unsigned long long>(plane_tiles * ((max_gd + BT - 1) / BT), 1u << 30);
  const uint32_t flags_stride = ((tiles + 255) / 256) * 256;
  const uint32_t columns = (gh - 3) * (gw - 3);  // the (row, column)s of the grid pixels can splat into
  constexpr uint32_t PARTIALS = 64;              // min / max blocks per frame
  // [scalars][min / max partials][tile flags][tile lists][colour -> channel tables][splat extents][packed][blurred]
  auto pad = [](size_t b) { return ((b + 255) / 256) * 256; };
  const size_t scal_only = pad((size_t)lf * SC_STRIDE * 4);
  const size_t partial_bytes = pad((size_t)lf * PARTIALS * 2 * 4);
  const size_t flag_bytes = (size_t)lf * flags_stride;
  const bool lists_in_blur = flags_stride <= BLUR_LIST_MAX;
  const size_t list_bytes = lists_in_blur ? 0 : (size_t)lf * 2 * flags_stride * 4;  // the two tile lists
  const size_t table_bytes = (size_t)lf * 65536 * 4;
  const size_t extent_bytes = pad((size_t)lf * columns * sizeof(uint2));
  const size_t head_bytes = scal_only + partial_bytes + flag_bytes + list_bytes + table_bytes + extent_bytes;
  capacity = (capacity + 3) & ~3ull;  // a multiple of four cells: every frame's packed grid starts 16-byte aligned
  // at most floor(sigma) + 1 image rows (columns) round to one grid row (column): 4-byte cells while a (row, column)
  // can not receive more than 255 pixels (A3D_BILATERAL_CELLS=wide: always 8-byte cells, a cross-check)
  const uint32_t reach = (uint32_t)std::min(sigma_space, 1e6) + 2;
  const char* cells_mode = A3D_DIAG_ENV("A3D_BILATERAL_CELLS");
  const bool narrow = reach * reach <= 255 && !(cells_mode && !strcmp(cells_mode, "wide"));
  const uint32_t cell_bytes = narrow ? 4 : 8;
  const size_t packed_bytes = pad((size_t)lf * capacity * cell_bytes);
  void* region = nullptr;
  A3D_TRY(ctx_scratch(ctx, 1, head_bytes + packed_bytes + (size_t)lf * capacity * 8 + 256, &region));
  out->scal = (uint32_t*)region;
  uint32_t* partials = (uint32_t*)((char*)region + scal_only);
  uint8_t* flags = (uint8_t*)partials + partial_bytes;
  uint32_t* lists = (uint32_t*)(flags + flag_bytes);
  uint32_t* channel_of = (uint32_t*)((char*)lists + list_bytes);
  uint2* extent = (uint2*)((char*)channel_of + table_bytes);
  out->packed = (char*)region + head_bytes;
  out->blurred = (double*)((char*)out->packed + packed_bytes);
  out->capacity = capacity;
  const uint32_t *row_starts = nullptr, *col_starts = nullptr;
  A3D_TRY(splat_starts(ctx, h, sigma_space, gh - 3, &row_starts));  // (before anything is enqueued: a first use synchronises)
  A3D_TRY(splat_starts(ctx, w, sigma_space, gw - 3, &col_starts));
  // the zero invariant of flags and packed cells: established once per layout, kept by unsplat_kernel
  a3d_context::GridLayoutKey key;
  key.region = region, key.capacity = capacity, key.cell_bytes = cell_bytes, key.flags_stride = flags_stride, key.frames = lf,
  key.columns = columns;
  const a3d_context::GridLayoutKey& have = ctx->grid_clean;
  if (!(have.region == key.region && have.capacity == key.capacity && have.cell_bytes == key.cell_bytes &&
        have.flags_stride == key.flags_stride && have.frames == key.frames && have.columns == key.columns)) {
    A3D_HIP_TRY(hipMemsetAsync(out->scal, 0, scal_only, s));  // (the tickets of minmax_dims_kernel)
    A3D_HIP_TRY(hipMemsetAsync(flags, 0, flag_bytes, s));
    A3D_HIP_TRY(hipMemsetAsync(out->packed, 0, packed_bytes, s));
  }
  ctx->grid_clean = a3d_context::GridLayoutKey{};  // unknown until the whole sequence is enqueued
  // Diagnostics build, A3D_BILATERAL_POISON=1: every blurred cell starts as NaN, so a slice that read a cell no marked tile
  // wrote (the tile marking's margin argument, splat_packed_kernel) fails the u16 cast check instead of passing on the stale
  // value of an earlier, similar frame (tests/test_gpu_frame_prep.py).
  if (A3D_DIAG_ENV("A3D_BILATERAL_POISON")) A3D_HIP_TRY(hipMemsetAsync(out->blurred, 0xFF, (size_t)lf * capacity * 8, s));
  const uint32_t mm_blocks = std::min<uint32_t>((n + 255) / 256, PARTIALS);
  const double inv_sc = 1.0 / sigma_color;
  // (diagnostics build, A3D_BILATERAL_MINMAX=fused: both in one launch, the frame's last block sizing the grid — measured
  // round 6: 12.5 us against 6.5 + 4.8 us for the two launches per 32 frames, no gain; kept as a cross-check)
  const char* mm_mode = A3D_DIAG_ENV("A3D_BILATERAL_MINMAX");
  if (mm_mode && !strcmp(mm_mode, "fused")) {
#ifdef A3D_DIAGNOSTICS
    hipLaunchKernelGGL(minmax_dims_kernel, dim3(mm_blocks, n_frames), dim3(256), 0, s, d_depth, n, out->scal, w, h,
                       sigma_space, sigma_color, capacity, channel_of, inv_sc);
#endif
  } else {
    hipLaunchKernelGGL(minmax_u16_kernel, dim3(mm_blocks, n_frames), dim3(256), 0, s, d_depth, n, (uint32_t*)nullptr, partials);
    hipLaunchKernelGGL(dims_table_kernel, dim3(16, n_frames), dim3(256), 0, s, (const uint32_t*)partials, mm_blocks, out->scal, w, h,
                       sigma_space, sigma_color, capacity, channel_of, inv_sc);
  }
  // (gh and gw depend on the image size only: the launch covers every (row, column) pixels can splat into)
  const dim3 splat_grid((columns + 255) / 256, n_frames);
  // one resident round of blur blocks, shared out over the frames: four blocks per CU with the narrow cells (39 KiB of
  // LDS and 110 VGPRs each), two with the wide ones (70 KiB, 224 VGPRs)
  static const int blur_per_cu[2] = {[] {
                                       int n = 0;
                                       return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blur_fused_kernel<unsigned long long>, 256, 0) == hipSuccess && n > 0 ? n : 2;
                                     }(),
                                     [] {
                                       int n = 0;
                                       return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blur_fused_kernel<uint32_t>, 256, 0) == hipSuccess && n > 0 ? n : 2;
                                     }()};
  const uint32_t resident = (uint32_t)blur_per_cu[narrow ? 1 : 0] * (uint32_t)std::max(1, ctx->num_cus);
  const uint32_t per_frame = std::max(8u, std::min(std::max(1u, tiles), (resident + n_frames - 1) / n_frames));
  const dim3 list_grid((std::max(1u, tiles) + 255) / 256, n_frames), blur_grid(per_frame, n_frames);
  if (narrow)
    enqueue_splat_blur<uint32_t>(s, d_depth, n_frames, w, h, row_starts, col_starts, inv_sc, splat_grid, blur_grid, list_grid, out,
                                 capacity, flags, flags_stride, lists, lists_in_blur, channel_of, extent, columns, defer_unsplat);
  else
    enqueue_splat_blur<unsigned long long>(s, d_depth, n_frames, w, h, row_starts, col_starts, inv_sc, splat_grid, blur_grid,
                                           list_grid, out, capacity, flags, flags_stride, lists, lists_in_blur, channel_of, extent,
                                           columns, defer_unsplat);
  A3D_HIP_TRY(hipGetLastError());
  out->clean = key;
  if (defer_unsplat) {  // (the caller's next kernel puts the zeros back, then commits out->clean)
    out->unsplat.packed = out->packed, out->unsplat.extent = extent, out->unsplat.flags = (uint32_t*)flags;
    out->unsplat.capacity = capacity, out->unsplat.columns = columns, out->unsplat.flag_words = flags_stride / 4;
    out->unsplat.cell_bytes = cell_bytes;
  } else {
    ctx->grid_clean = key;
  }
  return A3D_OK;
}

}  // namespace a3d

extern "C" a3d_status a3d_bilateral_filter_u16(a3d_context* ctx, const uint16_t* image, uint64_t width,
                                               uint64_t height, double sigma_space, double sigma_color,
                                               uint16_t* out_image, uint64_t out_grid_dims[3]) {
  A3D_REQUIRE(ctx && image && out_image, A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(width > 0 && height > 0 && width * height < (1ull << 28), A3D_INVALID_PARAMETER, "bad image size");
  A3D_REQUIRE(sigma_space > 0.0 && sigma_color > 0.0, A3D_INVALID_PARAMETER, "sigmas must be positive");
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint32_t w = (uint32_t)width, h = (uint32_t)height, n = w * h;
  // image and result staged in the context's grow-only scratch region (no hipMalloc / hipFree per call)
  const size_t stride = (((size_t)n * 2 + 255) / 256) * 256;
  void* region = nullptr;
  A3D_TRY(ctx_scratch(ctx, 0, 2 * stride, &region));
  uint16_t *d_img = (uint16_t*)region, *d_out = (uint16_t*)((char*)region + stride);
  a3d_status st = A3D_OK;
  if (hipMemcpyAsync(d_img, image, (size_t)n * 2, hipMemcpyHostToDevice, s) != hipSuccess) {
    set_error("a3d_bilateral_filter_u16: upload: %s", hipGetErrorString(hipGetLastError()));
    st = A3D_HIP_ERROR;
  }
  const char* mode = A3D_DIAG_ENV("A3D_BILATERAL");  // diagnostics build: force the pass-per-launch path
  const bool one_pass = n < (1u << PACK_SHIFT) && !(mode && (!strcmp(mode, "unfused") || !strcmp(mode, "sync")));
  if (st == A3D_OK && !one_pass) {  // the pass-per-launch path (huge images, cross-check): min / max via the host
    st = bilateral_filter_device(ctx, d_img, d_out, w, h, sigma_space, sigma_color, out_grid_dims);
    if (st == A3D_OK && (hipMemcpyAsync(out_image, d_out, n * 2, hipMemcpyDeviceToHost, s) != hipSuccess ||
                         hipStreamSynchronize(s) != hipSuccess)) {
      set_error("a3d_bilateral_filter_u16: download: %s", hipGetErrorString(hipGetLastError()));
      st = A3D_HIP_ERROR;
    }
    return st;
  }
  // One enqueue, no host round trip for min / max: the frame builder's grid path with one frame, then the slice; a
  // grid that outgrew the scratch region is run again with more room.
  for (int attempt = 0; st == A3D_OK && attempt < 3; ++attempt) {
    if (ctx->grid_capacity == 0) ctx->grid_capacity = bilateral_grid_cells(w, h, sigma_space, sigma_color, 4096);
    GridBatch gb;
    st = bilateral_grids_enqueue(ctx, d_img, 1, w, h, sigma_space, sigma_color, ctx->grid_capacity, &gb);
    if (st != A3D_OK) break;
    const GridDims none{0, 0, 0};
    hipLaunchKernelGGL(slice_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_img, w, h, 1.0 / sigma_space, 1.0 / sigma_color,
                       0u, none, (const double*)gb.blurred, d_out, gb.scal + SC_OVERFLOW, (const uint32_t*)gb.scal, gb.capacity);
    uint32_t* r = ctx->pinned_words;
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(r, gb.scal, SC_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(out_image, d_out, (size_t)n * 2, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      set_error("a3d_bilateral_filter_u16: %s", hipGetErrorString(hipGetLastError()));
      return A3D_HIP_ERROR;
    }
    if (r[SC_TOO_BIG]) {
      const unsigned long long need = (unsigned long long)r[SC_GH] * r[SC_GW] * r[SC_GD];
      ctx->grid_capacity = need + need / 4;
      A3D_REQUIRE(attempt < 2, A3D_HIP_ERROR, "a3d_bilateral_filter_u16: the grid kept outgrowing its scratch region");
      continue;
    }
    if (out_grid_dims) out_grid_dims[0] = r[SC_GH], out_grid_dims[1] = r[SC_GW], out_grid_dims[2] = r[SC_GD];
    // A3D_CAST_OVERFLOW mirrors `num::cast::cast(trilinear).unwrap()` (src/bilateral/grid.rs:129): None iff the value is
    // NaN, <= -1 or >= 65536.  With the sigmas this entry point accepts (positive, finite or +inf) it cannot happen:
    // every normalised cell is 0 or a weighted mean of u16 inputs (grid.rs:97-102), so it lies in [0, 65535]; the slice
    // position of a pixel is inside the padded grid (two cells of padding on every side, edge_aware_filter.rs:64-74), so
    // the three interpolation weights lie in [0, 1) and the trilinear value is a convex combination of such cells, off by
    // rounding of a few ulp of 65535 (1e-11) — nowhere near 65536 or -1.  The flag is kept (the kernel computes it for
    // free) as a guard on that argument; tests/test_gpu_frame_prep.py pins the extreme input (0 / 65535 checkerboards).
    if (r[SC_OVERFLOW]) {
      set_error("bilateral slice produced a value outside u16 (the reference panics in num::cast().unwrap())");
      return A3D_CAST_OVERFLOW;
    }
    break;
  }
  return st;
}

// The same filter on images that are ALREADY in device memory (the shape of benches/bench_bilateral.rs
// without PCIe in it): `n_images` images [n][height][width] u16 in HBM in, the same layout out.  Launch sequences of up to
// 32 images (min / max + grid sizing, splat, blur, slice, zeros back: six launches whatever the count); per sequence 64
// bytes of scalars per image come back to the host (did the grid fit its scratch region? — the sequence runs again with
// more room if not — and the cast check).
extern "C" a3d_status a3d_bilateral_filter_u16_device(a3d_context* ctx, const uint16_t* d_images, uint64_t n_images,
                                                      uint64_t width, uint64_t height, double sigma_space,
                                                      double sigma_color, uint16_t* d_out) {
  A3D_REQUIRE(ctx && d_images && d_out, A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(n_images >= 1 && n_images <= (1u << 20), A3D_INVALID_PARAMETER, "bad image count");
  A3D_REQUIRE(width > 0 && height > 0 && width * height < (1ull << PACK_SHIFT), A3D_INVALID_PARAMETER,
              "a3d_bilateral_filter_u16_device handles images below 2^24 pixels");
  A3D_REQUIRE(sigma_space > 0.0 && sigma_color > 0.0 && std::isfinite(sigma_space) && std::isfinite(sigma_color),
              A3D_INVALID_PARAMETER, "sigmas must be positive and finite");
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint32_t w = (uint32_t)width, h = (uint32_t)height, n = w * h;
  ctx->last_build_kernel_ms = 0.f;
  for (uint64_t& v : ctx->build_stats) v = 0;  // (a3d_context_last_build_stats: images, grid cells, marked / zero-written tiles)
  constexpr uint64_t MAX_SEQ = 32;
  for (uint64_t f0 = 0; f0 < n_images;) {
    if (ctx->grid_capacity == 0) ctx->grid_capacity = bilateral_grid_cells(w, h, sigma_space, sigma_color, 4096);
    // (a sequence's grids — 12 B per cell of capacity, narrow cells — stay below 4 GB of scratch)
    const uint64_t fit = std::max<uint64_t>(1, (4ull << 30) / (ctx->grid_capacity * 16 + 1));
    const uint32_t F = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(MAX_SEQ, fit), n_images - f0);
    bool done = false;
    for (int attempt = 0; attempt < 3 && !done; ++attempt) {
      GridBatch gb;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (ctx->build_profiling) {
        while (ctx->build_events.size() < 2) {
          hipEvent_t e;
          A3D_HIP_TRY(hipEventCreate(&e));
          ctx->build_events.push_back(e);
        }
        e0 = ctx->build_events[0], e1 = ctx->build_events[1];
        (void)hipEventRecord(e0, s);
      }
      A3D_TRY(bilateral_grids_enqueue(ctx, d_images + f0 * n, F, w, h, sigma_space, sigma_color, ctx->grid_capacity, &gb));
      const GridDims none{0, 0, 0};
      hipLaunchKernelGGL(slice_kernel, dim3((n + 255) / 256, F), dim3(256), 0, s, d_images + f0 * n, w, h, 1.0 / sigma_space,
                         1.0 / sigma_color, 0u, none, (const double*)gb.blurred, d_out + f0 * n, gb.scal + SC_OVERFLOW,
                         (const uint32_t*)gb.scal, gb.capacity);
      if (e1) (void)hipEventRecord(e1, s);
      uint32_t* r = ctx->pinned_words;
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(r, gb.scal, (size_t)F * SC_STRIDE * sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) {
        set_error("a3d_bilateral_filter_u16_device: %s", hipGetErrorString(hipGetLastError()));
        return A3D_HIP_ERROR;
      }
      unsigned long long need = 0;
      bool overflow = false;
      for (uint32_t f = 0; f < F; ++f) {
        const uint32_t* q = r + f * SC_STRIDE;
        if (q[SC_TOO_BIG]) need = std::max<unsigned long long>(need, (unsigned long long)q[SC_GH] * q[SC_GW] * q[SC_GD]);
        else overflow |= q[SC_OVERFLOW] != 0;
      }
      if (overflow) {  // (cannot happen for these sigmas: see a3d_bilateral_filter_u16)
        set_error("bilateral slice produced a value outside u16 (the reference panics in num::cast().unwrap())");
        return A3D_CAST_OVERFLOW;
      }
      if (need) {
        A3D_REQUIRE(need < (1ull << 29), A3D_INVALID_PARAMETER,
                    "an image's bilateral grid would exceed 2^29 cells (raise sigma_color or sigma_space)");
        A3D_REQUIRE(attempt < 2, A3D_HIP_ERROR, "a3d_bilateral_filter_u16_device: the grid kept outgrowing its scratch region");
        ctx->grid_capacity = need + need / 4;
        continue;  // (the sequence again, with room; F is recomputed from the new capacity by the outer loop's next trip)
      }
      float ms = 0.f;
      if (e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ctx->last_build_kernel_ms += ms;
      ctx->build_stats[0] += F;
      for (uint32_t f = 0; f < F; ++f) {
        const uint32_t* q = r + f * SC_STRIDE;
        ctx->build_stats[1] += (uint64_t)q[SC_GH] * q[SC_GW] * q[SC_GD];
        ctx->build_stats[2] += q[SC_NLIST], ctx->build_stats[3] += q[SC_NZERO];
      }
      done = true;
    }
    if (done) f0 += F;
  }
  return A3D_OK;
}

#if defined(A3D_DIAGNOSTICS) && defined(A3D_BLUR_STAMPS)
// scripts/blur_stamps.py: the stamps of the most recent blur launch's block 0 (and the tile counter back to zero)
extern "C" int a3d_debug_blur_stamps(unsigned long long out[80]) {
  const int zero = 0;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blur_stamps), 80 * sizeof(unsigned long long)) == hipSuccess &&
                 hipMemcpyToSymbol(HIP_SYMBOL(g_blur_tile), &zero, sizeof(int)) == hipSuccess
             ? 0
             : 1;
}
#endif
