// Device-side pieces of BilateralGrid shared by bilateral.hip (the filter) and frame.hip (the frame builder, which
// slices the blurred grid inside its back-projection kernel): src/bilateral/grid.rs:90-162.
#pragma once
#include "common.hpp"

namespace a3d {

struct GridDims {
  uint32_t gh, gw, gd;
};

// `x as usize` for a non-negative finite f64
__host__ __device__ __forceinline__ uint32_t f64_as_usize(double x) { return x > 0.0 ? (uint32_t)x : 0u; }

// Device-side scalars of one filter call.  When a kernel gets a non-null `dyn` pointer, the grid dimensions and the
// colour minimum come from there instead of from its arguments: the host then enqueues the whole filter without
// the min/max round trip.  A batch of frames has one block of SC_STRIDE words per frame.
enum { SC_MIN = 0, SC_MAX = 1, SC_OVERFLOW = 2, SC_TICKET = 3 /* minmax_dims_kernel's block counter */, SC_GH = 4, SC_GW = 5, SC_GD = 6, SC_TOO_BIG = 7, SC_WORDS = 8,
       SC_NLIST = 8, SC_NZERO = 9,    // (device only) lengths of the frame's blur tile lists
       SC_ACC_NMIN = 10, SC_ACC_MAX = 11 };  // (device only) minmax_dims_kernel's running maxima of 0xFFFF - min and of max
constexpr uint32_t SC_STRIDE = 16;  // words between the scalar blocks of consecutive frames (64 B: one per line)

__device__ __forceinline__ bool dyn_dims(const uint32_t* __restrict__ dyn, GridDims* g, uint32_t* color_min) {
  if (!dyn) return true;
  if (dyn[SC_TOO_BIG]) return false;  // the grid does not fit the scratch region: the host grows it and repeats
  *g = GridDims{dyn[SC_GH], dyn[SC_GW], dyn[SC_GD]};
  if (color_min) *color_min = dyn[SC_MIN];
  return true;
}

__device__ __forceinline__ uint32_t clampu(uint32_t v, uint32_t hi) { return v > hi ? hi : v; }

// BilateralGrid::normalize (grid.rs:90-104): where weight > 0, value /= weight.  Applied once per cell by whoever
// writes the blurred grid, which therefore holds ONE f64 per cell (the weight channel is never read again): the slice
// gathers 8 B per cell and divides nothing, instead of 16 B and eight f64 divisions per pixel.
__device__ __forceinline__ double normalized_cell(double value, double weight) { return weight > 0.0 ? value / weight : value; }
// trilinear (grid.rs:132-162) reads the value channel
__device__ __forceinline__ double cell_value(const double* __restrict__ grid, GridDims g, uint32_t r, uint32_t c, uint32_t z) {
  return grid[((size_t)r * g.gw + c) * g.gd + z];
}
// Grids of the frame builder and of the one-enqueue filter are below 2^29 cells with gh x gw and gd below 2^24
// (grid_fits_idx32; the device-side dims_table_kernel refuses anything else): a cell's byte offset is then 32-bit arithmetic on
// the full-rate 24-bit multiplier off a uniform base pointer, instead of a 64-bit multiply-add (v_mad_u64_u32, quarter
// rate) and a 64-bit shift-add per gathered cell — eight of each per pixel in a kernel that is VALU-issue bound.
__host__ __device__ __forceinline__ bool grid_fits_idx32(unsigned long long gh, unsigned long long gw, unsigned long long gd) {
  return gh * gw < (1ull << 24) && gd < (1ull << 24) && gh * gw * gd < (1ull << 29);
}
typedef const char __attribute__((address_space(1)))* a3d_gptr_c;
typedef char __attribute__((address_space(1)))* a3d_gptr;
__device__ __forceinline__ double cell_value32(const double* __restrict__ grid, uint32_t row_col_base, uint32_t z) {
  return *(const double __attribute__((address_space(1)))*)((a3d_gptr_c)grid + (row_col_base + z) * 8u);
}

// One axis of the trilinear slice (grid.rs:132-146): the cell below, the cell above (both clamped) and the fraction.
// The row and column parts depend on the pixel's row / column only: the frame builder computes them once per tile row and
// tile column (level0_kernel) instead of once per pixel — the same f64 operations on the same operands.
struct SliceAxis {
  uint32_t lo, hi;
  double frac;
};
__device__ __forceinline__ SliceAxis slice_axis(double coord, uint32_t cells) {
  SliceAxis a;
  a.lo = clampu(f64_as_usize(coord), cells - 1), a.hi = clampu(f64_as_usize(coord + 1.0), cells - 1);
  a.frac = coord - (double)a.lo;
  return a;
}
__device__ __forceinline__ SliceAxis slice_axis_spatial(uint32_t pixel, double inv_ss, uint32_t cells) {
  return slice_axis((double)pixel * inv_ss + 2.0, cells);
}

// BilateralGrid::slice for one pixel (grid.rs:106-130, trilinear :132-162) from its row and column parts: every pixel,
// zeros included.  Returns false when the value is not representable as u16 (num::cast::<f64,u16>().unwrap() would panic).
template <bool IDX32 = false>
__device__ __forceinline__ bool slice_pixel_axes(uint32_t color, const SliceAxis& ry, const SliceAxis& cx, double inv_sc,
                                                 uint32_t color_min, GridDims g, const double* __restrict__ grid,
                                                 uint16_t* out) {
  const SliceAxis cz = slice_axis((double)(color - color_min) * inv_sc + 2.0, g.gd);
  const uint32_t z = cz.lo, zz = cz.hi, y = ry.lo, yy = ry.hi, x = cx.lo, xx = cx.hi;
  const double za = cz.frac, ya = ry.frac, xa = cx.frac;
  double v000, v010, v100, v110, v001, v011, v101, v111;  // v[row][col][channel] of the cell's eight corners
  if (IDX32) {
    const uint32_t ry0 = __umul24(y, g.gw), ry1 = __umul24(yy, g.gw);
    const uint32_t b00 = __umul24(ry0 + x, g.gd), b01 = __umul24(ry0 + xx, g.gd), b10 = __umul24(ry1 + x, g.gd),
                   b11 = __umul24(ry1 + xx, g.gd);
    v000 = cell_value32(grid, b00, z), v010 = cell_value32(grid, b01, z), v100 = cell_value32(grid, b10, z);
    v110 = cell_value32(grid, b11, z), v001 = cell_value32(grid, b00, zz), v011 = cell_value32(grid, b01, zz);
    v101 = cell_value32(grid, b10, zz), v111 = cell_value32(grid, b11, zz);
  } else {
    v000 = cell_value(grid, g, y, x, z), v010 = cell_value(grid, g, y, xx, z), v100 = cell_value(grid, g, yy, x, z);
    v110 = cell_value(grid, g, yy, xx, z), v001 = cell_value(grid, g, y, x, zz), v011 = cell_value(grid, g, y, xx, zz);
    v101 = cell_value(grid, g, yy, x, zz), v111 = cell_value(grid, g, yy, xx, zz);
  }
  // the 8-term sum in the order written at grid.rs:152-159
  const double value = (1.0 - ya) * (1.0 - xa) * (1.0 - za) * v000 +
                       (1.0 - ya) * xa * (1.0 - za) * v010 +
                       ya * (1.0 - xa) * (1.0 - za) * v100 +
                       ya * xa * (1.0 - za) * v110 +
                       (1.0 - ya) * (1.0 - xa) * za * v001 +
                       (1.0 - ya) * xa * za * v011 +
                       ya * (1.0 - xa) * za * v101 +
                       ya * xa * za * v111;
  if (value > -1.0 && value < 65536.0) {
    *out = (uint16_t)value;  // truncation toward zero
    return true;
  }
  *out = 0;
  return false;
}
// The same slice in two steps, so that a thread can have the gathers of SEVERAL pixels in flight before it combines the
// first (level0_quad_kernel: the blurred grids of a launch sequence are far larger than the L2, every gather is a trip to
// the Infinity Cache or to HBM): slice_gather issues the eight loads (32-bit cell offsets: grid_fits_idx32), slice_combine
// is the 8-term sum of grid.rs:152-159 in the written order and the cast.
struct SliceTaps {
  double v[8];  // v000, v010, v100, v110, v001, v011, v101, v111 (row, column, channel)
  double za;
};
__device__ __forceinline__ void slice_gather(uint32_t color, const SliceAxis& ry, const SliceAxis& cx, double inv_sc,
                                             uint32_t color_min, GridDims g, const double* __restrict__ grid, SliceTaps* t) {
  const SliceAxis cz = slice_axis((double)(color - color_min) * inv_sc + 2.0, g.gd);
  const uint32_t ry0 = __umul24(ry.lo, g.gw), ry1 = __umul24(ry.hi, g.gw);
  const uint32_t b00 = __umul24(ry0 + cx.lo, g.gd), b01 = __umul24(ry0 + cx.hi, g.gd), b10 = __umul24(ry1 + cx.lo, g.gd),
                 b11 = __umul24(ry1 + cx.hi, g.gd);
  t->v[0] = cell_value32(grid, b00, cz.lo), t->v[1] = cell_value32(grid, b01, cz.lo), t->v[2] = cell_value32(grid, b10, cz.lo);
  t->v[3] = cell_value32(grid, b11, cz.lo), t->v[4] = cell_value32(grid, b00, cz.hi), t->v[5] = cell_value32(grid, b01, cz.hi);
  t->v[6] = cell_value32(grid, b10, cz.hi), t->v[7] = cell_value32(grid, b11, cz.hi);
  t->za = cz.frac;
}
__device__ __forceinline__ bool slice_combine(const SliceTaps& t, double ya, double xa, uint16_t* out) {
  const double za = t.za;
  const double value = (1.0 - ya) * (1.0 - xa) * (1.0 - za) * t.v[0] +
                       (1.0 - ya) * xa * (1.0 - za) * t.v[1] +
                       ya * (1.0 - xa) * (1.0 - za) * t.v[2] +
                       ya * xa * (1.0 - za) * t.v[3] +
                       (1.0 - ya) * (1.0 - xa) * za * t.v[4] +
                       (1.0 - ya) * xa * za * t.v[5] +
                       ya * (1.0 - xa) * za * t.v[6] +
                       ya * xa * za * t.v[7];
  if (value > -1.0 && value < 65536.0) {
    *out = (uint16_t)value;  // truncation toward zero
    return true;
  }
  *out = 0;
  return false;
}
template <bool IDX32 = false>
__device__ __forceinline__ bool slice_pixel(uint32_t color, uint32_t r, uint32_t c, double inv_ss, double inv_sc,
                                            uint32_t color_min, GridDims g, const double* __restrict__ grid,
                                            uint16_t* out) {
  return slice_pixel_axes<IDX32>(color, slice_axis_spatial(r, inv_ss, g.gh), slice_axis_spatial(c, inv_ss, g.gw), inv_sc,
                                 color_min, g, grid, out);
}

// What puts the zeros back (a3d_context::grid_clean) when the kernel behind the filter does it instead of unsplat_kernel:
// the splat's per-(row, column) channel extents, the packed cells and the tile flags of the batch.
struct Unsplat {
  void* packed = nullptr;          // null: nothing deferred
  const uint2* extent = nullptr;   // [frames][columns]: lowest, highest channel the splat wrote (lowest > highest: none)
  uint32_t* flags = nullptr;       // [frames][flag_words]
  unsigned long long capacity = 0;
  uint32_t columns = 0, flag_words = 0, cell_bytes = 0;
};
// One block's share of a frame's unsplat: the grid (row, column)s whose FIRST image row / column lie in the block's patch
// [r0, r1) x [c0, c1) of the image — t(x) = floor(x / sigma + 0.5) is the splat's own expression (grid.rs:60-78), so grid row
// t + 2 receives the image rows {x : t(x) = t} and its first one is in the patch iff t(r0 - 1) < t <= t(r1 - 1) — and an equal
// share of the frame's flag words.  Thread = one (row, column); `block`, `blocks`: this block's rank among the frame's.
// (256 threads per block: written out, `blockDim.x` is a load from the dispatch packet and its wait a wait for every store before it)
__device__ __forceinline__ void unsplat_columns(const Unsplat& U, uint32_t frame, GridDims g, double inv_ss, int r0, int r1, int c0,
                                                int c1, uint32_t block, uint32_t blocks) {
  constexpr uint32_t THREADS = 256;
  auto t_of = [&](int x) { return f64_as_usize((double)x * inv_ss + 0.5); };
  const uint32_t tr_lo = r0 == 0 ? 0u : t_of(r0 - 1) + 1u, tr_hi = t_of(r1 - 1) + 1u;
  const uint32_t tc_lo = c0 == 0 ? 0u : t_of(c0 - 1) + 1u, tc_hi = t_of(c1 - 1) + 1u;
  const uint32_t nr = tr_hi - tr_lo, nc = tc_hi - tc_lo, cols = g.gw - 3;
  char* cells = (char*)U.packed + (size_t)frame * U.capacity * U.cell_bytes;
  for (uint32_t e = threadIdx.x; e < nr * nc; e += THREADS) {
    const uint32_t t_row = tr_lo + e / nc, t_col = tc_lo + e % nc;
    if (t_row >= g.gh - 3 || t_col >= cols) continue;
    const uint2 x = U.extent[(size_t)frame * U.columns + (t_row * cols + t_col)];
    const uint32_t column = __umul24(__umul24(t_row + 2, g.gw) + t_col + 2, g.gd);
    for (uint32_t ch = x.x; ch <= x.y; ++ch) {  // (an untouched column has x.x > x.y)
      if (U.cell_bytes == 4) *(uint32_t __attribute__((address_space(1)))*)((a3d_gptr)cells + (column + ch) * 4u) = 0u;
      else *(unsigned long long __attribute__((address_space(1)))*)((a3d_gptr)cells + (size_t)(column + ch) * 8u) = 0ull;
    }
  }
  for (uint32_t k = block * THREADS + threadIdx.x; k < U.flag_words; k += blocks * THREADS)
    U.flags[(size_t)frame * U.flag_words + k] = 0u;
}

// Where the blurred grids of a batch of frames live (the context's grid scratch region):
// [n_frames x SC_STRIDE words of scalars][n_frames x capacity packed u32 / u64 cells][n_frames x capacity f64 cells (normalised values)]
struct GridBatch {
  uint32_t* scal = nullptr;
  void* packed = nullptr;  // u32 or u64 cells (bilateral.hip: Pack)
  double* blurred = nullptr;
  unsigned long long capacity = 0;  // cells per frame
  // `defer_unsplat`: the enqueue stopped before its last kernel (unsplat_kernel: the zeros go back where the splat wrote);
  // the caller's next kernel on the stream does it (unsplat_columns: level0_quad_kernel) and then commits `clean`
  Unsplat unsplat{};
  a3d_context::GridLayoutKey clean{};
};

// Enqueues min/max, grid sizing, splat and the fused blur for `n_frames` depth images ([n_frames][h][w] u16,
// device) on the context's stream WITHOUT a host round trip; afterwards out->blurred + f * capacity is frame f's
// blurred grid and out->scal + f * SC_STRIDE its scalars (dimensions, colour minimum, SC_TOO_BIG when its grid did
// not fit `capacity` cells: the caller checks that after its own synchronisation and calls again with more room).
a3d_status bilateral_grids_enqueue(a3d_context* ctx, const uint16_t* d_depth, uint32_t n_frames, uint32_t w, uint32_t h,
                                   double sigma_space, double sigma_color, unsigned long long capacity_cells,
                                   GridBatch* out, bool defer_unsplat = false);
// Cells a frame's grid needs: enough for a depth range of `depth_span` units (BilateralGrid::from_image, grid.rs:37-56)
unsigned long long bilateral_grid_cells(uint32_t w, uint32_t h, double sigma_space, double sigma_color, uint32_t depth_span);

}  // namespace a3d
