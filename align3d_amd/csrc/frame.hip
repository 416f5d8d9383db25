// RangeImageBuilder::build (src/range_image/builder.rs:74-91) entirely on the device, for a BATCH of frames: the
// frames enter HBM as u16 depth + u8 RGB (1.5 MB per 640x480 frame instead of ~12 MB of f32 pyramid levels) and
// leave as resident pyramids:
//   bilateral filter                 src/bilateral/edge_aware_filter.rs:126-135   (grids: bilateral.hip)
//   RangeImage::from_rgbd_image      src/range_image/structure.rs:56-95
//   RangeImage::compute_normals      src/range_image/structure.rs:184-262
//   RangeImage::pyr_scale_down       src/range_image/structure.rs:309-340, src/range_image/resize.rs:4-104
//   compute_intensity / _map         src/range_image/structure.rs:266-297, src/image/luma.rs:81-83,
//                                    src/intensity_map.rs:37-92
// Every kernel has a frame dimension (blockIdx.z, or .y where noted), so up to MAX_BATCH frames cost the same
// 10 launches as one frame; the per-frame arenas are laid out identically, so a kernel needs one base pointer per
// frame (FrameBases, passed by value) plus array offsets that are the same for all frames.
// The RGB blur of the pyramid (image 0.24.7 imageops::blur) is restated from its published algorithm
// like the oracle's: PARITY UNPINNED (no reference test pins its values).
#include <cmath>
#include <cstdlib>
#include <memory>

#include "bilateral.hpp"

using namespace a3d;

namespace {

// Frames per launch sequence.  The ten launches of a sequence have fixed costs (boundaries, ramps, the tail of the
// slowest block) that more frames share: kernels per frame measured 14.9 us at 16 frames per sequence, 14.0 at 32, 13.7
// at 64.  Against that, a sequence starts when ITS uploads have landed and the last one runs with nothing under it, so
// a build from host memory wants several sequences: a call's frames are split into equal chunks of at most MAX_BATCH
// (64 frames -> 2 x 32, 65 -> 33 + 32).
#ifndef A3D_MAX_BATCH
#define A3D_MAX_BATCH 48
#endif
#ifndef A3D_GRID_BUDGET_GB
#define A3D_GRID_BUDGET_GB 4  // a chunk's bilateral grids (24 B per cell of capacity) stay below this
#endif
constexpr uint32_t MAX_BATCH = A3D_MAX_BATCH;
struct FrameBases {
  char* arena[MAX_BATCH];
};

// One Vector3<f32> as ONE 12-byte store (global_store_dwordx3): three dword stores at a 12-byte lane stride make the
// memory pipeline touch every line of the wave's span three times, each time partially.
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef f32x3 __attribute__((aligned(4))) f32x3_u;
__device__ __forceinline__ void st_v3(float* base, size_t idx, V3 v) { *(f32x3_u*)(base + 3 * idx) = f32x3{v.x, v.y, v.z}; }
// The same off a block-uniform base pointer with a 32-bit pixel index (images are below 2^28 pixels: idx * 12 fits):
// `global_store … v_off, s[base]` — no 64-bit address arithmetic in the VALU (v_mad_u64_u32 runs at a quarter rate).
__device__ __forceinline__ void st_v3u(void* base, uint32_t idx, V3 v) {
  *(f32x3_u __attribute__((address_space(1)))*)((a3d_gptr)base + idx * 12u) = f32x3{v.x, v.y, v.z};
}
__device__ __forceinline__ void st_u8u(void* base, uint32_t idx, uint8_t v) {
  *(uint8_t __attribute__((address_space(1)))*)((a3d_gptr)base + idx) = v;
}
// Streaming forms for arrays nothing reads back soon (level 0's points, mask and normals; intensities and their maps).
__device__ __forceinline__ void st_v3u_stream(void* base, uint32_t idx, V3 v) {
#ifdef A3D_BUILDER_NO_NT
  st_v3u(base, idx, v);
#else
  __builtin_nontemporal_store(f32x3{v.x, v.y, v.z}, (f32x3_u __attribute__((address_space(1)))*)((a3d_gptr)base + idx * 12u));
#endif
}
__device__ __forceinline__ void st_u8u_stream(void* base, uint32_t idx, uint8_t v) {
#ifdef A3D_BUILDER_NO_NT
  st_u8u(base, idx, v);
#else
  __builtin_nontemporal_store(v, (uint8_t __attribute__((address_space(1)))*)((a3d_gptr)base + idx));
#endif
}
__device__ __forceinline__ V3 ld_v3g(const float* base, size_t idx) {
  const f32x3 v = *(const f32x3_u*)(base + 3 * idx);
  return V3{v.x, v.y, v.z};
}

// One pyramid level as the kernels see it: size and the byte offsets of its arrays inside a frame's arena.
struct LevelLayout {
  uint32_t w, h;
  size_t points, mask, normals, colors, intensities, imap;
};
constexpr uint32_t MAX_LEVELS = 16;
struct PyramidLayout {
  LevelLayout lv[MAX_LEVELS];
};

// get_neighborhood_mean_point (src/range_image/resize.rs:4-40) on the four candidates of a 2 x 2 block in block order
// (00, 01, 10, 11): among the entries whose SOURCE mask is 1, the one nearest to their mean (strict <, first wins);
// (0,0,0) when none is valid.  *n_valid = how many were.
__device__ __forceinline__ V3 pick_nearest_to_mean(const V3 (&cand)[4], const bool (&ok)[4], int* n_valid) {
  int n = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) n += ok[q] ? 1 : 0;
  *n_valid = n;
  V3 nearest{0.f, 0.f, 0.f};
  if (n > 0) {
    V3 sum{0.f, 0.f, 0.f};  // valid entries in block order, as the reference's `local` list
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ok[q]) sum = sum + cand[q];
    // sum / n (resize.rs:22-25): n is 1, 2, 3 or 4 — a division by 1, 2 or 4 is a multiplication by an exact power of two,
    // only n == 3 needs a real (IEEE) quotient
    V3 mean = sum * (n == 1 ? 1.0f : (n == 2 ? 0.5f : 0.25f));
    if (__builtin_amdgcn_ballot_w64(n == 3) != 0ull)  // (rare: three of the four valid — only at the edge of a hole)
      if (n == 3) mean = sum / 3.0f;
    float min_dist = 3.402823466e+38f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d = norm_squared(cand[q] - mean);
      if (ok[q] && d < min_dist) {  // strict <: the first minimum wins
        min_dist = d;
        nearest = cand[q];
      }
    }
  }
  return nearest;
}

#ifdef A3D_DIAGNOSTICS  // the round-2..5 level-0 kernel: the cross-check of level0_quad_kernel (A3D_BUILDER_L0=patch)
// ---- level 0: bilateral slice + back-projection + normals in one pass ---------------------------------------
// A block stages the points of a 32 x 16 patch (the 30 x 14 pixels it owns plus a one-pixel halo) in LDS: every
// staged pixel's filtered depth (BilateralGrid::slice, grid.rs:106-162 — same f64 arithmetic as slice_kernel) is
// back-projected (CameraIntrinsics::backproject, camera.rs:101-107; mask = depth > 0, structure.rs:56-95) by the
// thread that staged it; the owned pixels then take their normals from the staged neighbours
// (structure.rs:184-262).  The filtered depth image never exists in memory, points and mask are written once and
// the normals need no second read of them.  Halo pixels are sliced twice (22 % more slices) — cheaper than a
// round trip of the filtered image through HBM and two more launches.
constexpr int ST_W = 32, ST_H = 16, OWN_W = ST_W - 2, OWN_H = ST_H - 2;

// 256 threads per block, TWO staged pixels per thread (rows ly and ly + 8 of the 32 x 16 patch): the kernel is bound by
// the latency of its dependent phases (depth load -> eight grid gathers -> LDS -> normals -> stores) times the rounds of
// resident blocks, so a wave that carries two independent pixel chains halves the rounds.
constexpr int L0_PPT = 2, L0_THREADS = ST_W * ST_H / L0_PPT;
#ifndef A3D_L0_PROBE  // (scripts/build_frame_variant.sh -DA3D_L0_PROBE=n): 1 no level-0 stores, 2 no normals,
#define A3D_L0_PROBE 0  // 3 no level-1 picks, 4 nothing but the staging
#endif
template <bool FILTER>
__global__ void __launch_bounds__(L0_THREADS)
    level0_kernel(const uint16_t* __restrict__ depth, uint32_t w, uint32_t h, double inv_ss, double inv_sc,
                  const double* __restrict__ grids, unsigned long long capacity, uint32_t* __restrict__ scal, float fx,
                  float fy, float cx, float cy, float scale, FrameBases bases, size_t off_points, size_t off_mask,
                  size_t off_normals, bool with_normals, LevelLayout L1, bool emit_l1) {
  __shared__ float sp[3][ST_H][ST_W + 1];
  __shared__ float sn[3][OWN_H][OWN_W + 1];  // the owned pixels' normals, for the fused level-1 pick
  __shared__ uint8_t sm[ST_H][ST_W];         // their masks (1: depth > 0)
  const uint32_t f = blockIdx.z;
  const int lx = threadIdx.x % ST_W, ly0 = threadIdx.x / ST_W;
  const int col = (int)blockIdx.x * OWN_W + lx - 1;
  int ly[L0_PPT], row[L0_PPT];
  bool in[L0_PPT];
  uint32_t d[L0_PPT];
  float px[L0_PPT], py[L0_PPT], pz[L0_PPT];
#pragma unroll
  for (int k = 0; k < L0_PPT; ++k) {
    ly[k] = ly0 + k * (ST_H / L0_PPT);
    row[k] = (int)blockIdx.y * OWN_H + ly[k] - 1;
    in[k] = col >= 0 && col < (int)w && row[k] >= 0 && row[k] < (int)h;
    // (both loads issued before either is used; the frame's image off a uniform base, the pixel by a 24-bit multiply-add)
    d[k] = in[k] ? *(const uint16_t __attribute__((address_space(1)))*)((a3d_gptr_c)(depth + (size_t)f * w * h) +
                                                                        (__umul24((uint32_t)row[k], w) + (uint32_t)col) * 2u)
                 : 0u;
    px[k] = py[k] = pz[k] = 0.f;
  }
  const DivBy dfx = div_prepare(fx), dfy = div_prepare(fy);
  const bool focal_ok = div_den_ok(fx) & div_den_ok(fy);
#pragma unroll
  for (int k = 0; k < L0_PPT; ++k) {
    if (in[k]) {
      if (FILTER) {
        uint32_t* sc = scal + f * SC_STRIDE;
        GridDims g;
        uint32_t cmin;
        if (dyn_dims(sc, &g, &cmin)) {  // (false: this frame's grid did not fit; the host grows the region and repeats)
          // (the column part of the slice is shared by the thread's two pixels: the compiler keeps one copy; moving the row
          // and column parts to per-tile LDS tables was measured in round 4: 75.3 against 73.4 us per 16 frames, dropped)
          uint16_t v;
          if (!slice_pixel<true>(d[k], (uint32_t)row[k], (uint32_t)col, inv_ss, inv_sc, cmin, g, grids + f * capacity, &v))
            atomicOr(&sc[SC_OVERFLOW], 1u);  // the reference's .unwrap() would panic
          d[k] = v;
        }
      }
      if (d[k] > 0) {  // CameraIntrinsics::backproject (camera.rs:101-107): x = (u - cx) z / fx, y = (v - cy) z / fy
        pz[k] = (float)d[k] * scale;
        const float ax = ((float)col - cx) * pz[k], ay = ((float)row[k] - cy) * pz[k];
        // the two IEEE quotients by the (uniform) focal lengths through their refined reciprocals (div_by: bit-identical
        // to `/` inside its operand range, devmath.hpp); plain division for the wave when anything is outside it
        const bool fast = focal_ok & div_num_ok(ax) & div_num_ok(ay);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast) != 0ull, 0)) {
          px[k] = ax / fx, py[k] = ay / fy;
        } else {
          px[k] = ax == 0.0f ? ax : div_by(ax, dfx), py[k] = ay == 0.0f ? ay : div_by(ay, dfy);
        }
      }
    }
    sp[0][ly[k]][lx] = px[k], sp[1][ly[k]][lx] = py[k], sp[2][ly[k]][lx] = pz[k];
    sm[ly[k]][lx] = d[k] > 0 ? 1 : 0;
  }
  __syncthreads();
  char* base = bases.arena[f];
  auto at = [&](int y, int x) { return V3{sp[0][y][x], sp[1][y][x], sp[2][y][x]}; };
  bool owned[L0_PPT];
  V3 nrm[L0_PPT];
#pragma unroll
  for (int k = 0; k < L0_PPT; ++k) {
    owned[k] = in[k] && lx != 0 && lx != ST_W - 1 && ly[k] != 0 && ly[k] != ST_H - 1;  // (the rest is halo)
    nrm[k] = V3{0.f, 0.f, 0.f};
    if (owned[k] && with_normals && A3D_L0_PROBE != 2 && A3D_L0_PROBE != 4) {
      // an invalid neighbour's point is (0,0,0) already (= get_point(...).unwrap_or_else(zeros)); so is everything
      // outside the image; the centre is used as stored, its mask is NOT checked (structure.rs:207)
      nrm[k] = normal_from_neighbours_dev(V3{px[k], py[k], pz[k]}, at(ly[k], lx - 1), at(ly[k], lx + 1), at(ly[k] - 1, lx),
                                      at(ly[k] + 1, lx));
      if (emit_l1) sn[0][ly[k] - 1][lx - 1] = nrm[k].x, sn[1][ly[k] - 1][lx - 1] = nrm[k].y, sn[2][ly[k] - 1][lx - 1] = nrm[k].z;
    }
  }
  if (emit_l1) __syncthreads();  // (before the stores below: the level-1 picks then run under them)
#pragma unroll
  for (int k = 0; k < L0_PPT; ++k)
    if (owned[k] && ((A3D_L0_PROBE != 1 && A3D_L0_PROBE != 4) || px[k] == 12345.678f)) {
      const uint32_t idx = __umul24((uint32_t)row[k], w) + (uint32_t)col;
      st_v3u_stream(base + off_points, idx, V3{px[k], py[k], pz[k]});
      st_u8u_stream(base + off_mask, idx, d[k] > 0 ? 1 : 0);
      if (with_normals) st_v3u_stream(base + off_normals, idx, nrm[k]);
    }
  if (!emit_l1 || A3D_L0_PROBE == 3 || A3D_L0_PROBE == 4) return;
  // ---- level 1 of the pyramid from the staged level-0 patch (pyr_scale_down: resize_range_points / _normals,
  // src/range_image/resize.rs:42-104): the patch origin is even and the image sides are even (the host checks), so the
  // 2 x 2 blocks of the 30 x 14 owned pixels are whole and block (dv, du) is source pixels (2 dv .. 2 dv + 1, 2 du ..).
  // Saves reading level 0's points, mask and normals again (123 of the 154 MB a separate kernel moves per 16 frames).
  // 15 x 7 level-1 pixels per patch, each picked twice (points, normals): 210 tasks on the block's first 210 threads
  constexpr int L1W = OWN_W / 2, L1N = (OWN_W / 2) * (OWN_H / 2);
  const int t = (int)threadIdx.x;
  const bool normals_task = t >= L1N;
  const int task = normals_task ? t - L1N : t;
  if (task >= L1N || (normals_task && !with_normals)) return;
  const int oy = 2 * (task / L1W), ox = 2 * (task % L1W);  // owned coordinates of the 2 x 2 block's first pixel
  const uint32_t r0 = blockIdx.y * OWN_H + (uint32_t)oy, c0 = blockIdx.x * OWN_W + (uint32_t)ox;
  if (r0 >= h || c0 >= w) return;
  V3 cand[4];
  bool ok[4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = a * 2 + b;
      cand[q] = normals_task ? V3{sn[0][oy + a][ox + b], sn[1][oy + a][ox + b], sn[2][oy + a][ox + b]}
                             : at(oy + 1 + a, ox + 1 + b);
      ok[q] = sm[oy + 1 + a][ox + 1 + b] == 1;
    }
  int n_valid;
  const V3 pk = pick_nearest_to_mean(cand, ok, &n_valid);
  const uint32_t i1 = __umul24(r0 >> 1, L1.w) + (c0 >> 1);
  st_v3u(base + (normals_task ? L1.normals : L1.points), i1, pk);  // (a per-lane choice of array: two bases, one select)
  if (!normals_task) st_u8u(base + L1.mask, i1, n_valid > 0 ? 1 : 0);
}
#endif  // A3D_DIAGNOSTICS

// ---- level 0, round 6: a block owns an aligned 32 x 32 patch, a thread a 2 x 2 quad of it ------------------------
// What the round-5 probes of the kernel above showed (profiles/round6_level0_phase_probes.txt; us per 32 frames):
// slice + back-projection + staging alone 73, + normals + level-1 picks 115, + the level-0 stores 154 — the stores were a
// quarter of it and the picks (210 tasks behind a second barrier, candidates back from LDS) a sixth, although together they
// are a fifth of the instructions.  The 30-pixel-wide owned patches start at multiples of 360 bytes: every row of every
// patch ends inside a 128-byte line its neighbour block completes later, and streaming stores leave such lines partial
// (write traffic 1.24 x the arrays).  So:
//  * owned patch = 32 x 32 pixels at multiples of 32: a patch row of points or normals is 384 bytes = three whole lines
//    when the image width is a multiple of 32 / 3 pixels (640: yes), a mask row one 32-byte sector; nothing is shared;
//  * a thread owns the 2 x 2 quad (2 ty .. 2 ty + 1, 2 tx .. 2 tx + 1): both pyramid picks of its level-1 pixel take their
//    four candidates from the thread's own registers (no second staging array, no barrier, every lane busy), two of a
//    pixel's four neighbours for the normals are the thread's own, and the slice's row / column parts serve two pixels each;
//  * the one-pixel halo (4 x 32 pixels) is sliced by the block's first 128 threads as a fifth pixel: 4.5 slices per thread for
//    four owned pixels (0.89) against 2 for 1.64 (0.82), and a 640 x 480 image is 20 x 15 whole patches (the 30 x 14 grid
//    overhung it by 5 %);
//  * level 2 as well when the sides are multiples of four: the level-1 picks of a patch are 16 x 16 = whole 2 x 2 blocks, so
//    they meet in LDS and 128 threads pick level 2's 8 x 8 points and normals: resize_pick_kernel's launch (0.57 us per
//    frame) is gone for three-level pyramids.
// The arithmetic per pixel is the kernel above's, expression by expression (slice_pixel_axes, the back-projection's
// shared-reciprocal quotients, normal_from_neighbours_dev, pick_nearest_to_mean): same bits.
constexpr int QS = 32, QT = QS / 2;

// CameraIntrinsics::backproject (camera.rs:101-107) of a filtered depth: x = (u - cx) z / fx, y = (v - cy) z / fy; (0,0,0)
// for an invalid pixel (mask = depth > 0, structure.rs:56-95).
__device__ __forceinline__ V3 backproject_px(uint32_t d, int row, int col, float fx, float fy, float cx, float cy, float scale,
                                             const DivBy dfx, const DivBy dfy, bool focal_ok) {
  V3 p{0.f, 0.f, 0.f};
  if (d > 0) {
    p.z = (float)d * scale;
    const float ax = ((float)col - cx) * p.z, ay = ((float)row - cy) * p.z;
    // the two IEEE quotients by the (uniform) focal lengths through their refined reciprocals (div_by: bit-identical to `/`
    // inside its operand range, devmath.hpp); plain division for the wave when anything is outside it
    const bool fast = focal_ok & div_num_ok(ax) & div_num_ok(ay);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast) != 0ull, 0)) {
      p.x = ax / fx, p.y = ay / fy;
    } else {
      p.x = ax == 0.0f ? ax : div_by(ax, dfx), p.y = ay == 0.0f ? ay : div_by(ay, dfy);
    }
  }
  return p;
}

template <bool FILTER>
__global__ void __launch_bounds__(256)
    level0_quad_kernel(const uint16_t* __restrict__ depth, uint32_t w, uint32_t h, double inv_ss, double inv_sc,
                       const double* __restrict__ grids, unsigned long long capacity, uint32_t* __restrict__ scal, float fx,
                       float fy, float cx, float cy, float scale, FrameBases bases, size_t off_points, size_t off_mask,
                       size_t off_normals, bool with_normals, LevelLayout L1, bool emit_l1, LevelLayout L2, bool emit_l2,
                       bool l2_is_last, Unsplat unsplat, uint32_t patches_x, uint32_t patches_y) {
  __shared__ float sp[3][QS + 2][QS + 3];  // the patch's points at (y + 1, x + 1), halo included
  __shared__ float s1[2][3][QT][QT + 1];   // level-1 picks (0: points, 1: normals) for the level-2 picks
  __shared__ uint8_t s1m[QT][QT];          // level-1 masks
  __shared__ __attribute__((aligned(4))) uint8_t sm[QS][QS];  // the owned pixels' masks (1: depth > 0), for the row-ordered stores
#ifndef A3D_LQ_XCD  // 1: the patches placed XCD by XCD (a frame's neighbouring patches share an L2) — measured 144 against
#define A3D_LQ_XCD 0  // 139-140 us per 32 frames in the plain order (patches share few grid lines; the stores spread worse)
#endif
  uint32_t f = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
  if (patches_x) {  // 1-D launch of frames x patches_y x patches_x blocks
    const uint32_t v = A3D_LQ_XCD ? xcd_contiguous_index(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t per_frame = patches_x * patches_y;
    f = v / per_frame;
    const uint32_t in_frame = v - f * per_frame;
    by = in_frame / patches_x, bx = in_frame - by * patches_x;
  }
  const int t = (int)threadIdx.x, tx = t & (QT - 1), ty = t >> 4;
  const int r0 = (int)by * QS, c0 = (int)bx * QS;
  const uint16_t __attribute__((address_space(1)))* dimg =
      (const uint16_t __attribute__((address_space(1)))*)(depth + (size_t)f * w * h);
  // (unconditional loads at clamped coordinates: a conditional load is a branch with its own wait, and the kernel lives on
  // how many loads a thread has in flight; a pixel outside the image counts as depth 0 and is never stored)
  auto load_depth = [&](int row, int col) -> uint32_t {
    const uint32_t r = (uint32_t)min(max(row, 0), (int)h - 1), c = (uint32_t)min(max(col, 0), (int)w - 1);
    return *(const uint16_t __attribute__((address_space(1)))*)((a3d_gptr_c)dimg + (__umul24(r, w) + c) * 2u);
  };
  // ---- the thread's pixels: its quad and (threads 0 .. 127: the first two waves) one pixel of the halo ----
  int row[2], col[2];
  bool rin[2], cin[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    row[a] = r0 + 2 * ty + a, col[a] = c0 + 2 * tx + a;
    rin[a] = row[a] < (int)h, cin[a] = col[a] < (int)w;
  }
  const bool halo_thread = t < 4 * QS;  // (wave-uniform)
  const int side = t >> 5, hi = t & (QS - 1);  // 0: the row above, 1: the row below, 2: the column left, 3: the column right
  const int hy = side == 0 ? 0 : (side == 1 ? QS + 1 : hi + 1), hx = side == 2 ? 0 : (side == 3 ? QS + 1 : hi + 1);  // in sp
  const int hrow = r0 + hy - 1, hcol = c0 + hx - 1;
  const bool hin = halo_thread && hrow >= 0 && hrow < (int)h && hcol >= 0 && hcol < (int)w;
  uint32_t d[2][2], dh = 0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) d[a][b] = load_depth(row[a], col[b]);
  if (halo_thread) dh = load_depth(hrow, hcol);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) d[a][b] = (rin[a] && cin[b]) ? d[a][b] : 0u;
  dh = hin ? dh : 0u;
  const DivBy dfx = div_prepare(fx), dfy = div_prepare(fy);
  const bool focal_ok = div_den_ok(fx) & div_den_ok(fy);
  GridDims g{0, 0, 0};
  bool grid_ok = false;
  if (FILTER) {
    uint32_t* sc = scal + f * SC_STRIDE;
    uint32_t cmin;
    grid_ok = dyn_dims(sc, &g, &cmin);
    if (grid_ok) {  // (false: this frame's grid did not fit; the host grows the region and repeats)
      // BilateralGrid::slice (grid.rs:106-162) of every pixel in the image, zeros included.
      const double* grid = grids + f * capacity;
      SliceAxis ry[2], cxs[2];  // the row / column parts (grid.rs:132-146) serve two pixels each
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        ry[a] = slice_axis_spatial((uint32_t)min(row[a], (int)h - 1), inv_ss, g.gh);
        cxs[a] = slice_axis_spatial((uint32_t)min(col[a], (int)w - 1), inv_ss, g.gw);
      }
      SliceAxis hry{}, hcx{};
      if (halo_thread) {
        hry = slice_axis_spatial((uint32_t)min(max(hrow, 0), (int)h - 1), inv_ss, g.gh);
        hcx = slice_axis_spatial((uint32_t)min(max(hcol, 0), (int)w - 1), inv_ss, g.gw);
      }
      auto colour_axis = [&](uint32_t dv) { return slice_axis((double)(dv - cmin) * inv_sc + 2.0, g.gd); };
      SliceAxis cz[2][2], czh{};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) cz[a][b] = colour_axis(d[a][b]);
      if (halo_thread) czh = colour_axis(dh);
      // (Round 6, measured and removed: the grid cells under the patch staged in LDS — the block's valid pixels' channel range
      // found by a block-wide min / max, the <= 10 x 10 (row, column)s x <= 16 channels loaded along the channels, the
      // eight gathers of a pixel then ds_read_b64 — on the idea that a CU's L1 turns line requests around too slowly for
      // eight scattered gathers per pixel.  Bit-identical, 178 against 137 us per 32 frames: the two extra barriers and the
      // third dependent trip to memory (depth -> range -> cells) cost far more than the gathers' queueing.)
      auto gather = [&](const SliceAxis& ry_, const SliceAxis& cx_, const SliceAxis& cz_, SliceTaps* tp) {
        tp->za = cz_.frac;
        const uint32_t g0 = __umul24(ry_.lo, g.gw), g1 = __umul24(ry_.hi, g.gw);
        const uint32_t b00 = __umul24(g0 + cx_.lo, g.gd), b01 = __umul24(g0 + cx_.hi, g.gd), b10 = __umul24(g1 + cx_.lo, g.gd),
                       b11 = __umul24(g1 + cx_.hi, g.gd);
        tp->v[0] = cell_value32(grid, b00, cz_.lo), tp->v[1] = cell_value32(grid, b01, cz_.lo);
        tp->v[2] = cell_value32(grid, b10, cz_.lo), tp->v[3] = cell_value32(grid, b11, cz_.lo);
        tp->v[4] = cell_value32(grid, b00, cz_.hi), tp->v[5] = cell_value32(grid, b01, cz_.hi);
        tp->v[6] = cell_value32(grid, b10, cz_.hi), tp->v[7] = cell_value32(grid, b11, cz_.hi);
      };
      bool overflow = false;
      auto finish = [&](const SliceTaps& tp, double ya, double xa, bool in) -> uint32_t {
        uint16_t v;
        const bool fits = slice_combine(tp, ya, xa, &v);
        overflow |= in && !fits;
        return in ? (uint32_t)v : 0u;
      };
      // The gathers of a quad row's two pixels (and the halo pixel's, with the first row) are in flight together: the blurred
      // grids of a launch sequence (150 MB) do not fit the L2, a gather is a trip to the Infinity Cache.  Measured per 32
      // frames: one pixel at a time 150 us, two 127-138, two + the halo pixel on its own 145, all five 151 (131 registers).
      {
        SliceTaps taps[2], htaps;
        gather(ry[0], cxs[0], cz[0][0], &taps[0]);
        gather(ry[0], cxs[1], cz[0][1], &taps[1]);
        if (halo_thread) gather(hry, hcx, czh, &htaps);
        d[0][0] = finish(taps[0], ry[0].frac, cxs[0].frac, rin[0] && cin[0]);
        d[0][1] = finish(taps[1], ry[0].frac, cxs[1].frac, rin[0] && cin[1]);
        __builtin_amdgcn_sched_barrier(0);
        gather(ry[1], cxs[0], cz[1][0], &taps[0]);
        gather(ry[1], cxs[1], cz[1][1], &taps[1]);
        if (halo_thread) dh = finish(htaps, hry.frac, hcx.frac, hin);
        d[1][0] = finish(taps[0], ry[1].frac, cxs[0].frac, rin[1] && cin[0]);
        d[1][1] = finish(taps[1], ry[1].frac, cxs[1].frac, rin[1] && cin[1]);
      }
      if (overflow) atomicOr(&sc[SC_OVERFLOW], 1u);  // the reference's .unwrap() would panic
    }
  }
  V3 P[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      P[a][b] = backproject_px(d[a][b], row[a], col[b], fx, fy, cx, cy, scale, dfx, dfy, focal_ok);  // (outside the image: d = 0)
      const int y = 2 * ty + a + 1, x = 2 * tx + b + 1;
      sp[0][y][x] = P[a][b].x, sp[1][y][x] = P[a][b].y, sp[2][y][x] = P[a][b].z;
      sm[y - 1][x - 1] = d[a][b] > 0 ? 1 : 0;
    }
  if (halo_thread) {
    const V3 ph = backproject_px(dh, hrow, hcol, fx, fy, cx, cy, scale, dfx, dfy, focal_ok);
    sp[0][hy][hx] = ph.x, sp[1][hy][hx] = ph.y, sp[2][hy][hx] = ph.z;
  }
  __syncthreads();
#ifdef A3D_DIAGNOSTICS
  // A3D_BUILDER_UNSPLAT=fused: the filter's last step, taken over from unsplat_kernel: the blur is done with the packed cells,
  // so the grid (row, column)s that begin in this patch get their zeros back here — a few dozen small stores per block,
  // behind the kernel's last global load (in front of the gathers a later load's wait became a wait for these stores' round
  // trip: loads and stores share one in-order counter).  Measured: 140 against 123 us per 32 frames — more than the
  // 10 us launch it saves; not the product's path.
  if (FILTER && grid_ok && unsplat.packed)
    unsplat_columns(unsplat, f, g, inv_ss, r0, min(r0 + QS, (int)h), c0, min(c0 + QS, (int)w), by * patches_x + bx,
                    patches_x * patches_y);
#endif
  char* base = bases.arena[f];
  auto at = [&](int y, int x) { return V3{sp[0][y][x], sp[1][y][x], sp[2][y][x]}; };
  // ---- normals (structure.rs:184-262): an invalid neighbour's point is (0,0,0) already (= get_point(..).unwrap_or_else(
  // zeros)), so is everything outside the image; the centre is used as stored, its mask is NOT checked (structure.rs:207).
  // Of a pixel's four neighbours two are the thread's own.
  V3 N[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      N[a][b] = V3{0.f, 0.f, 0.f};
      if (with_normals && rin[a] && cin[b]) {
        const int y = 2 * ty + a + 1, x = 2 * tx + b + 1;
        const V3 left = b ? P[a][0] : at(y, x - 1), right = b ? at(y, x + 1) : P[a][1];
        const V3 top = a ? P[0][b] : at(y - 1, x), bottom = a ? at(y + 1, x) : P[1][b];
        N[a][b] = normal_from_neighbours_dev(P[a][b], left, right, top, bottom);
      }
    }
#ifndef A3D_LQ_PROBE  // (scripts/build_frame_variant.sh -DA3D_LQ_PROBE=n: 1 no level-0 stores, 3 no picks)
#define A3D_LQ_PROBE 0
#endif
  // ---- level-0 stores in ROW order: a store instruction of the quad layout would write every other 12-byte element of
  // four rows — twelve half-written lines that the next instruction completes — and cost 80 of the kernel's 180 us per 32
  // frames (plain instead of streaming stores: 66).  The arrays leave in the order they lie in memory instead: store k of
  // thread t is pixel (t / 32 + 8 k, t % 32) of the patch, read back from LDS, so a wave's instruction writes two whole
  // patch rows (2 x 384 bytes of points or normals = six whole lines).  The normals take the points' place in `sp` once
  // every wave has stored its points (two more barriers, no more LDS: seven blocks per CU).
  const int sx = t & (QS - 1), sy0 = t >> 5;
  const bool scol_in = c0 + sx < (int)w;
  auto store_rows = [&](size_t off, bool with_mask) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = sy0 + 8 * k;
      if (scol_in && r0 + y < (int)h && (A3D_LQ_PROBE != 1 || sp[0][y + 1][sx + 1] == 12345.678f)) {
        const uint32_t idx = __umul24((uint32_t)(r0 + y), w) + (uint32_t)(c0 + sx);
        st_v3u_stream(base + off, idx, at(y + 1, sx + 1));
        if (with_mask) st_u8u_stream(base + off_mask, idx, sm[y][sx]);
      }
    }
  };
  // the patch's masks are 32 rows x 32 bytes = 256 words: ONE word per thread (row t / 8, pixels 4 (t % 8) ..) when the
  // image width is a multiple of four (rows of the mask array then start word-aligned), else a byte per pixel with the points
  const bool mask_words = (w & 3u) == 0;  // (block-uniform)
  if (mask_words) {
    const int my = t >> 3, mx = 4 * (t & 7);
    if (r0 + my < (int)h && c0 + mx < (int)w && A3D_LQ_PROBE != 1)
      __builtin_nontemporal_store(*(const uint32_t*)&sm[my][mx],
                                  (uint32_t __attribute__((address_space(1)))*)((a3d_gptr)(base + off_mask) +
                                                                               (__umul24((uint32_t)(r0 + my), w) + (uint32_t)(c0 + mx))));
  }
  store_rows(off_points, !mask_words);
  if (with_normals) {  // (block-uniform)
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int y = 2 * ty + a + 1, x = 2 * tx + b + 1;
        sp[0][y][x] = N[a][b].x, sp[1][y][x] = N[a][b].y, sp[2][y][x] = N[a][b].z;
      }
    __syncthreads();
    store_rows(off_normals, false);
  }
  if (!emit_l1 || A3D_LQ_PROBE == 3) return;  // (block-uniform)
  // ---- level 1 (pyr_scale_down: resize_range_points / _normals, src/range_image/resize.rs:42-104): the image sides are
  // even (the host checks), so the quad is source block (2 dv .. 2 dv + 1, 2 du .. 2 du + 1) of level-1 pixel (dv, du) and
  // lies inside the image whole or not at all; candidates in block order 00, 01, 10, 11 from the registers
  const bool quad_in = rin[0] && cin[0];
  V3 cand[4];
  bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) cand[q] = P[q >> 1][q & 1], ok[q] = d[q >> 1][q & 1] > 0;
  int n_valid = 0;
  const V3 pk_p = pick_nearest_to_mean(cand, ok, &n_valid);
  V3 pk_n{0.f, 0.f, 0.f};
  if (with_normals) {
#pragma unroll
    for (int q = 0; q < 4; ++q) cand[q] = N[q >> 1][q & 1];
    int unused;
    pk_n = pick_nearest_to_mean(cand, ok, &unused);
  }
  if (quad_in) {
    const uint32_t i1 = __umul24((uint32_t)row[0] >> 1, L1.w) + ((uint32_t)col[0] >> 1);
    if (emit_l2) {  // (nothing in the builder reads level 1 again)
      st_v3u_stream(base + L1.points, i1, pk_p);
      st_u8u_stream(base + L1.mask, i1, n_valid > 0 ? 1 : 0);
      if (with_normals) st_v3u_stream(base + L1.normals, i1, pk_n);
    } else {
      st_v3u(base + L1.points, i1, pk_p);
      st_u8u(base + L1.mask, i1, n_valid > 0 ? 1 : 0);
      if (with_normals) st_v3u(base + L1.normals, i1, pk_n);
    }
  }
  if (!emit_l2) return;  // (block-uniform)
  // ---- level 2 from the patch's 16 x 16 level-1 picks: the level-1 sides are even too (the host checks), blocks are whole
  s1[0][0][ty][tx] = pk_p.x, s1[0][1][ty][tx] = pk_p.y, s1[0][2][ty][tx] = pk_p.z;
  s1[1][0][ty][tx] = pk_n.x, s1[1][1][ty][tx] = pk_n.y, s1[1][2][ty][tx] = pk_n.z;
  s1m[ty][tx] = (quad_in && n_valid > 0) ? 1 : 0;
  __syncthreads();
  const int which = t >> 6;  // 0: points (and the mask), 1: normals; threads 128 .. 255 have no task
  if (which > 1 || (which == 1 && !with_normals)) return;
  const int ly = (t >> 3) & 7, lx = t & 7;
  const uint32_t r2 = ((uint32_t)r0 >> 2) + (uint32_t)ly, c2 = ((uint32_t)c0 >> 2) + (uint32_t)lx;
  if (r2 >= L2.h || c2 >= L2.w) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int y = 2 * ly + (q >> 1), x = 2 * lx + (q & 1);
    cand[q] = V3{s1[which][0][y][x], s1[which][1][y][x], s1[which][2][y][x]};
    ok[q] = s1m[y][x] == 1;
  }
  const V3 pk2 = pick_nearest_to_mean(cand, ok, &n_valid);
  const uint32_t i2 = __umul24(r2, L2.w) + c2;
  char* dst = base + (which ? L2.normals : L2.points);
  if (l2_is_last) {
    st_v3u_stream(dst, i2, pk2);
    if (!which) st_u8u_stream(base + L2.mask, i2, n_valid > 0 ? 1 : 0);
  } else {
    st_v3u(dst, i2, pk2);
    if (!which) st_u8u(base + L2.mask, i2, n_valid > 0 ? 1 : 0);
  }
}

// Colours that arrived as ONE upload for the whole chunk ([F][h][w][3] in the staging region) to each frame's arena.
__global__ void __launch_bounds__(256) scatter_colors_kernel(const uint4* __restrict__ staged, size_t bytes_per_frame,
                                                             size_t off_colors, FrameBases bases) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n16 = bytes_per_frame / 16;
  const uint8_t* src = (const uint8_t*)staged + (size_t)blockIdx.z * bytes_per_frame;
  uint8_t* dst = (uint8_t*)(bases.arena[blockIdx.z] + off_colors);
  if (i < n16) ((uint4*)dst)[i] = ((const uint4*)src)[i];
  if (i == 0)
    for (size_t k = n16 * 16; k < bytes_per_frame; ++k) dst[k] = src[k];
}

// rgb_to_luma_u8 (src/image/luma.rs:81-83): (r*0.3 + g*0.59 + b*0.11) as u8 (saturating truncation)
__device__ __forceinline__ uint8_t luma_u8(const uint8_t* __restrict__ rgb, uint32_t i) {
  const float l = (float)rgb[3 * i] * 0.3f + (float)rgb[3 * i + 1] * 0.59f + (float)rgb[3 * i + 2] * 0.11f;
  return l >= 255.0f ? 255 : (l <= 0.0f ? 0 : (uint8_t)l);
}

// compute_intensity + compute_intensity_map (structure.rs:266-297) for every level of every frame in one launch:
// blockIdx.y = level, blockIdx.z = frame.
// IntensityMap::from_luma_image (src/intensity_map.rs:37-92) including the incomplete border: rows h, h+1 copy row
// h-1 for cols < w-1; cols w, w+1 copy col w-1 for rows < h-1; (h, w) and (h+1, w+1) take the last pixel; the other
// border cells stay 0.  The luma of a cell's pixel is computed from the level's colours on the fly; interior cells
// also store it as the level's `intensities`.
// QUADS (the width is a multiple of four): a thread takes four interior cells of a row — 12 colour bytes as three
// aligned words, four lumas as one word, four map cells — and the threads behind the interior take the border cells
// one each; otherwise one thread per cell of the (h+2) x (w+2) map.
__device__ __forceinline__ uint8_t luma_of(uint32_t r, uint32_t g, uint32_t b) {
  const float l = (float)r * 0.3f + (float)g * 0.59f + (float)b * 0.11f;
  return l >= 255.0f ? 255 : (l <= 0.0f ? 0 : (uint8_t)l);
}
__device__ __forceinline__ float imap_border_cell(const uint8_t* __restrict__ rgb, uint32_t w, uint32_t h, uint32_t r, uint32_t c) {
  if (r >= h && c + 1 < w) return (float)luma_u8(rgb, (h - 1) * w + c) / 255.0f;
  if (c >= w && r + 1 < h) return (float)luma_u8(rgb, r * w + (w - 1)) / 255.0f;
  if ((r == h && c == w) || (r == h + 1 && c == w + 1)) return (float)luma_u8(rgb, (h - 1) * w + (w - 1)) / 255.0f;
  return 0.0f;
}
template <bool QUADS>
__global__ void __launch_bounds__(256) luma_imap_kernel(PyramidLayout layout, FrameBases bases) {
  const LevelLayout L = layout.lv[blockIdx.y];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t w = L.w, h = L.h, mw = w + 2, mh = h + 2;
  char* base = bases.arena[blockIdx.z];
  const uint8_t* __restrict__ rgb = (const uint8_t*)(base + L.colors);
  float* __restrict__ imap = (float*)(base + L.imap);
  if (QUADS) {
    const uint32_t qpr = w / 4, n_quads = h * qpr, n_border = 2 * mw + 2 * h;
    if (i < n_quads) {
      const uint32_t r = i / qpr, c = 4 * (i % qpr), px = r * w + c;
      const uint32_t* src = (const uint32_t*)(rgb + (size_t)px * 3);  // 12 bytes at a multiple of 12: word-aligned
      const uint32_t a = src[0], b = src[1], d = src[2];
      const uint8_t l0 = luma_of(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u);
      const uint8_t l1 = luma_of(a >> 24, b & 255u, (b >> 8) & 255u);
      const uint8_t l2 = luma_of((b >> 16) & 255u, b >> 24, d & 255u);
      const uint8_t l3 = luma_of((d >> 8) & 255u, (d >> 16) & 255u, d >> 24);
      const uint32_t packed4 = (uint32_t)l0 | ((uint32_t)l1 << 8) | ((uint32_t)l2 << 16) | ((uint32_t)l3 << 24);
      float* o = imap + (size_t)r * mw + c;
#ifdef A3D_BUILDER_NO_NT
      *(uint32_t*)((uint8_t*)(base + L.intensities) + px) = packed4;
      o[0] = (float)l0 / 255.0f, o[1] = (float)l1 / 255.0f, o[2] = (float)l2 / 255.0f, o[3] = (float)l3 / 255.0f;
#else  // streaming stores: the alignment reads these, much later
      __builtin_nontemporal_store(packed4, (uint32_t*)((uint8_t*)(base + L.intensities) + px));
      __builtin_nontemporal_store((float)l0 / 255.0f, o), __builtin_nontemporal_store((float)l1 / 255.0f, o + 1);
      __builtin_nontemporal_store((float)l2 / 255.0f, o + 2), __builtin_nontemporal_store((float)l3 / 255.0f, o + 3);
#endif
    } else if (i < n_quads + n_border) {
      const uint32_t j = i - n_quads;
      uint32_t r, c;
      if (j < 2 * mw) r = h + j / mw, c = j % mw;
      else r = (j - 2 * mw) / 2, c = w + ((j - 2 * mw) & 1u);
      imap[(size_t)r * mw + c] = imap_border_cell(rgb, w, h, r, c);
    }
    return;
  }
  if (i >= mw * mh) return;
  const uint32_t r = i / mw, c = i % mw;
  float v;
  if (r < h && c < w) {
    const uint8_t l = luma_u8(rgb, r * w + c);
    ((uint8_t*)(base + L.intensities))[r * w + c] = l;
    v = (float)l / 255.0f;
  } else {
    v = imap_border_cell(rgb, w, h, r, c);
  }
  imap[i] = v;
}

// get_neighborhood_mean_point over every 2x2 block (src/range_image/resize.rs:4-40): among the entries
// whose SOURCE mask is 1, the one nearest to their mean (strict <, first wins).  Used for points
// (writes the destination mask) and for normals (mask output null).
// blockIdx.y = 0 picks the points (and writes the destination mask), blockIdx.y = 1 the normals (if any);
// blockIdx.z = frame.  The four candidates stay in registers (no dynamically indexed private array).
__global__ void __launch_bounds__(256) resize_pick_kernel(LevelLayout S, LevelLayout D, FrameBases bases) {
  const uint32_t sw = S.w, sh = S.h, dw = D.w, dh = D.h;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dw * dh) return;
  char* base = bases.arena[blockIdx.z];
  const bool normals = blockIdx.y == 1;
  const float* __restrict__ src = (const float*)(base + (normals ? S.normals : S.points));
  float* __restrict__ dst = (float*)(base + (normals ? D.normals : D.points));
  const uint8_t* __restrict__ src_mask = (const uint8_t*)(base + S.mask);
  const uint32_t dv = i / dw, du = i % dw;
  const float hr = (float)sh / (float)dh, wr = (float)sw / (float)dw;
  const uint32_t sv = (uint32_t)((float)dv * hr), su = (uint32_t)((float)du * wr);
  V3 cand[4];
  bool ok[4];
#pragma unroll
  for (uint32_t a = 0; a < 2; ++a)
#pragma unroll
    for (uint32_t b = 0; b < 2; ++b) {
      const uint32_t r = sv + a, c = su + b, q = a * 2 + b;
      const bool in = r < sh && c < sw;
      const uint32_t k = in ? r * sw + c : 0u;
      ok[q] = in && src_mask[k] == 1;
      cand[q] = ld_v3g(src, k);
    }
  int n;
  const V3 nearest = pick_nearest_to_mean(cand, ok, &n);
  st_v3u_stream(dst, i, nearest);  // (the coarsest levels: read by the alignment, not by the builder)
  if (!normals) st_u8u_stream(base + D.mask, i, n > 0 ? 1 : 0);
}

// One tap table entry per output row / column: first tap, tap count, normalised weights.
constexpr int MAX_TAPS = 14;  // ceil(in + 2 sigma) - floor(in - 2 sigma) <= 14  <=>  sigma <= 3
struct TapRow {
  int32_t left, count;
  float w[MAX_TAPS];
};
// First tap and tap count of output sample `o` (source coordinates) of image::imageops::blur's sampling filter: the same
// f32 operations on host (make_taps) and device, so a kernel can place its loads without waiting for the table.
__host__ __device__ __forceinline__ void tap_range(uint32_t o, float support, uint32_t size, int32_t* left, int32_t* count) {
  const float in = (float)o + 0.5f;
  int32_t l = (int32_t)floorf(in - support);
  l = l < 0 ? 0 : (l > (int32_t)size - 1 ? (int32_t)size - 1 : l);
  int32_t r = (int32_t)ceilf(in + support);
  r = r < l + 1 ? l + 1 : (r > (int32_t)size ? (int32_t)size : r);
  *left = l, *count = r - l < MAX_TAPS ? r - l : MAX_TAPS;
}

// imageops::blur + 2x subsample fused: the pyramid keeps only the even rows and columns of the blurred image,
// so the vertical pass is evaluated at even rows only and never leaves the chip.  One block = BLUR_ROWS output rows x
// BLUR_TILE output columns: the source bytes its taps touch are fetched once, as aligned 32-bit words, into LDS
// (consecutive output rows share all but two of their source rows); the vertical sums (u8 -> f32, the crate's f32
// intermediate) go to a second LDS array, then the horizontal pass, clamp and round-half-away (u8) read them back.
// Per output the additions run in tap order from 0.0f in both passes, as in the crate.
constexpr uint32_t BLUR_TILE = 64, BLUR_ROWS = 4;  // (2 / 4 / 8 rows per block measured 32 / 31 / 37 us per 16 frames)
constexpr uint32_t BLUR_SPAN = 3 * (2 * BLUR_TILE + MAX_TAPS + 2);   // bytes / vertical results a tile's row can need
constexpr uint32_t RAW_ROWS = 2 * (BLUR_ROWS - 1) + MAX_TAPS;        // source rows under BLUR_ROWS output rows
constexpr uint32_t RAW_PITCH = ((BLUR_SPAN + 3 + 3) / 4) * 4;        // bytes per staged source row (+ alignment slack)
__global__ void __launch_bounds__(256)
    blur_halve_kernel(size_t off_src, uint32_t w, uint32_t dw, uint32_t dh, const TapRow* __restrict__ taps_v,
                      const TapRow* __restrict__ taps_h, size_t off_dst, FrameBases bases) {
  __shared__ uint32_t s_raw[RAW_ROWS * RAW_PITCH / 4];
  __shared__ float s_v[BLUR_ROWS * BLUR_SPAN];
  __shared__ uint32_t s_shift[RAW_ROWS];
  const uint8_t* __restrict__ rgb = (const uint8_t*)(bases.arena[blockIdx.z] + off_src);  // blockIdx.z = frame
  uint8_t* __restrict__ out = (uint8_t*)(bases.arena[blockIdx.z] + off_dst);
  const uint32_t dy0 = blockIdx.y * BLUR_ROWS, rows = min(BLUR_ROWS, dh - dy0);
  const uint32_t dx0 = blockIdx.x * BLUR_TILE, dx1 = min(dx0 + BLUR_TILE, dw) - 1;
  // rows [vtop, vbot) and columns [cmin, cmax) of the source under this tile (tap tables: block-uniform reads)
  const int32_t vtop = taps_v[dy0].left, vbot = taps_v[dy0 + rows - 1].left + taps_v[dy0 + rows - 1].count;
  const int32_t cmin = taps_h[dx0].left, cmax = taps_h[dx1].left + taps_h[dx1].count;
  const uint32_t span = (uint32_t)(cmax - cmin) * 3u, nraw = (uint32_t)(vbot - vtop);
  // ---- the source bytes, whole aligned words per row (the arena's arrays are 256-byte aligned and padded, and the
  // colour array is never the arena's last, so the word holding a row's last byte may be read) --------------------
  const uint32_t words = (span + 3 + 3) / 4;  // a row starts 0..3 bytes into its first word
  for (uint32_t e = threadIdx.x; e < nraw * words; e += 256) {
    const uint32_t j = e / words, k = e % words;
    const size_t first = ((size_t)(vtop + (int32_t)j) * w + (size_t)cmin) * 3;
    s_raw[j * (RAW_PITCH / 4) + k] = *(const uint32_t*)(rgb + (first & ~(size_t)3) + 4 * (size_t)k);
  }
  if (threadIdx.x < nraw)  // where in its first word each staged row starts
    s_shift[threadIdx.x] = (uint32_t)((((size_t)(vtop + (int32_t)threadIdx.x) * w + (size_t)cmin) * 3) & 3);
  __syncthreads();
  // ---- vertical pass: rows x span sums (row and taps block-uniform, the bytes of a row across lanes) ----------
  const uint8_t* raw = (const uint8_t*)s_raw;
  for (uint32_t r = 0; r < rows; ++r) {
    const TapRow* tv = taps_v + dy0 + r;  // row 2 * (dy0 + r) of the source (table built with stride 2)
    const int32_t j0 = tv->left - vtop, vcount = tv->count;
    for (uint32_t x = threadIdx.x; x < span; x += 256) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < MAX_TAPS; ++k)
        if (k < vcount) {
          const uint32_t j = (uint32_t)(j0 + k);
          acc += (float)raw[j * RAW_PITCH + s_shift[j] + x] * tv->w[k];
        }
      s_v[r * BLUR_SPAN + x] = acc;
    }
  }
  __syncthreads();
  // ---- horizontal pass: a thread owns one (column, channel) of the tile for all its rows: taps read once ------
  const uint32_t o = threadIdx.x, dx = dx0 + o / 3, ch = o % 3;
  if (o >= BLUR_TILE * 3 || dx >= dw) return;
  const TapRow th = taps_h[dx];
  const uint32_t h0 = (uint32_t)(th.left - cmin) * 3u + ch;
  for (uint32_t r = 0; r < rows; ++r) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < MAX_TAPS; ++k)
      if (k < th.count) acc += s_v[r * BLUR_SPAN + h0 + 3u * (uint32_t)k] * th.w[k];
    acc = fminf(fmaxf(acc, 0.0f), 255.0f);
    *(uint8_t __attribute__((address_space(1)))*)((a3d_gptr)out + (((dy0 + r) * dw + dx) * 3u + ch)) = (uint8_t)roundf(acc);
  }
}

// The same kernel when a colour row is a multiple of four bytes (w * 3 % 4 == 0: every staged row then starts at the
// same offset `sh` inside its first word): the vertical pass works on whole 32-bit words — one LDS read, four
// v_cvt_f32_ubyteN, four multiply / add pairs per word and tap instead of a byte-wide LDS read per output and tap — and
// the staging loop has no integer division.  Same operations per output in the same order: same bits.
__global__ void __launch_bounds__(256)
    blur_halve_words_kernel(size_t off_src, uint32_t w, uint32_t h, uint32_t dw, uint32_t dh, float support,
                            const TapRow* __restrict__ taps_v, const TapRow* __restrict__ taps_h, size_t off_dst,
                            FrameBases bases) {
  constexpr uint32_t PITCH_W = RAW_PITCH / 4;                    // words per staged row
  __shared__ uint32_t s_raw[RAW_ROWS * PITCH_W];
  __shared__ __attribute__((aligned(16))) float s_v[BLUR_ROWS * PITCH_W * 4];  // vertical sums, indexed by RAW byte position
  __shared__ uint32_t s_tv[BLUR_ROWS][16];                       // the tile's rows of the vertical tap table
  static_assert(sizeof(TapRow) == 64, "a tap row is sixteen words");
  const uint8_t* __restrict__ rgb = (const uint8_t*)(bases.arena[blockIdx.z] + off_src);  // blockIdx.z = frame
  uint8_t* __restrict__ out = (uint8_t*)(bases.arena[blockIdx.z] + off_dst);
  const uint32_t dy0 = blockIdx.y * BLUR_ROWS, rows = min(BLUR_ROWS, dh - dy0);
  const uint32_t dx0 = blockIdx.x * BLUR_TILE, dx1 = min(dx0 + BLUR_TILE, dw) - 1;
  // The source rows [vtop, vbot) and columns [cmin, cmax) under this tile, from tap_range (what the tables hold): a block
  // then has ONE round of global loads — source words, its rows of the vertical table (to LDS) and each thread's row of
  // the horizontal table (to registers) are all in flight together — instead of table -> addresses -> source -> table.
  int32_t vtop, vbot, cmin, cmax, cnt;
  tap_range(2 * dy0, support, h, &vtop, &cnt);
  tap_range(2 * (dy0 + rows - 1), support, h, &vbot, &cnt), vbot += cnt;
  tap_range(2 * dx0, support, w, &cmin, &cnt);
  tap_range(2 * dx1, support, w, &cmax, &cnt), cmax += cnt;
  const uint32_t span = (uint32_t)(cmax - cmin) * 3u, nraw = (uint32_t)(vbot - vtop);
  const uint32_t sh = ((uint32_t)cmin * 3u) & 3u;               // the same for every row: w * 3 is a multiple of four
  const uint32_t words = (span + sh + 3) / 4;
  const uint32_t o = threadIdx.x, dx = dx0 + o / 3, ch = o % 3;
  const bool owns_output = o < BLUR_TILE * 3 && dx < dw;
  uint4 th4[4] = {};  // taps_h[dx]
  if (owns_output) {
    const uint4* src = (const uint4*)(taps_h + dx);
#pragma unroll
    for (int i = 0; i < 4; ++i) th4[i] = src[i];
  }
  const bool stages_taps = threadIdx.x < rows * 16u;
  const uint32_t tv_word = stages_taps ? ((const uint32_t*)(taps_v + dy0))[threadIdx.x] : 0u;  // (parked behind the source loads)
  // ---- staging: thread = (row slot t / 128, word t % 128); a tile's row has at most 108 words ----
  {
    const uint32_t k = threadIdx.x & 127u, jj = threadIdx.x >> 7;
    // (32-bit byte offsets off the frame's uniform colour pointer: an image is below 2^28 pixels)
    const uint32_t row_bytes = w * 3u;
    const uint32_t first = (((uint32_t)vtop * w + (uint32_t)cmin) * 3u & ~3u) + 4u * k;
    // (every load of the thread issued before the first is parked in LDS: a loop over j with a run-time trip count is
    // compiled into rounds of two loads, each round waiting for the previous one)
    uint32_t got[RAW_ROWS / 2];
#pragma unroll
    for (uint32_t i = 0; i < RAW_ROWS / 2; ++i) {
      const uint32_t j = jj + 2 * i;
      got[i] = (k < words && j < nraw) ? *(const uint32_t __attribute__((address_space(1)))*)((a3d_gptr_c)rgb + (first + j * row_bytes)) : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < RAW_ROWS / 2; ++i) {
      const uint32_t j = jj + 2 * i;
      if (k < words && j < nraw) s_raw[j * PITCH_W + k] = got[i];
    }
  }
  if (stages_taps) s_tv[threadIdx.x >> 4][threadIdx.x & 15u] = tv_word;
  __syncthreads();
  // ---- vertical pass, word-wise: item = (output row r, word q) ----
  for (uint32_t e = threadIdx.x; e < rows * 128u; e += 256) {
    const uint32_t r = e >> 7, q = e & 127u;
    if (q >= words) continue;
    // row 2 * (dy0 + r) of the source (table built with stride 2)
    // (r is the same for the 64 lanes of a wave — 128 items per row — so the tap count is a scalar: up to SHORT_TAPS
    // taps (sigma = 1 has five) run an unrolled body of that many predicated taps instead of fourteen)
    const int32_t j0 = (int32_t)s_tv[r][0] - vtop, vcount = __builtin_amdgcn_readfirstlane((int32_t)s_tv[r][1]);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    auto tap = [&](int k) {
      const uint32_t word = s_raw[(uint32_t)(j0 + k) * PITCH_W + q];
      const float wk = __uint_as_float(s_tv[r][2 + k]);
      a0 += (float)(word & 255u) * wk, a1 += (float)((word >> 8) & 255u) * wk;
      a2 += (float)((word >> 16) & 255u) * wk, a3 += (float)(word >> 24) * wk;
    };
    constexpr int SHORT_TAPS = 6;
    if (vcount <= SHORT_TAPS) {
#pragma unroll
      for (int k = 0; k < SHORT_TAPS; ++k)
        if (k < vcount) tap(k);
    } else {
#pragma unroll
      for (int k = 0; k < MAX_TAPS; ++k)
        if (k < vcount) tap(k);
    }
    *(float4*)(s_v + (r * PITCH_W + q) * 4) = make_float4(a0, a1, a2, a3);
  }
  __syncthreads();
  // ---- horizontal pass: a thread owns one (column, channel) of the tile for all its rows: taps read once ------
  if (!owns_output) return;
  const uint32_t tw[16] = {th4[0].x, th4[0].y, th4[0].z, th4[0].w, th4[1].x, th4[1].y, th4[1].z, th4[1].w,
                           th4[2].x, th4[2].y, th4[2].z, th4[2].w, th4[3].x, th4[3].y, th4[3].z, th4[3].w};
  const int32_t hcount = (int32_t)tw[1];
  const uint32_t h0 = (uint32_t)((int32_t)tw[0] - cmin) * 3u + ch + sh;
  // (the same short body when no lane of the wave has more than SHORT_TAPS taps)
  const bool short_taps = __builtin_amdgcn_ballot_w64(hcount > 6) == 0ull;
  for (uint32_t r = 0; r < rows; ++r) {
    float acc = 0.0f;
    if (short_taps) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (k < hcount) acc += s_v[r * PITCH_W * 4 + h0 + 3u * (uint32_t)k] * __uint_as_float(tw[2 + k]);
    } else {
#pragma unroll
      for (int k = 0; k < MAX_TAPS; ++k)
        if (k < hcount) acc += s_v[r * PITCH_W * 4 + h0 + 3u * (uint32_t)k] * __uint_as_float(tw[2 + k]);
    }
    acc = fminf(fmaxf(acc, 0.0f), 255.0f);
    *(uint8_t __attribute__((address_space(1)))*)((a3d_gptr)out + (((dy0 + r) * dw + dx) * 3u + ch)) = (uint8_t)roundf(acc);
  }
}

// Tap tables of image::imageops::blur's sampling filter (support 2 sigma, weights renormalised over the
// clamped range), computed on the host in f32 exactly as the oracle computes them.
std::vector<TapRow> make_taps(uint32_t size, float sigma, uint32_t stride, uint32_t count) {
  const float support = 2.0f * sigma;
  std::vector<TapRow> rows(count);
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t o = k * stride;
    const float in = (float)o + 0.5f;
    const float c = in - 0.5f;
    TapRow r{};
    tap_range(o, support, size, &r.left, &r.count);  // (callers reject sigma > 3: more than MAX_TAPS taps)
    const int64_t left = r.left;
    float sum = 0.0f, wv[MAX_TAPS];
    for (int i = 0; i < r.count; ++i) {
      const float x = (float)(left + i) - c;
      wv[i] = 1.0f / (std::sqrt(2.0f * 3.14159265358979323846f) * sigma) * std::exp(-(x * x) / (2.0f * sigma * sigma));
      sum += wv[i];
    }
    for (int i = 0; i < r.count; ++i) r.w[i] = wv[i] / sum;
    rows[k] = r;
  }
  return rows;
}

inline size_t padded(size_t bytes) { return ((std::max<size_t>(1, bytes) + 255) / 256) * 256; }

// Where each array of each level lives inside a frame's arena (256-byte aligned pieces, same for every frame).
struct ArenaPlan {
  PyramidLayout layout{};
  size_t bytes = 0;
};
ArenaPlan plan_arena(uint32_t w, uint32_t h, const a3d_builder_params* prm) {
  ArenaPlan p;
  auto take = [&](size_t bytes) {
    const size_t at = p.bytes;
    p.bytes += padded(bytes);
    return at;
  };
  for (uint64_t l = 0; l < prm->pyramid_levels; ++l) {
    LevelLayout& L = p.layout.lv[l];
    L.w = w >> l, L.h = h >> l;
    const size_t n = (size_t)L.w * L.h;
    L.colors = take(n * 3);
    L.points = take(n * 12);
    L.mask = take(n);
    L.normals = prm->with_normals ? take(n * 12) : 0;
    L.intensities = prm->with_intensity ? take(n) : 0;
    L.imap = prm->with_intensity ? take((size_t)(L.w + 2) * (L.h + 2) * 4) : 0;
  }
  return p;
}

// The tap tables depend on (size, sigma) only: computed and uploaded once per context, then reused.
a3d_status taps_for(a3d_context* ctx, uint32_t size, uint32_t count, float sigma, TapRow** out) {
  uint32_t key[4] = {0x54415053u /* 'TAPS' */, size, count, 0};
  memcpy(&key[3], &sigma, 4);
  for (const auto& t : ctx->tables)
    if (!memcmp(t.key, key, sizeof(key))) {
      *out = (TapRow*)t.d;
      return A3D_OK;
    }
  const std::vector<TapRow> rows = make_taps(size, sigma, 2, count);
  return ctx_cached_table(ctx, key, rows.data(), rows.size() * sizeof(TapRow), (void**)out);
}

// Everything after the uploads for up to MAX_BATCH frames whose depth images sit at d_depth ([F][h][w]) and whose
// colours are already in their arenas.  Enqueue only; the caller synchronises and then reads `result`.
// The work has two independent halves that meet in nobody's input: the DEPTH half (bilateral grid, level 0, picked
// levels: `depth_part`) and the COLOUR half (blurred + halved colours, intensities and their maps: `color_part`); a
// chunk whose grids outgrew the scratch region repeats the depth half only.  (Running the colour half on a second
// stream under the depth half was measured in round 4 (colour kernels enqueued last: 15.3 against 15.4 us per frame) and
// again in round 6 (enqueued FIRST, to fill the ~55 us in which min / max, grid sizing, splat and unsplat leave the chip all
// but idle: 11.6-12.2 against 11.8 us per frame resident, and streaming 17.1 -> 14.4-14.9 k pairs/s next to an aligning
// context) — dropped both times.)
a3d_status enqueue_chunk(a3d_context* ctx, const a3d_builder_params* prm, uint32_t F, const uint16_t* d_depth, uint32_t w,
                         uint32_t h, float fx, float fy, float cx, float cy, float depth_scale, const ArenaPlan& plan,
                         const FrameBases& bases, uint32_t* result, bool depth_part, bool color_part) {
  hipStream_t s = ctx->stream;
  hipStream_t color_stream = s;
  const PyramidLayout& P = plan.layout;
  const LevelLayout& L0 = P.lv[0];
  // level 1's points / mask / normals come out of the level-0 kernel when the sides are even (2 x 2 blocks are whole and the
  // resize's float index arithmetic is exactly 2 dv, 2 du), level 2's as well when they are multiples of four;
  // A3D_BUILDER_FUSE_L1=0 / A3D_BUILDER_FUSE_L2=0 keep the separate kernel (cross-checks, diagnostics build, read per call)
  const bool fuse_allowed = !(A3D_DIAG_ENV("A3D_BUILDER_FUSE_L1") && atoi(A3D_DIAG_ENV("A3D_BUILDER_FUSE_L1")) == 0);
  const bool fuse2_allowed = !(A3D_DIAG_ENV("A3D_BUILDER_FUSE_L2") && atoi(A3D_DIAG_ENV("A3D_BUILDER_FUSE_L2")) == 0);
  const bool fuse_l1 = fuse_allowed && prm->pyramid_levels >= 2 && w % 2 == 0 && h % 2 == 0;
  bool fuse_l2 = fuse_l1 && fuse2_allowed && prm->pyramid_levels >= 3 && w % 4 == 0 && h % 4 == 0;
  bool quad = true;
#ifdef A3D_DIAGNOSTICS
  const bool patch_kernel = A3D_DIAG_ENV("A3D_BUILDER_L0") && !strcmp(A3D_DIAG_ENV("A3D_BUILDER_L0"), "patch");
  quad = !patch_kernel;
  if (!quad) fuse_l2 = false;
#endif
  const LevelLayout& L2 = P.lv[prm->pyramid_levels >= 3 ? 2 : 0];
  const bool l2_is_last = prm->pyramid_levels == 3;
  const dim3 gridq((w + QS - 1) / QS, (h + QS - 1) / QS, F);
  if (!depth_part) {
  } else if (prm->use_bilateral) {  // builder.rs:75-77
    GridBatch gb;
    // (diagnostics build, A3D_BUILDER_UNSPLAT=fused: the level-0 kernel puts the filter's zeros back instead of unsplat_kernel —
    // measured round 6: level 0 140 instead of 123 us per 32 frames for a 10 us launch saved; kept as a cross-check)
    const bool defer = quad && A3D_DIAG_ENV("A3D_BUILDER_UNSPLAT") && !strcmp(A3D_DIAG_ENV("A3D_BUILDER_UNSPLAT"), "fused");
    A3D_TRY(bilateral_grids_enqueue(ctx, d_depth, F, w, h, prm->sigma_space, prm->sigma_color, ctx->grid_capacity, &gb, defer));
    if (quad) {
      hipLaunchKernelGGL(level0_quad_kernel<true>, dim3(gridq.x * gridq.y * gridq.z), dim3(256), 0, s, d_depth, w, h, 1.0 / prm->sigma_space,
                         1.0 / prm->sigma_color, (const double*)gb.blurred, gb.capacity, gb.scal, fx, fy, cx, cy, depth_scale,
                         bases, L0.points, L0.mask, L0.normals, prm->with_normals != 0, P.lv[1], fuse_l1, L2, fuse_l2, l2_is_last,
                         gb.unsplat, gridq.x, gridq.y);
      if (defer) {
        A3D_HIP_TRY(hipGetLastError());
        ctx->grid_clean = gb.clean;  // (the zeros are back once this kernel has run)
      }
    }
#ifdef A3D_DIAGNOSTICS
    else
      hipLaunchKernelGGL(level0_kernel<true>, dim3((w + OWN_W - 1) / OWN_W, (h + OWN_H - 1) / OWN_H, F), dim3(L0_THREADS), 0, s,
                         d_depth, w, h, 1.0 / prm->sigma_space, 1.0 / prm->sigma_color, (const double*)gb.blurred, gb.capacity,
                         gb.scal, fx, fy, cx, cy, depth_scale, bases, L0.points, L0.mask, L0.normals, prm->with_normals != 0,
                         P.lv[1], fuse_l1);
#endif
    A3D_HIP_TRY(hipMemcpyAsync(result, gb.scal, (size_t)F * SC_STRIDE * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  } else {
    if (quad)
      hipLaunchKernelGGL(level0_quad_kernel<false>, dim3(gridq.x * gridq.y * gridq.z), dim3(256), 0, s, d_depth, w, h, 0.0, 0.0,
                         (const double*)nullptr, 0ull, (uint32_t*)nullptr, fx, fy, cx, cy, depth_scale, bases, L0.points, L0.mask,
                         L0.normals, prm->with_normals != 0, P.lv[1], fuse_l1, L2, fuse_l2, l2_is_last, Unsplat{}, gridq.x, gridq.y);
#ifdef A3D_DIAGNOSTICS
    else
      hipLaunchKernelGGL(level0_kernel<false>, dim3((w + OWN_W - 1) / OWN_W, (h + OWN_H - 1) / OWN_H, F), dim3(L0_THREADS), 0, s,
                         d_depth, w, h, 0.0, 0.0, (const double*)nullptr, 0ull, (uint32_t*)nullptr, fx, fy, cx, cy, depth_scale,
                         bases, L0.points, L0.mask, L0.normals, prm->with_normals != 0, P.lv[1], fuse_l1);
#endif
  }
  // RangeImage::pyramid (structure.rs:342-351): normals exist at level 0 only (builder.rs:79-82), coarser levels
  // inherit picked normals; colours are blurred and halved level by level
  float sigma = prm->blur_sigma;
  if (sigma <= 0.0f) sigma = 1.0f;
  for (uint64_t l = 1; l < prm->pyramid_levels; ++l) {
    const LevelLayout &S = P.lv[l - 1], &D = P.lv[l];
    if (depth_part && !(l == 1 && fuse_l1) && !(l == 2 && fuse_l2))
      hipLaunchKernelGGL(resize_pick_kernel, dim3((D.w * D.h + 255) / 256, prm->with_normals ? 2 : 1, F), dim3(256), 0, s, S,
                         D, bases);
    if (!color_part) continue;
    TapRow *d_tv = nullptr, *d_th = nullptr;
    A3D_TRY(taps_for(ctx, S.h, D.h, sigma, &d_tv));
    A3D_TRY(taps_for(ctx, S.w, D.w, sigma, &d_th));
    static const bool blur_words = !(A3D_DIAG_ENV("A3D_BUILDER_BLUR") && !strcmp(A3D_DIAG_ENV("A3D_BUILDER_BLUR"), "bytes"));  // cross-check knob
    const dim3 blur_grid((D.w + BLUR_TILE - 1) / BLUR_TILE, (D.h + BLUR_ROWS - 1) / BLUR_ROWS, F);
    if (blur_words && (S.w * 3) % 4 == 0 && (RAW_PITCH / 4) <= 128)
      hipLaunchKernelGGL(blur_halve_words_kernel, blur_grid, dim3(256), 0, color_stream, S.colors, S.w, S.h, D.w, D.h, 2.0f * sigma, d_tv,
                         d_th, D.colors, bases);
    else
      hipLaunchKernelGGL(blur_halve_kernel, blur_grid, dim3(256), 0, color_stream, S.colors, S.w, D.w, D.h, d_tv, d_th, D.colors, bases);
  }
  if (prm->with_intensity && color_part) {
    bool quads = true;  // every level's width a multiple of four (the colour, intensity and map rows then stay word-aligned)
    for (uint64_t l = 0; l < prm->pyramid_levels; ++l) quads = quads && P.lv[l].w % 4 == 0;
    if (quads)
      hipLaunchKernelGGL(luma_imap_kernel<true>, dim3((h * (w / 4) + 2 * (w + 2) + 2 * h + 255) / 256, (uint32_t)prm->pyramid_levels, F),
                         dim3(256), 0, color_stream, P, bases);
    else
      hipLaunchKernelGGL(luma_imap_kernel<false>, dim3(((w + 2) * (h + 2) + 255) / 256, (uint32_t)prm->pyramid_levels, F),
                         dim3(256), 0, color_stream, P, bases);
  }
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

// One chunk of a batched build: up to MAX_BATCH frames that share a launch sequence.
struct Chunk {
  uint32_t F = 0;
  FrameBases bases{};
  uint16_t* d_depth = nullptr;            // [F][h][w] in the context's staging region
  uint8_t* d_colors = nullptr;            // [F][h][w][3] in the staging region (used when the host frames are contiguous)
  bool colors_staged = false;
  std::vector<a3d_device_image*> images;  // [F][L]
  uint32_t* result = nullptr;             // this chunk's page-locked scalar blocks
};

// Arenas and image handles of a chunk; its uploads go to the copy stream (they only touch memory no kernel of an
// earlier chunk reads), followed by an event the compute stream waits for.
a3d_status chunk_prepare(a3d_context* ctx, const a3d_builder_params* prm, const ArenaPlan& plan, Chunk& c,
                         const uint16_t* const* depth, const uint8_t* const* rgb, uint32_t w, uint32_t h, double fx,
                         double fy, double cx, double cy, hipEvent_t uploaded, bool mask_is_z) {
  const size_t n = (size_t)w * h;
  const uint64_t L = prm->pyramid_levels;
  for (uint32_t f = 0; f < c.F; ++f) {
    DeviceArena* arena = new DeviceArena();
    if (ctx_arena_acquire(ctx, plan.bytes, arena) != A3D_OK) {
      delete arena;
      set_error("a3d_range_image_build_pyramids: hipMalloc(%zu) failed", plan.bytes);
      return A3D_HIP_ERROR;
    }
    c.bases.arena[f] = (char*)arena->base;
    for (uint64_t l = 0; l < L; ++l) {
      const LevelLayout& Y = plan.layout.lv[l];
      a3d_device_image* im = new a3d_device_image();
      im->ctx = ctx, im->arena = arena, im->built = true;
      // mask = (depth > 0) and z = (float)depth * scale: the two agree when the smallest depth unit already gives a
      // non-zero z (a positive, normal f32 scale); picked pyramid levels inherit it (an invalid pick is (0, 0, 0))
      im->mask_is_z = mask_is_z;
      ++arena->refs;
      im->width = Y.w, im->height = Y.h;
      const double k = std::ldexp(1.0, -(int)l);  // CameraIntrinsics::scale(0.5) per level (camera.rs:119-127): exact
      im->fx64 = fx * k, im->fy64 = fy * k, im->cx64 = cx * k, im->cy64 = cy * k;
      im->fx = (float)im->fx64, im->fy = (float)im->fy64, im->cx = (float)im->cx64, im->cy = (float)im->cy64;
      char* b = (char*)arena->base;
      im->colors = (uint8_t*)(b + Y.colors), im->points = (float*)(b + Y.points), im->mask = (uint8_t*)(b + Y.mask);
      if (prm->with_normals) im->normals = (float*)(b + Y.normals), im->has_normals = true;
      if (prm->with_intensity) {
        im->intensities = (uint8_t*)(b + Y.intensities), im->imap = (float*)(b + Y.imap);
        im->has_intensities = im->has_imap = true;
      }
      c.images.push_back(im);
    }
  }
  // kernels index the chunk's depth images as [F][n]: n * 2 bytes apart (no padding between frames).
  // Page-locked host frames that lie back to back (one buffer for the stream) go up as ONE copy per array: a copy of
  // 0.6-0.9 MB pays ~15 us of fixed cost on top of its bytes (23 GB/s observed frame by frame, ~50 GB/s for 10 MB).
  // Only for page-locked memory: a large copy from pageable memory goes through the runtime's staging path at a few
  // GB/s (measured 1.2 ms per frame against 0.08 frame by frame).
  auto page_locked = [](const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
      (void)hipGetLastError();  // ordinary pageable memory is "invalid value" to this query
      return false;
    }
    return a.type == hipMemoryTypeHost;
  };
  bool depth_contig = page_locked(depth[0]), rgb_contig = (n * 3) % 16 == 0 && page_locked(rgb[0]);
  for (uint32_t f = 1; f < c.F; ++f) {
    depth_contig &= depth[f] == depth[f - 1] + n;
    rgb_contig &= rgb[f] == rgb[f - 1] + n * 3;
  }
  hipError_t e = hipSuccess;
  if (depth_contig) {
    e = hipMemcpyAsync(c.d_depth, depth[0], (size_t)c.F * n * 2, hipMemcpyHostToDevice, ctx->copy_stream);
  } else {
    for (uint32_t f = 0; f < c.F && e == hipSuccess; ++f)
      e = hipMemcpyAsync(c.d_depth + (size_t)f * n, depth[f], n * 2, hipMemcpyHostToDevice, ctx->copy_stream);
  }
  c.colors_staged = rgb_contig && c.F > 1;
  if (e == hipSuccess && c.colors_staged) {
    e = hipMemcpyAsync(c.d_colors, rgb[0], (size_t)c.F * n * 3, hipMemcpyHostToDevice, ctx->copy_stream);
  } else {
    for (uint32_t f = 0; f < c.F && e == hipSuccess; ++f)
      e = hipMemcpyAsync(c.bases.arena[f] + plan.layout.lv[0].colors, rgb[f], n * 3, hipMemcpyHostToDevice, ctx->copy_stream);
  }
  if (e != hipSuccess) {
    set_error("a3d_range_image_build_pyramids: upload failed: %s", hipGetErrorString(e));
    return A3D_HIP_ERROR;
  }
  A3D_HIP_TRY(hipEventRecord(uploaded, ctx->copy_stream));
  return A3D_OK;
}

// Up to PINNED_WORDS / (MAX_BATCH * SC_STRIDE) chunks in one pipelined pass: all uploads are queued on the copy
// stream first, the chunks' kernels follow on the context stream as their uploads land, ONE synchronisation at the end.
a3d_status build_frames(a3d_context* ctx, const a3d_builder_params* prm, uint64_t n_frames, uint64_t chunk_frames,
                        const uint16_t* const* depth, const uint8_t* const* rgb, uint32_t w, uint32_t h, double fx,
                        double fy, double cx, double cy, double depth_scale, a3d_device_image** out_levels) {
  hipStream_t s = ctx->stream;
  const size_t n = (size_t)w * h;
  const ArenaPlan plan = plan_arena(w, h, prm);
  const size_t n_chunks = (size_t)((n_frames + chunk_frames - 1) / chunk_frames);
  std::vector<Chunk> chunks(n_chunks);
  auto fail = [&](a3d_status st) {
    hipStreamSynchronize(ctx->copy_stream);
    hipStreamSynchronize(s);
    for (Chunk& c : chunks)
      for (a3d_device_image* im : c.images) a3d_range_image_free(im);  // a frame's last level releases its arena
    return st;
  };
  void* staging = nullptr;
  const size_t depth_bytes = padded((size_t)n_frames * n * 2);
  if (ctx_scratch(ctx, 0, depth_bytes + (size_t)n_frames * n * 3 + 256, &staging) != A3D_OK) return A3D_HIP_ERROR;
  while (ctx->copy_events.size() < n_chunks) {
    hipEvent_t e;
    A3D_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->copy_events.push_back(e);
  }
  if (prm->use_bilateral && ctx->grid_capacity == 0)  // first guess: a depth span of 4096 units (4 m at 1 mm)
    ctx->grid_capacity = bilateral_grid_cells(w, h, prm->sigma_space, prm->sigma_color, 4096);
  // the staging region may have moved (ctx_scratch synchronised the context stream then); nothing older is in flight
  for (size_t k = 0; k < n_chunks; ++k) {
    Chunk& c = chunks[k];
    const uint64_t f0 = k * chunk_frames;
    c.F = (uint32_t)std::min<uint64_t>(chunk_frames, n_frames - f0);
    c.d_depth = (uint16_t*)staging + f0 * n;
    c.d_colors = (uint8_t*)staging + depth_bytes + f0 * n * 3;
    c.result = ctx->pinned_words + k * (MAX_BATCH * SC_STRIDE);
    const float fscale = (float)depth_scale;
    const bool mask_is_z = fscale >= 1.1754944e-38f && std::isfinite(fscale);  // 1.0f * scale != 0, and so is d * scale
    const a3d_status st = chunk_prepare(ctx, prm, plan, c, depth + f0, rgb + f0, w, h, fx, fy, cx, cy, ctx->copy_events[k],
                                        mask_is_z);
    if (st != A3D_OK) return fail(st);
  }
  std::vector<size_t> todo(n_chunks);
  for (size_t k = 0; k < n_chunks; ++k) todo[k] = k;
  size_t n_profiled = 0;
  for (int attempt = 0; attempt < 3 && !todo.empty(); ++attempt) {
    for (size_t k : todo) {
      Chunk& c = chunks[k];
      if (attempt == 0 && hipStreamWaitEvent(s, ctx->copy_events[k], 0) != hipSuccess) return fail(A3D_HIP_ERROR);
      if (ctx->build_profiling) {
        while (ctx->build_events.size() < 2 * (n_profiled + 1)) {
          hipEvent_t e;
          if (hipEventCreate(&e) != hipSuccess) return fail(A3D_HIP_ERROR);
          ctx->build_events.push_back(e);
        }
        (void)hipEventRecord(ctx->build_events[2 * n_profiled], s);
      }
      if (attempt == 0 && c.colors_staged)
        hipLaunchKernelGGL(scatter_colors_kernel, dim3((uint32_t)((n * 3 / 16 + 255) / 256), 1, c.F), dim3(256), 0, s,
                           (const uint4*)c.d_colors, n * 3, plan.layout.lv[0].colors, c.bases);
      // (the colour half runs once: a repeated attempt only redoes the grids)
      const a3d_status st = enqueue_chunk(ctx, prm, c.F, c.d_depth, w, h, (float)fx, (float)fy, (float)cx, (float)cy,
                                          (float)depth_scale, plan, c.bases, c.result, true, attempt == 0);
      if (st != A3D_OK) return fail(st);
      if (ctx->build_profiling) (void)hipEventRecord(ctx->build_events[2 * n_profiled++ + 1], s);
    }
    if (hipStreamSynchronize(s) != hipSuccess) {
      set_error("a3d_range_image_build_pyramids: %s", hipGetErrorString(hipGetLastError()));
      return fail(A3D_HIP_ERROR);
    }
    if (!prm->use_bilateral) {
      todo.clear();
      break;
    }
    unsigned long long need = 0;
    bool overflow = false;
    std::vector<size_t> again;
    for (size_t k : todo) {  // the filter's scalars arrived with the synchronisation above
      bool redo = false;
      for (uint32_t f = 0; f < chunks[k].F; ++f) {
        const uint32_t* r = chunks[k].result + f * SC_STRIDE;
        if (r[SC_TOO_BIG]) need = std::max<unsigned long long>(need, (unsigned long long)r[SC_GH] * r[SC_GW] * r[SC_GD]), redo = true;
        overflow |= r[SC_OVERFLOW] != 0 && !r[SC_TOO_BIG];
      }
      if (redo) again.push_back(k);
    }
    if (overflow) {
      set_error("bilateral slice produced a value outside u16 (the reference panics in num::cast().unwrap())");
      return fail(A3D_CAST_OVERFLOW);
    }
    if (need >= (1ull << 29)) {  // (dims_table_kernel refuses such grids: the slice addresses cells with 32-bit byte offsets)
      set_error("a3d_range_image_build_pyramids: a frame's bilateral grid would have %llu cells (the device builder handles "
                "grids below 2^29 cells: raise sigma_color or sigma_space)", need);
      return fail(A3D_INVALID_PARAMETER);
    }
    // a frame's bilateral grid outgrew the scratch region: grow it (25 % head room) and run those chunks again (their
    // inputs are still resident)
    if (need) ctx->grid_capacity = need + need / 4;
    todo.swap(again);
  }
  if (!todo.empty()) {
    set_error("a3d_range_image_build_pyramids: the bilateral grid kept outgrowing its scratch region");
    return fail(A3D_HIP_ERROR);
  }
  for (size_t k = 0; k < n_profiled; ++k) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->build_events[2 * k], ctx->build_events[2 * k + 1]) == hipSuccess) ctx->last_build_kernel_ms += ms;
  }
  size_t o = 0;
  for (Chunk& c : chunks) {
    for (a3d_device_image* im : c.images) out_levels[o++] = im;
    ctx->build_stats[0] += c.F;
    if (prm->use_bilateral)
      for (uint32_t f = 0; f < c.F; ++f) {
        const uint32_t* r = c.result + f * SC_STRIDE;
        ctx->build_stats[1] += (uint64_t)r[SC_GH] * r[SC_GW] * r[SC_GD];
        ctx->build_stats[2] += r[SC_NLIST], ctx->build_stats[3] += r[SC_NZERO];
      }
  }
  return A3D_OK;
}

}  // namespace

extern "C" {

// RangeImageBuilder::default() (builder.rs:16-26)
void a3d_builder_params_default(a3d_builder_params* out) {
  out->with_normals = 1;
  out->with_intensity = 1;
  out->use_bilateral = 0;
  a3d_bilateral_default_sigmas(&out->sigma_space, &out->sigma_color);
  out->pyramid_levels = 3;
  out->blur_sigma = 1.0f;
}

a3d_status a3d_range_image_build_pyramids(a3d_context* ctx, const a3d_builder_params* prm, uint64_t n_frames,
                                          const uint16_t* const* depth_frames, const uint8_t* const* rgb_frames,
                                          uint64_t width, uint64_t height, double fx, double fy, double cx, double cy,
                                          double depth_scale, a3d_device_image** out_levels) {
  A3D_REQUIRE(ctx && prm && depth_frames && rgb_frames && out_levels, A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(n_frames >= 1 && n_frames <= (1u << 20), A3D_INVALID_PARAMETER, "bad frame count");
  for (uint64_t f = 0; f < n_frames; ++f)
    A3D_REQUIRE(depth_frames[f] && rgb_frames[f], A3D_INVALID_PARAMETER, "null frame pointer");
  A3D_REQUIRE(width > 0 && height > 0 && width * height < (1ull << 28), A3D_INVALID_PARAMETER, "bad image size");
  // the kernels form texel offsets with 24-bit multiplies
  A3D_REQUIRE(width < (1ull << 23) && height < (1ull << 23), A3D_INVALID_PARAMETER, "image side too long");
  A3D_REQUIRE(prm->pyramid_levels >= 1 && prm->pyramid_levels <= MAX_LEVELS, A3D_INVALID_PARAMETER, "bad pyramid_levels");
  A3D_REQUIRE((width >> (prm->pyramid_levels - 1)) >= 2 && (height >> (prm->pyramid_levels - 1)) >= 2,
              A3D_INVALID_PARAMETER, "image too small for this many pyramid levels");
  // BilateralFilter::new takes any sigma; a non-positive or non-finite one would size the grid by a division by zero
  A3D_REQUIRE(!prm->use_bilateral || (prm->sigma_space > 0.0 && prm->sigma_color > 0.0 && std::isfinite(prm->sigma_space) &&
                                      std::isfinite(prm->sigma_color)),
              A3D_INVALID_PARAMETER, "bilateral sigmas must be positive and finite");
  A3D_REQUIRE(std::isfinite(prm->blur_sigma), A3D_INVALID_PARAMETER, "blur_sigma must be finite");
  A3D_REQUIRE(prm->pyramid_levels == 1 || prm->blur_sigma <= 3.0f, A3D_INVALID_PARAMETER,
              "blur_sigma above 3 is not supported by the device builder");
  A3D_REQUIRE(!prm->use_bilateral || width * height < (1ull << 24), A3D_INVALID_PARAMETER,
              "the device frame builder's bilateral filter handles images below 2^24 pixels");
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  const uint64_t L = prm->pyramid_levels;
  // frames per launch sequence: at most MAX_BATCH, and few enough that their bilateral grids fit A3D_GRID_BUDGET_GB of scratch
  uint64_t chunk = MAX_BATCH;
  if (prm->use_bilateral) {
    const unsigned long long cap = ctx->grid_capacity ? ctx->grid_capacity
                                                      : bilateral_grid_cells((uint32_t)width, (uint32_t)height, prm->sigma_space,
                                                                             prm->sigma_color, 4096);
    chunk = std::max<uint64_t>(1, std::min<uint64_t>(MAX_BATCH, ((unsigned long long)A3D_GRID_BUDGET_GB << 30) / (cap * 24 + 1)));
  }
  for (uint64_t& v : ctx->build_stats) v = 0;
  ctx->last_build_kernel_ms = 0.f;
  // frames per pipelined pass: as many chunks as the page-locked result area has scalar blocks for
  const uint64_t pass = chunk * (a3d_context::PINNED_WORDS / (MAX_BATCH * SC_STRIDE));
  for (uint64_t f0 = 0; f0 < n_frames; f0 += pass) {
    const uint64_t F = std::min<uint64_t>(pass, n_frames - f0);
    const uint64_t pieces = (F + chunk - 1) / chunk, balanced = (F + pieces - 1) / pieces;  // equal chunks, none tiny
    const a3d_status st = build_frames(ctx, prm, F, balanced, depth_frames + f0, rgb_frames + f0, (uint32_t)width, (uint32_t)height,
                                       fx, fy, cx, cy, depth_scale, out_levels + f0 * L);
    if (st != A3D_OK) {  // the caller gets all the pyramids or none
      for (uint64_t k = 0; k < f0 * L; ++k) a3d_range_image_free(out_levels[k]);
      return st;
    }
  }
  return A3D_OK;
}

a3d_status a3d_range_image_build_pyramid(a3d_context* ctx, const a3d_builder_params* prm, const uint16_t* depth,
                                         const uint8_t* rgb, uint64_t width, uint64_t height, double fx, double fy,
                                         double cx, double cy, double depth_scale, a3d_device_image** out_levels) {
  return a3d_range_image_build_pyramids(ctx, prm, 1, &depth, &rgb, width, height, fx, fy, cx, cy, depth_scale, out_levels);
}

// Instrumentation: what the most recent a3d_range_image_build_pyramids call on this context processed.
a3d_status a3d_context_last_build_stats(a3d_context* ctx, uint64_t out_stats[4]) {
  A3D_REQUIRE(ctx && out_stats, A3D_INVALID_PARAMETER, "null argument");
  for (int k = 0; k < 4; ++k) out_stats[k] = ctx->build_stats[k];
  return A3D_OK;
}

// Instrumentation: device time of the builder's kernels in the most recent build (see a3d_context::build_profiling).
a3d_status a3d_context_set_build_profiling(a3d_context* ctx, int32_t on) {
  A3D_REQUIRE(ctx, A3D_INVALID_PARAMETER, "ctx is null");
  ctx->build_profiling = on != 0;
  return A3D_OK;
}
a3d_status a3d_context_last_build_kernel_ms(a3d_context* ctx, float* out_ms) {
  A3D_REQUIRE(ctx && out_ms, A3D_INVALID_PARAMETER, "null argument");
  *out_ms = ctx->last_build_kernel_ms;
  return A3D_OK;
}

a3d_status a3d_range_image_size(const a3d_device_image* im, uint64_t* out_width, uint64_t* out_height) {
  A3D_REQUIRE(im && out_width && out_height, A3D_INVALID_PARAMETER, "null argument");
  *out_width = im->width, *out_height = im->height;
  return A3D_OK;
}

a3d_status a3d_range_image_download(a3d_device_image* im, float* points, uint8_t* mask, float* normals,
                                    uint8_t* intensities, float* intensity_map, uint8_t* colors, double out_intrinsics[4]) {
  A3D_REQUIRE(im, A3D_INVALID_PARAMETER, "image is null");
  hipStream_t s = im->ctx->stream;
  const size_t n = (size_t)im->width * im->height;
  auto get = [&](void* dst, const void* src, size_t bytes, const char* what) -> a3d_status {
    if (!dst) return A3D_OK;
    A3D_REQUIRE(src, A3D_MISSING_FIELD, what);
    A3D_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    return A3D_OK;
  };
  A3D_TRY(get(points, im->points, n * 12, "image has no points"));
  A3D_TRY(get(mask, im->mask, n, "image has no mask"));
  A3D_TRY(get(normals, im->has_normals ? im->normals : nullptr, n * 12, "image has no normals"));
  A3D_TRY(get(intensities, im->has_intensities ? im->intensities : nullptr, n, "image has no intensities"));
  A3D_TRY(get(intensity_map, im->has_imap ? im->imap : nullptr, (size_t)(im->width + 2) * (im->height + 2) * 4,
              "image has no intensity map"));
  A3D_TRY(get(colors, im->colors, n * 3, "image has no colors"));
  if (out_intrinsics) out_intrinsics[0] = im->fx64, out_intrinsics[1] = im->fy64, out_intrinsics[2] = im->cx64, out_intrinsics[3] = im->cy64;
  A3D_HIP_TRY(hipStreamSynchronize(s));
  return A3D_OK;
}

}  // extern "C"
