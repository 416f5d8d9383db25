// ImageIcp::align (src/icp/image_icp.rs:43-164) and MultiscaleAlign (src/icp/multiscale.rs:26-67)
// for P independent frame pairs at once: per iteration one per-pixel kernel (grid = tiles x pairs)
// and one solve kernel (grid = pairs); the whole coarse-to-fine sequence is enqueued without a host
// round trip.
#include <cstdlib>
#include <memory>

#include "icp_engine.hpp"

using namespace a3d;

namespace {

// What the per-pixel kernel needs to know about one (pair, level): resident arrays + intrinsics.
struct LevelDesc {
  const float4* src;  // [src_n] {x, y, z, intensity | -1}
  const float4* tgt;  // [th*tw][2] {x, y, z, valid}, {nx, ny, nz, 0}
  const float* imap;  // [(th+2)][(tw+2)]
  // the same data in the reference's own layout (RAW kernels): 14 B per source pixel, 25 B per target pixel
  const float* src_points;         // [src_n][3]
  const uint8_t* src_mask;         // [src_n]
  const uint8_t* src_intensities;  // [src_n]
  const float* tgt_points;         // [th*tw][3]
  const float* tgt_normals;        // [th*tw][3]
  const uint8_t* tgt_mask;         // [th*tw]
  uint32_t src_n;
  uint32_t tw, th;
  float fx, fy, cx, cy;
  uint32_t pad;
};

struct Gates {
  float max_distance_sqr;
  float max_color_distance_sqr;
  float dot_reject_max;  // reject iff -1 <= p.n <= dot_reject_max  (== acos(p.n).abs() >= max_normal_angle)
};

// `u as usize` (Rust): NaN and negatives -> 0; the callers only see u < width.
__device__ __forceinline__ uint32_t f32_as_usize(float x) { return x > 0.0f ? (uint32_t)x : 0u; }

// Pointers read out of a descriptor are generic; the arrays live in global memory, and saying so
// gives global_load instead of flat_load (one counter to wait on, free scheduling).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gptr_f4;
typedef const float __attribute__((address_space(1)))* gptr_f;
typedef const uint8_t __attribute__((address_space(1)))* gptr_u8;
// One 12-byte Vector3<f32> as ONE global_load_dwordx3.  Three separate dword loads at a 12-byte lane
// stride make the L1 look up every 64-byte line of the wave's 768-byte span three times; measured, that
// tag traffic (5.3 line lookups per pixel) was what bounded the kernel, not HBM.
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef f32x3 __attribute__((aligned(4))) f32x3_u;
typedef const f32x3_u __attribute__((address_space(1)))* gptr_f3;
__device__ __forceinline__ f32x3 load3(gptr_f base, uint32_t idx) { return *(gptr_f3)(base + 3 * idx); }

// IntensityMap::bilinear (src/intensity_map.rs:150-169) from four already-loaded texels.
__device__ __forceinline__ float bilerp(float v00, float v10, float v01, float v11, float uf, float vf) {
  float u0 = v00 * (1.0f - uf) + v10 * uf;
  float u1 = v01 * (1.0f - uf) + v11 * uf;
  return u0 * (1.0f - vf) + u1 * vf;
}

__device__ __forceinline__ float bilinear_at(gptr_f imap, uint32_t mw, float u, float v) {
  uint32_t ui = f32_as_usize(u), vi = f32_as_usize(v);
  gptr_f r0 = imap + (size_t)vi * mw + ui;
  return bilerp(r0[0], r0[1], r0[mw], r0[mw + 1], u - (float)ui, v - (float)vi);
}

// The reference's pixel loop (image_icp.rs:101-139).  grid = (tiles, pairs); block = 256.  A thread
// visits PPT source pixels, 256 apart (coalesced), G at a time: the G source records are loaded
// together, then the G projective gathers of the target record, then the G intensity-map cells, so
// that each dependent memory round trip is paid once per G pixels; only the accumulation is under
// the per-pixel gates.
template <int PPT, int G, bool RAW>
__global__ void __launch_bounds__(256)
    image_icp_kernel(const LevelDesc* __restrict__ descs, JobState* __restrict__ states, Gates gt,
                     float* __restrict__ partials, unsigned* __restrict__ counters, SolveArgs solve) {
  static_assert(PPT % G == 0, "PPT must be a multiple of G");
  const int pair = blockIdx.y;
  float acc[GN_PARTIAL];
#pragma unroll
  for (int k = 0; k < GN_PARTIAL; ++k) acc[k] = 0.0f;
  JobState* st = &states[pair];
  if (st->status == A3D_OK) {
    const LevelDesc d = descs[pair];
    const Pose T = st->pose;
    const gptr_f4 src = (gptr_f4)d.src;
    const gptr_f4 tgt = (gptr_f4)d.tgt;
    const gptr_f imap = (gptr_f)d.imap;
    const gptr_f src_points = (gptr_f)d.src_points, tgt_points = (gptr_f)d.tgt_points,
                 tgt_normals = (gptr_f)d.tgt_normals;
    const gptr_u8 src_mask = (gptr_u8)d.src_mask, src_int = (gptr_u8)d.src_intensities,
                  tgt_mask = (gptr_u8)d.tgt_mask;
    const uint32_t mw = d.tw + 2;
    const float twf = (float)d.tw, thf = (float)d.th;
    const uint32_t base = blockIdx.x * (256u * PPT) + threadIdx.x;
    // stage A (source records) runs one batch ahead: the loads of batch k+1 are issued right after the
    // target gathers of batch k, so they fly under its gates, map fetches and accumulation
    auto load_source = [&](int k0, f32x4(&sv)[G], bool(&lv)[G]) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const uint32_t i = base + (uint32_t)(k0 + g) * 256u;
        const bool inb = (k0 < PPT) && (i < d.src_n);
        const uint32_t ii = inb ? i : 0u;
        if constexpr (RAW) {
          const f32x3 sp = load3(src_points, ii);
          sv[g].x = sp.x, sv[g].y = sp.y, sv[g].z = sp.z;
          sv[g].w = (float)src_int[ii];
          lv[g] = inb && (src_mask[ii] != 0);  // mask != 0 (image_icp.rs:102)
        } else {
          sv[g] = src[ii];
          lv[g] = inb && (sv[g].w >= 0.0f);
        }
      }
    };
    f32x4 s_next[G];
    bool live_next[G];
    load_source(0, s_next, live_next);
#pragma unroll 1
    for (int k0 = 0; k0 < PPT; k0 += G) {
      f32x4 s[G];
      bool live[G];
#pragma unroll
      for (int g = 0; g < G; ++g) s[g] = s_next[g], live[g] = live_next[g];
      // ---- stage B: transform, project, gather the target record -------------------------------
      V3 p[G];
      float u[G], v[G];
      f32x4 tp[G], tn[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        p[g] = transform_vector(T, V3{s[g].x, s[g].y, s[g].z});
        // CameraIntrinsics::project (src/camera.rs:64-70)
        u[g] = p[g].x * d.fx / p[g].z + d.cx;
        v[g] = p[g].y * d.fy / p[g].z + d.cy;
        // (u + 0.5) as i32 -> as usize -> get_point bounds test: in range iff -1 < x < dim (NaN casts to 0)
        const float ur = u[g] + 0.5f, vr = v[g] + 0.5f;
        live[g] = live[g] && !(ur <= -1.0f || ur >= twf || vr <= -1.0f || vr >= thf);
        const uint32_t col = (ur != ur) ? 0u : (uint32_t)(int)ur;
        const uint32_t row = (vr != vr) ? 0u : (uint32_t)(int)vr;
        const uint32_t tidx = live[g] ? row * d.tw + col : 0u;
        if constexpr (RAW) {
          const f32x3 tpp = load3(tgt_points, tidx), tnn = load3(tgt_normals, tidx);
          tp[g].x = tpp.x, tp[g].y = tpp.y, tp[g].z = tpp.z;
          tp[g].w = tgt_mask[tidx] == 1 ? 1.0f : 0.0f;
          tn[g].x = tnn.x, tn[g].y = tnn.y, tn[g].z = tnn.z;
          tn[g].w = 0.0f;
        } else {
          tp[g] = tgt[2 * tidx];
          tn[g] = tgt[2 * tidx + 1];
        }
      }
      if (PPT > G) load_source(k0 + G, s_next, live_next);
      // ---- stage C: gates, intensity-map cell -----------------------------------------------------
      float t00[G], t10[G], t01[G], t11[G];
      uint32_t ui[G], vi[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const V3 diff = V3{tp[g].x, tp[g].y, tp[g].z} - p[g];
        // angle_between_normals(&p, &n) >= max_normal_angle with NaN (|p.n| > 1) passing (image_icp.rs:118-123)
        const float pn = dot(p[g], V3{tn[g].x, tn[g].y, tn[g].z});
        live[g] = live[g] && (tp[g].w == 1.0f)                            // mask == 1 (structure.rs:176)
                  && !(norm_squared(diff) > gt.max_distance_sqr)         // image_icp.rs:114
                  && !(pn >= -1.0f && pn <= gt.dot_reject_max);
        ui[g] = live[g] ? f32_as_usize(u[g]) : 0u;
        vi[g] = live[g] ? f32_as_usize(v[g]) : 0u;
        const gptr_f r0 = imap + (size_t)vi[g] * mw + ui[g];
        t00[g] = r0[0], t10[g] = r0[1], t01[g] = r0[mw], t11[g] = r0[mw + 1];
      }
      // ---- stage D: residuals, Jacobians, accumulate -----------------------------------------------
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (!live[g]) continue;
        const V3 P = p[g];
        const V3 n{tn[g].x, tn[g].y, tn[g].z};
        {  // PointPlaneDistance::jacobian (src/icp/cost_function.rs:33-41)
          const V3 diff = V3{tp[g].x, tp[g].y, tp[g].z} - P;
          const float r = dot(diff, n);
          const V3 tw = cross(P, n);
          const float J[6] = {n.x, n.y, n.z, tw.x, tw.y, tw.z};
          gn_step(acc, r, J);
        }
        // IntensityMap::bilinear_grad (src/intensity_map.rs:184-210), H = 0.005
        const float uf = u[g] - (float)ui[g], vf = v[g] - (float)vi[g];
        const float value = bilerp(t00[g], t10[g], t01[g], t11[g], uf, vf);
        const float Hh = 0.005f, H_INV = 1.0f / 0.005f;
        const float u2 = u[g] + Hh, v2 = v[g] + Hh;
        // the shifted samples share the cell except within 0.005 of a texel boundary
        const float uh = (f32_as_usize(u2) == ui[g]) ? bilerp(t00[g], t10[g], t01[g], t11[g], u2 - (float)ui[g], vf)
                                                     : bilinear_at(imap, mw, u2, v[g]);
        const float vh = (f32_as_usize(v2) == vi[g]) ? bilerp(t00[g], t10[g], t01[g], t11[g], uf, v2 - (float)vi[g])
                                                     : bilinear_at(imap, mw, u[g], v2);
        const float du = (uh - value) * H_INV;
        const float dv = (vh - value) * H_INV;
        const float sc = s[g].w * 0.003921569f;  // image_icp.rs:131
        // CameraIntrinsics::project_grad (src/camera.rs:82-89)
        const float z = P.z, zz = z * z;
        const float dfx = d.fx / z, dcx = -P.x * d.fx / zz;
        const float dfy = d.fy / z, dcy = -P.y * d.fy / zz;
        const V3 gr{du * dfx, dv * dfy, du * dcx + dv * dcy};
        const float rc = sc - value;
        if (rc * rc <= gt.max_color_distance_sqr) {  // image_icp.rs:136
          const V3 tw = cross(P, gr);
          const float J[6] = {gr.x, gr.y, gr.z, tw.x, tw.y, tw.z};
          gn_step(acc + GN_ACC, rc, J);
        }
      }
    }
  }
  // a failed job stays frozen: its blocks contribute nothing and nobody runs its solve
  SolveArgs sa = solve;
  if (st->status != A3D_OK) sa.mode = SOLVE_NONE;
  block_finish<GN_PARTIAL>(acc, partials + (size_t)pair * gridDim.x * GN_PARTIAL, blockIdx.x, gridDim.x,
                           counters + pair, st, sa, pair);
}


// ---- MFMA accumulation -----------------------------------------------------------------------------
// The per-pixel sums  H += J J^T, g += J r, ssq += r^2, count += 1  for the geometric and the colour
// term are all entries of X^T X, where row p of X holds pixel p's 16 "features"
//   [ Jg(6) | rg | Jc(6) | rc | live_g | live_c ].
// v_mfma_f32_16x16x4_f32 computes a 16x16 f32 tile of A B with K = 4, exact f32 fma chains; with
// A = X^T and B = X both operands are the SAME register: lane l supplies X[pixel l>>4][feature l&15]
// (CDNA guide §3).  A wave's 64 pixels therefore take 16 MFMAs, fed by a transpose through a 4 KiB LDS
// slab per wave: lane p writes its 16 features as one row, then reads X[4m + (l>>4)][l&15] for MFMA m.
// What it buys: the 58 per-thread accumulators (58 VGPRs) become one 16x16 tile = 8 VGPRs (two
// interleaved tiles), occupancy doubles, and the 54 accumulate FMAs per pixel leave the VALU for the
// matrix pipe, which runs beside it.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Row p of the slab is 16 floats; the four 16-byte chunks of a row are XOR-swizzled by (p >> 1) & 3 so
// that the b128 row writes of 8 neighbouring lanes and the b32 transposed reads are both conflict-free.
__device__ __forceinline__ int feat_chunk(int row, int chunk) { return row * 16 + ((chunk ^ ((row >> 1) & 3)) << 2); }

// Index into the 16x16 tile of partial entry k (0..57): geometric then colour accumulator, each
// 21 upper-triangle H, 6 g, ssq, count.
__device__ __forceinline__ int tile_index_of_partial(int k) {
  const int a = k >= GN_ACC ? 1 : 0, kk = k - a * GN_ACC, base = a * 7;
  int row, col;
  if (kk < 21) {
    int i = 0, rem = kk;
    while (rem >= 6 - i) { rem -= 6 - i; ++i; }
    row = base + i, col = base + i + rem;
  } else if (kk < 27) {
    row = base + (kk - 21), col = base + 6;
  } else if (kk == 27) {
    row = col = base + 6;
  } else {
    row = col = 14 + a;
  }
  return row * 16 + col;
}

template <int PPT, int G, bool RAW>
__global__ void __launch_bounds__(256)
    image_icp_mfma_kernel(const LevelDesc* __restrict__ descs, JobState* __restrict__ states, Gates gt,
                          float* __restrict__ partials, unsigned* __restrict__ counters, SolveArgs solve) {
  static_assert(PPT % G == 0, "PPT must be a multiple of G");
  __shared__ __attribute__((aligned(16))) float slab[4][64 * 16];  // per wave: 64 pixels x 16 features
  const int pair = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const my_slab = slab[wave];
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  JobState* st = &states[pair];
  if (st->status == A3D_OK) {
    const LevelDesc d = descs[pair];
    const Pose T = st->pose;
    const gptr_f4 src = (gptr_f4)d.src;
    const gptr_f4 tgt = (gptr_f4)d.tgt;
    const gptr_f imap = (gptr_f)d.imap;
    const gptr_f src_points = (gptr_f)d.src_points, tgt_points = (gptr_f)d.tgt_points,
                 tgt_normals = (gptr_f)d.tgt_normals;
    const gptr_u8 src_mask = (gptr_u8)d.src_mask, src_int = (gptr_u8)d.src_intensities,
                  tgt_mask = (gptr_u8)d.tgt_mask;
    const uint32_t mw = d.tw + 2;
    const float twf = (float)d.tw, thf = (float)d.th;
    const uint32_t base = blockIdx.x * (256u * PPT) + threadIdx.x;
    // transposed-read offsets of this lane: row 4m + (lane >> 4), feature lane & 15
    const int rd_row0 = lane >> 4, rd_chunk = (lane & 15) >> 2, rd_word = lane & 3;
    auto load_source = [&](int k0, f32x4(&sv)[G], bool(&lv)[G]) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const uint32_t i = base + (uint32_t)(k0 + g) * 256u;
        const bool inb = (k0 < PPT) && (i < d.src_n);
        const uint32_t ii = inb ? i : 0u;
        if constexpr (RAW) {
          const f32x3 sp = load3(src_points, ii);
          sv[g].x = sp.x, sv[g].y = sp.y, sv[g].z = sp.z;
          sv[g].w = (float)src_int[ii];
          lv[g] = inb && (src_mask[ii] != 0);
        } else {
          sv[g] = src[ii];
          lv[g] = inb && (sv[g].w >= 0.0f);
        }
      }
    };
    f32x4 s_next[G];
    bool live_next[G];
    load_source(0, s_next, live_next);
#pragma unroll 1
    for (int k0 = 0; k0 < PPT; k0 += G) {
      f32x4 s[G];
      bool live[G];
#pragma unroll
      for (int g = 0; g < G; ++g) s[g] = s_next[g], live[g] = live_next[g];
      V3 p[G];
      float u[G], v[G];
      f32x4 tp[G], tn[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        p[g] = transform_vector(T, V3{s[g].x, s[g].y, s[g].z});
        u[g] = p[g].x * d.fx / p[g].z + d.cx;
        v[g] = p[g].y * d.fy / p[g].z + d.cy;
        const float ur = u[g] + 0.5f, vr = v[g] + 0.5f;
        live[g] = live[g] && !(ur <= -1.0f || ur >= twf || vr <= -1.0f || vr >= thf);
        const uint32_t col = (ur != ur) ? 0u : (uint32_t)(int)ur;
        const uint32_t row = (vr != vr) ? 0u : (uint32_t)(int)vr;
        const uint32_t tidx = live[g] ? row * d.tw + col : 0u;
        if constexpr (RAW) {
          const f32x3 tpp = load3(tgt_points, tidx), tnn = load3(tgt_normals, tidx);
          tp[g].x = tpp.x, tp[g].y = tpp.y, tp[g].z = tpp.z;
          tp[g].w = tgt_mask[tidx] == 1 ? 1.0f : 0.0f;
          tn[g].x = tnn.x, tn[g].y = tnn.y, tn[g].z = tnn.z;
          tn[g].w = 0.0f;
        } else {
          tp[g] = tgt[2 * tidx];
          tn[g] = tgt[2 * tidx + 1];
        }
      }
      if (PPT > G) load_source(k0 + G, s_next, live_next);
      float t00[G], t10[G], t01[G], t11[G];
      uint32_t ui[G], vi[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const V3 diff = V3{tp[g].x, tp[g].y, tp[g].z} - p[g];
        const float pn = dot(p[g], V3{tn[g].x, tn[g].y, tn[g].z});
        live[g] = live[g] && (tp[g].w == 1.0f) && !(norm_squared(diff) > gt.max_distance_sqr) &&
                  !(pn >= -1.0f && pn <= gt.dot_reject_max);
        ui[g] = live[g] ? f32_as_usize(u[g]) : 0u;
        vi[g] = live[g] ? f32_as_usize(v[g]) : 0u;
        const gptr_f r0 = imap + (size_t)vi[g] * mw + ui[g];
        t00[g] = r0[0], t10[g] = r0[1], t01[g] = r0[mw], t11[g] = r0[mw + 1];
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        // ---- features of this lane's pixel (all zero when it is gated out) ----------------------
        f32x4_t f0 = {0.f, 0.f, 0.f, 0.f}, f1 = f0, f2 = f0, f3 = f0;
        if (live[g]) {
          const V3 P = p[g];
          const V3 n{tn[g].x, tn[g].y, tn[g].z};
          const V3 diff = V3{tp[g].x, tp[g].y, tp[g].z} - P;
          const float r = dot(diff, n);
          const V3 tw = cross(P, n);
          f0 = f32x4_t{n.x, n.y, n.z, tw.x};
          f1.x = tw.y, f1.y = tw.z, f1.z = r;
          f3.z = 1.0f;
          const float uf = u[g] - (float)ui[g], vf = v[g] - (float)vi[g];
          const float value = bilerp(t00[g], t10[g], t01[g], t11[g], uf, vf);
          const float Hh = 0.005f, H_INV = 1.0f / 0.005f;
          const float u2 = u[g] + Hh, v2 = v[g] + Hh;
          const float uh = (f32_as_usize(u2) == ui[g])
                               ? bilerp(t00[g], t10[g], t01[g], t11[g], u2 - (float)ui[g], vf)
                               : bilinear_at(imap, mw, u2, v[g]);
          const float vh = (f32_as_usize(v2) == vi[g])
                               ? bilerp(t00[g], t10[g], t01[g], t11[g], uf, v2 - (float)vi[g])
                               : bilinear_at(imap, mw, u[g], v2);
          const float du = (uh - value) * H_INV;
          const float dv = (vh - value) * H_INV;
          const float sc = s[g].w * 0.003921569f;
          const float z = P.z, zz = z * z;
          const float dfx = d.fx / z, dcx = -P.x * d.fx / zz;
          const float dfy = d.fy / z, dcy = -P.y * d.fy / zz;
          const V3 gr{du * dfx, dv * dfy, du * dcx + dv * dcy};
          const float rc = sc - value;
          if (rc * rc <= gt.max_color_distance_sqr) {
            const V3 twc = cross(P, gr);
            f1.w = gr.x;
            f2 = f32x4_t{gr.y, gr.z, twc.x, twc.y};
            f3.x = twc.z, f3.y = rc, f3.w = 1.0f;
          }
        }
        // ---- transpose through the wave's LDS slab, 16 MFMAs --------------------------------------
        *(f32x4_t*)(my_slab + feat_chunk(lane, 0)) = f0;
        *(f32x4_t*)(my_slab + feat_chunk(lane, 1)) = f1;
        *(f32x4_t*)(my_slab + feat_chunk(lane, 2)) = f2;
        *(f32x4_t*)(my_slab + feat_chunk(lane, 3)) = f3;
        float x[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) x[m] = my_slab[feat_chunk(4 * m + rd_row0, rd_chunk) + rd_word];
#pragma unroll
        for (int m = 0; m < 16; m += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[m], x[m], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[m + 1], x[m + 1], acc1, 0, 0, 0);
        }
      }
    }
  }
  // ---- block partial: sum the four waves' tiles, pick the 58 entries, store write-through ----------
  __syncthreads();  // every wave is done with its slab; reuse the first 4 KiB as [wave][256]
  float* tiles_lds = &slab[0][0];
  {
    const f32x4_t t = acc0 + acc1;  // C/D map: col = lane & 15, row = 4 (lane >> 4) + reg
    const int colx = lane & 15, row4 = (lane >> 4) * 4;
    tiles_lds[wave * 256 + (row4 + 0) * 16 + colx] = t.x;
    tiles_lds[wave * 256 + (row4 + 1) * 16 + colx] = t.y;
    tiles_lds[wave * 256 + (row4 + 2) * 16 + colx] = t.z;
    tiles_lds[wave * 256 + (row4 + 3) * 16 + colx] = t.w;
  }
  __syncthreads();
  float* job_partials = partials + (size_t)pair * gridDim.x * GN_PARTIAL;
  if (threadIdx.x < GN_PARTIAL) {
    const int e = tile_index_of_partial(threadIdx.x);
    const float v = (tiles_lds[e] + tiles_lds[256 + e]) + (tiles_lds[512 + e] + tiles_lds[768 + e]);
    __hip_atomic_store((unsigned*)(job_partials + (size_t)blockIdx.x * GN_PARTIAL) + threadIdx.x, __float_as_uint(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  SolveArgs sa = solve;
  if (st->status != A3D_OK) sa.mode = SOLVE_NONE;
  block_publish_and_finish(job_partials, gridDim.x, counters + pair, st, sa, pair);
}

}  // namespace

// P independent coarse-to-fine alignments.  Owns only small state; the images are borrowed.
struct a3d_multiscale_batch {
  a3d_context* ctx = nullptr;
  uint32_t n_pairs = 0, n_levels = 0;
  std::vector<a3d_icp_params> params;  // index 0 = finest
  std::vector<Gates> gates;
  std::vector<uint32_t> tiles, ppt, group;  // per level: tiles per pair, pixels per thread, pixels in flight
  std::vector<LevelDesc> h_descs;      // [level][pair]
  LevelDesc* d_descs = nullptr;
  JobState* d_states = nullptr;
  float* d_partials = nullptr;
  unsigned* d_counters = nullptr;  // per pair: blocks that have published their partial in this launch
  Pose* d_poses = nullptr;
  int32_t* d_status = nullptr;
  double* d_readback = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> kev;  // per pixel-kernel launch: start/stop pairs, when profiling
  bool profile_kernels = false;
  bool raw_layout = true;
  bool use_mfma = false;  // VALU accumulation measured faster on MI355X so far (kernel variants)
  float last_total_ms = 0.f, last_kernel_ms = 0.f;
  uint64_t last_kernel_launches = 0;

  ~a3d_multiscale_batch() {
    hipFree(d_descs);
    hipFree(d_states);
    hipFree(d_partials);
    hipFree(d_counters);
    hipFree(d_poses);
    hipFree(d_status);
    hipFree(d_readback);
    if (ev0) hipEventDestroy(ev0);
    if (ev1) hipEventDestroy(ev1);
    for (auto e : kev) hipEventDestroy(e);
  }
};

namespace {

a3d_status fill_desc(const a3d_device_image* target, const a3d_device_image* source, LevelDesc* d) {
  A3D_REQUIRE(target && source, A3D_INVALID_PARAMETER, "null image handle");
  // the reference `expect`s these three (image_icp.rs:44-57)
  A3D_REQUIRE(target->has_imap, A3D_MISSING_FIELD, "Please, the target image should have a intensity map.");
  A3D_REQUIRE(target->has_normals, A3D_MISSING_FIELD, "Please, the target image should have normals.");
  A3D_REQUIRE(source->has_intensities, A3D_MISSING_FIELD,
              "Please, the source image should have intensity colors.");
  A3D_REQUIRE(target->tgt_pack_valid && source->src_pack_valid, A3D_INVALID_PARAMETER, "image not packed");
  d->src = source->src_pack;
  d->tgt = target->tgt_pack;
  d->imap = target->imap;
  d->src_points = source->points, d->src_mask = source->mask, d->src_intensities = source->intensities;
  d->tgt_points = target->points, d->tgt_normals = target->normals, d->tgt_mask = target->mask;
  d->src_n = source->width * source->height;
  d->tw = target->width;
  d->th = target->height;
  d->fx = target->fx, d->fy = target->fy, d->cx = target->cx, d->cy = target->cy;
  d->pad = 0;
  return A3D_OK;
}

Gates make_gates(const a3d_icp_params& p) {
  Gates g;
  g.max_distance_sqr = p.max_distance * p.max_distance;
  g.max_color_distance_sqr = p.max_color_distance * p.max_color_distance;
  g.dot_reject_max = acos_gate_threshold(p.max_normal_angle, /*strict=*/false);
  return g;
}

// Pixels per thread: as many as keep >= 1024 blocks in flight (4 per CU), at most 8.
uint32_t choose_ppt(uint32_t n_pairs, uint32_t max_src_n) {
  for (uint32_t ppt = 8; ppt > 1; ppt >>= 1) {
    uint64_t blocks = (uint64_t)n_pairs * ((max_src_n + 256 * ppt - 1) / (256 * ppt));
    if (blocks >= 1024) return ppt;
  }
  return 1;
}

a3d_status launch_pixel_kernel(a3d_multiscale_batch* b, uint32_t level, const SolveArgs& solve) {
  dim3 grid(b->tiles[level], b->n_pairs), block(256);
  const LevelDesc* descs = b->d_descs + (size_t)level * b->n_pairs;
  hipStream_t s = b->ctx->stream;
#define A3D_LAUNCH(PPT, G)                                                                               \
  do {                                                                                                      \
    if (b->use_mfma && b->raw_layout)                                                                       \
      hipLaunchKernelGGL((image_icp_mfma_kernel<PPT, G, true>), grid, block, 0, s, descs, b->d_states,      \
                         b->gates[level], b->d_partials, b->d_counters, solve);                             \
    else if (b->use_mfma)                                                                                   \
      hipLaunchKernelGGL((image_icp_mfma_kernel<PPT, G, false>), grid, block, 0, s, descs, b->d_states,     \
                         b->gates[level], b->d_partials, b->d_counters, solve);                             \
    else if (b->raw_layout)                                                                                 \
      hipLaunchKernelGGL((image_icp_kernel<PPT, G, true>), grid, block, 0, s, descs, b->d_states,           \
                         b->gates[level], b->d_partials, b->d_counters, solve);                             \
    else                                                                                                    \
      hipLaunchKernelGGL((image_icp_kernel<PPT, G, false>), grid, block, 0, s, descs, b->d_states,          \
                         b->gates[level], b->d_partials, b->d_counters, solve);                             \
  } while (0)
  const uint32_t g = b->group[level];
  switch (b->ppt[level] * 16 + g) {
    case 16 * 16 + 2: A3D_LAUNCH(16, 2); break;
    case 16 * 16 + 1: A3D_LAUNCH(16, 1); break;
    case 32 * 16 + 2: A3D_LAUNCH(32, 2); break;
    case 8 * 16 + 4: A3D_LAUNCH(8, 4); break;
    case 8 * 16 + 2: A3D_LAUNCH(8, 2); break;
    case 8 * 16 + 1: A3D_LAUNCH(8, 1); break;
    case 4 * 16 + 4: A3D_LAUNCH(4, 4); break;
    case 4 * 16 + 2: A3D_LAUNCH(4, 2); break;
    case 4 * 16 + 1: A3D_LAUNCH(4, 1); break;
    case 2 * 16 + 2: A3D_LAUNCH(2, 2); break;
    case 2 * 16 + 1: A3D_LAUNCH(2, 1); break;
    case 1 * 16 + 1: A3D_LAUNCH(1, 1); break;
    default:
      set_error("unsupported kernel variant ppt=%u g=%u", b->ppt[level], g);
      return A3D_INVALID_PARAMETER;
  }
#undef A3D_LAUNCH
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

// (Re)derives tiling from h_descs and uploads the descriptors.
a3d_status batch_commit_descs(a3d_multiscale_batch* b) {
  const uint32_t P = b->n_pairs, L = b->n_levels;
  size_t max_partials = 1;
  for (uint32_t l = 0; l < L; ++l) {
    uint32_t max_n = 0;
    for (uint32_t p = 0; p < P; ++p) max_n = std::max(max_n, b->h_descs[(size_t)l * P + p].src_n);
    b->ppt[l] = choose_ppt(P, max_n);
    b->group[l] = std::min<uint32_t>(b->ppt[l], 2);  // measured best on MI355X (kernel variants)
    if (const char* env = getenv("A3D_ICP_VARIANT")) {  // tuning knob: "ppt,g"
      unsigned ep = 0, eg = 0;
      if (sscanf(env, "%u,%u", &ep, &eg) == 2 && ep && eg) b->ppt[l] = ep, b->group[l] = eg;
    }
    b->tiles[l] = (max_n + 256 * b->ppt[l] - 1) / (256 * b->ppt[l]);
    max_partials = std::max(max_partials, (size_t)P * b->tiles[l] * GN_PARTIAL);
  }
  if (b->d_partials) A3D_HIP_TRY(hipFree(b->d_partials));
  b->d_partials = nullptr;
  A3D_HIP_TRY(hipMalloc((void**)&b->d_partials, max_partials * sizeof(float)));
  A3D_HIP_TRY(hipMemcpyAsync(b->d_descs, b->h_descs.data(), b->h_descs.size() * sizeof(LevelDesc),
                             hipMemcpyHostToDevice, b->ctx->stream));
  return A3D_OK;
}

a3d_status batch_create(a3d_context* ctx, const a3d_icp_params* params, uint32_t n_levels, uint32_t n_pairs,
                        std::unique_ptr<a3d_multiscale_batch>* out) {
  auto b = std::make_unique<a3d_multiscale_batch>();
  b->ctx = ctx;
  b->n_pairs = n_pairs;
  b->n_levels = n_levels;
  b->params.assign(params, params + n_levels);
  for (uint32_t l = 0; l < n_levels; ++l) b->gates.push_back(make_gates(params[l]));
  b->tiles.assign(n_levels, 0);
  b->ppt.assign(n_levels, 1);
  b->group.assign(n_levels, 1);
  b->h_descs.resize((size_t)n_levels * n_pairs);
  if (const char* env = getenv("A3D_ICP_LAYOUT")) b->raw_layout = strcmp(env, "packed") != 0;  // tuning knob
  if (const char* env = getenv("A3D_ICP_ACCUM")) b->use_mfma = strcmp(env, "mfma") == 0;        // tuning knob
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  A3D_HIP_TRY(hipMalloc((void**)&b->d_descs, b->h_descs.size() * sizeof(LevelDesc)));
  A3D_HIP_TRY(hipMalloc((void**)&b->d_states, n_pairs * sizeof(JobState)));
  A3D_HIP_TRY(hipMalloc((void**)&b->d_counters, n_pairs * sizeof(unsigned)));
  A3D_HIP_TRY(hipMemsetAsync(b->d_counters, 0, n_pairs * sizeof(unsigned), ctx->stream));
  A3D_HIP_TRY(hipMalloc((void**)&b->d_poses, n_pairs * sizeof(Pose)));
  A3D_HIP_TRY(hipMalloc((void**)&b->d_status, n_pairs * sizeof(int32_t)));
  A3D_HIP_TRY(hipMalloc((void**)&b->d_readback, GN_PARTIAL * sizeof(double)));
  A3D_HIP_TRY(hipEventCreate(&b->ev0));
  A3D_HIP_TRY(hipEventCreate(&b->ev1));
  *out = std::move(b);
  return A3D_OK;
}

// Enqueues init -> levels (coarsest first) -> finish.  d_init: device Pose[P] or null (identity).
a3d_status batch_enqueue(a3d_multiscale_batch* b, const Pose* d_init, uint32_t levels_to_run, float* d_matrices,
                         float* d_trace, int trace_stride) {
  hipStream_t s = b->ctx->stream;
  const uint32_t P = b->n_pairs;
  A3D_HIP_TRY(hipEventRecord(b->ev0, s));
  A3D_TRY(launch_job_init(s, b->d_states, d_init, (int)P));
  size_t kidx = 0;
  int trace_index = 0;
  for (uint32_t l = levels_to_run; l-- > 0;) {  // .rev(): coarsest level first (multiscale.rs:54-60)
    const a3d_icp_params& prm = b->params[l];
    for (uint64_t it = 0; it < prm.max_iterations; ++it) {
      if (b->profile_kernels) {
        if (b->kev.size() < 2 * (kidx + 1)) {
          hipEvent_t e0, e1;
          A3D_HIP_TRY(hipEventCreate(&e0));
          A3D_HIP_TRY(hipEventCreate(&e1));
          b->kev.push_back(e0);
          b->kev.push_back(e1);
        }
        A3D_HIP_TRY(hipEventRecord(b->kev[2 * kidx], s));
      }
      SolveArgs sa;
      sa.weight = prm.weight, sa.color_weight = prm.color_weight;
      sa.mode = SOLVE_IMAGE_ICP;
      sa.first_in_level = it == 0, sa.last_in_level = it + 1 == prm.max_iterations;
      sa.trace = d_trace, sa.trace_stride = trace_stride, sa.trace_index = trace_index;
      A3D_TRY(launch_pixel_kernel(b, l, sa));
      if (b->profile_kernels) A3D_HIP_TRY(hipEventRecord(b->kev[2 * kidx + 1], s));
      ++kidx;
      ++trace_index;
    }
  }
  A3D_TRY(launch_job_finish(s, b->d_states, b->d_poses, b->d_status, d_matrices, (int)P));
  A3D_HIP_TRY(hipEventRecord(b->ev1, s));
  b->last_kernel_launches = kidx;
  return A3D_OK;
}

a3d_status batch_collect_timing(a3d_multiscale_batch* b) {
  A3D_HIP_TRY(hipEventSynchronize(b->ev1));
  A3D_HIP_TRY(hipEventElapsedTime(&b->last_total_ms, b->ev0, b->ev1));
  b->last_kernel_ms = 0.f;
  if (b->profile_kernels) {
    for (size_t k = 0; k < b->last_kernel_launches; ++k) {
      float ms = 0.f;
      A3D_HIP_TRY(hipEventElapsedTime(&ms, b->kev[2 * k], b->kev[2 * k + 1]));
      b->last_kernel_ms += ms;
    }
  }
  return A3D_OK;
}

a3d_status read_results(a3d_multiscale_batch* b, a3d_pose* out_poses, int32_t* out_status, a3d_status* worst) {
  const uint32_t P = b->n_pairs;
  std::vector<Pose> poses(P);
  std::vector<int32_t> status(P);
  hipStream_t s = b->ctx->stream;
  A3D_HIP_TRY(hipMemcpyAsync(poses.data(), b->d_poses, P * sizeof(Pose), hipMemcpyDeviceToHost, s));
  A3D_HIP_TRY(hipMemcpyAsync(status.data(), b->d_status, P * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  A3D_HIP_TRY(hipStreamSynchronize(s));
  *worst = A3D_OK;
  for (uint32_t p = 0; p < P; ++p) {
    if (out_poses) pose_to_c(poses[p], &out_poses[p]);
    if (out_status) out_status[p] = status[p];
    if (status[p] != A3D_OK) *worst = (a3d_status)status[p];
  }
  return A3D_OK;
}

// One pair, any number of levels, host-synchronous: shared by image_icp_align and multiscale_align.
a3d_status align_single(a3d_context* ctx, const a3d_icp_params* params, uint32_t n_levels,
                        const a3d_device_image* const* targets, const a3d_device_image* const* sources,
                        const a3d_pose* init, a3d_pose* out_pose, float* host_trace) {
  std::unique_ptr<a3d_multiscale_batch> b;
  A3D_TRY(batch_create(ctx, params, n_levels, 1, &b));
  for (uint32_t l = 0; l < n_levels; ++l) A3D_TRY(fill_desc(targets[l], sources[l], &b->h_descs[l]));
  A3D_TRY(batch_commit_descs(b.get()));
  Pose* d_init = nullptr;
  Pose h_init;
  if (init) {
    h_init = pose_from_c(init);
    A3D_HIP_TRY(hipMalloc((void**)&d_init, sizeof(Pose)));
    A3D_HIP_TRY(hipMemcpyAsync(d_init, &h_init, sizeof(Pose), hipMemcpyHostToDevice, ctx->stream));
  }
  uint64_t total_iters = 0;
  for (uint32_t l = 0; l < n_levels; ++l) total_iters += params[l].max_iterations;
  float* d_trace = nullptr;
  if (host_trace && total_iters) {
    A3D_HIP_TRY(hipMalloc((void**)&d_trace, total_iters * 8 * sizeof(float)));
    A3D_HIP_TRY(hipMemsetAsync(d_trace, 0, total_iters * 8 * sizeof(float), ctx->stream));
  }
  a3d_status st = batch_enqueue(b.get(), d_init, n_levels, nullptr, d_trace, (int)total_iters);
  a3d_status worst = A3D_OK;
  if (st == A3D_OK) st = read_results(b.get(), out_pose, nullptr, &worst);
  if (st == A3D_OK && d_trace)
    if (hipMemcpy(host_trace, d_trace, total_iters * 8 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
      st = A3D_HIP_ERROR;
  hipFree(d_init);
  hipFree(d_trace);
  if (st != A3D_OK) return st;
  if (worst == A3D_SOLVE_FAILED) set_error("GaussNewton::solve() returned None (count == 0 or Cholesky failed)");
  return worst;
}

}  // namespace

// MultiscaleAlign: params + borrowed target pyramid (src/icp/multiscale.rs:7-10).
struct a3d_multiscale {
  a3d_context* ctx = nullptr;
  std::vector<a3d_icp_params> params;
  std::vector<const a3d_device_image*> targets;
};

extern "C" {

a3d_status a3d_image_icp_align(a3d_context* ctx, const a3d_icp_params* params, const a3d_device_image* target,
                               const a3d_device_image* source, const a3d_pose* init_pose, a3d_pose* out_pose) {
  A3D_REQUIRE(ctx && params && out_pose, A3D_INVALID_PARAMETER, "null argument");
  return align_single(ctx, params, 1, &target, &source, init_pose, out_pose, nullptr);
}

// Test hook (not part of the reference surface): a3d_image_icp_align that also returns, per
// iteration, [residual, t(3), q(4)] of the transform after that iteration's update.
a3d_status a3d_image_icp_align_trace(a3d_context* ctx, const a3d_icp_params* params,
                                     const a3d_device_image* target, const a3d_device_image* source,
                                     const a3d_pose* init_pose, a3d_pose* out_pose, float* out_trace) {
  A3D_REQUIRE(ctx && params && out_pose, A3D_INVALID_PARAMETER, "null argument");
  return align_single(ctx, params, 1, &target, &source, init_pose, out_pose, out_trace);
}

a3d_status a3d_image_icp_accumulate(a3d_context* ctx, const a3d_icp_params* params, const a3d_device_image* target,
                                    const a3d_device_image* source, const a3d_pose* pose, a3d_gn_state* out_geom,
                                    a3d_gn_state* out_color) {
  A3D_REQUIRE(ctx && params, A3D_INVALID_PARAMETER, "null argument");
  std::unique_ptr<a3d_multiscale_batch> b;
  A3D_TRY(batch_create(ctx, params, 1, 1, &b));
  A3D_TRY(fill_desc(target, source, &b->h_descs[0]));
  A3D_TRY(batch_commit_descs(b.get()));
  Pose h_pose = pose ? pose_from_c(pose) : pose_eye();
  Pose* d_pose = nullptr;
  A3D_HIP_TRY(hipMalloc((void**)&d_pose, sizeof(Pose)));
  a3d_status st = A3D_OK;
  double sums[GN_PARTIAL];
  hipStream_t s = ctx->stream;
  if (hipMemcpyAsync(d_pose, &h_pose, sizeof(Pose), hipMemcpyHostToDevice, s) != hipSuccess) st = A3D_HIP_ERROR;
  if (st == A3D_OK) st = launch_job_init(s, b->d_states, d_pose, 1);
  SolveArgs none{};
  none.mode = SOLVE_NONE;
  if (st == A3D_OK) st = launch_pixel_kernel(b.get(), 0, none);
  if (st == A3D_OK) st = launch_gn_readback(s, b->d_partials, (int)b->tiles[0], b->d_readback);
  if (st == A3D_OK && hipMemcpyAsync(sums, b->d_readback, sizeof(sums), hipMemcpyDeviceToHost, s) != hipSuccess)
    st = A3D_HIP_ERROR;
  if (hipStreamSynchronize(s) != hipSuccess) st = A3D_HIP_ERROR;
  hipFree(d_pose);
  if (st == A3D_HIP_ERROR) set_error("a3d_image_icp_accumulate: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  if (st != A3D_OK) return st;
  gn_states_from_sums(sums, out_geom, out_color);
  return A3D_OK;
}

a3d_status a3d_multiscale_new(a3d_context* ctx, const a3d_icp_params* params, uint64_t n_params,
                              const a3d_device_image* const* target_pyramid, uint64_t n_levels,
                              a3d_multiscale** out) {
  A3D_REQUIRE(ctx && out && (params || n_params == 0) && (target_pyramid || n_levels == 0), A3D_INVALID_PARAMETER,
              "null argument");
  // multiscale.rs:30-34
  A3D_REQUIRE(n_params == n_levels, A3D_INVALID_PARAMETER,
              "The number of range images pyramid levels and ICP parameters must be equal.");
  a3d_multiscale* ms = new a3d_multiscale();
  ms->ctx = ctx;
  ms->params.assign(params, params + n_params);
  ms->targets.assign(target_pyramid, target_pyramid + n_levels);
  *out = ms;
  return A3D_OK;
}

a3d_status a3d_multiscale_align(a3d_multiscale* ms, const a3d_device_image* const* source_pyramid,
                                uint64_t n_source_levels, a3d_pose* out_pose) {
  A3D_REQUIRE(ms && out_pose && (source_pyramid || n_source_levels == 0), A3D_INVALID_PARAMETER, "null argument");
  // izip! stops at the shortest of (params, targets, sources) BEFORE .rev() (multiscale.rs:54-59)
  uint32_t n = (uint32_t)std::min<uint64_t>(ms->params.size(), n_source_levels);
  if (n == 0) {
    pose_to_c(pose_eye(), out_pose);
    return A3D_OK;
  }
  return align_single(ms->ctx, ms->params.data(), n, ms->targets.data(), source_pyramid, nullptr, out_pose, nullptr);
}

a3d_status a3d_multiscale_free(a3d_multiscale* ms) {
  delete ms;
  return A3D_OK;
}

a3d_status a3d_multiscale_batch_new(a3d_context* ctx, const a3d_icp_params* params, uint64_t n_params,
                                    uint64_t n_pairs, uint64_t n_levels,
                                    const a3d_device_image* const* target_pyramids,
                                    const a3d_device_image* const* source_pyramids, a3d_multiscale_batch** out) {
  A3D_REQUIRE(ctx && params && target_pyramids && source_pyramids && out, A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(n_params == n_levels, A3D_INVALID_PARAMETER,
              "The number of range images pyramid levels and ICP parameters must be equal.");
  A3D_REQUIRE(n_pairs > 0 && n_pairs <= 65535 && n_levels > 0 && n_levels <= 16, A3D_INVALID_PARAMETER,
              "bad batch shape");
  std::unique_ptr<a3d_multiscale_batch> b;
  A3D_TRY(batch_create(ctx, params, (uint32_t)n_levels, (uint32_t)n_pairs, &b));
  for (uint32_t p = 0; p < n_pairs; ++p)
    for (uint32_t l = 0; l < n_levels; ++l)
      A3D_TRY(fill_desc(target_pyramids[(size_t)p * n_levels + l], source_pyramids[(size_t)p * n_levels + l],
                        &b->h_descs[(size_t)l * n_pairs + p]));
  A3D_TRY(batch_commit_descs(b.get()));
  A3D_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *out = b.release();
  return A3D_OK;
}

a3d_status a3d_multiscale_batch_align(a3d_multiscale_batch* b, a3d_pose* out_poses_host, float* out_matrices_device,
                                      int32_t* out_status_host) {
  A3D_REQUIRE(b, A3D_INVALID_PARAMETER, "batch is null");
  A3D_TRY(batch_enqueue(b, nullptr, b->n_levels, out_matrices_device, nullptr, 0));
  if (!out_poses_host && !out_status_host) return A3D_OK;
  a3d_status worst;
  A3D_TRY(read_results(b, out_poses_host, out_status_host, &worst));
  return A3D_OK;  // per-pair failures are reported through out_status_host
}

// When on, every per-pixel kernel launch is bracketed by its own hipEvent pair on the context stream.
a3d_status a3d_multiscale_batch_set_profiling(a3d_multiscale_batch* b, int32_t on) {
  A3D_REQUIRE(b, A3D_INVALID_PARAMETER, "batch is null");
  b->profile_kernels = on != 0;
  return A3D_OK;
}

a3d_status a3d_multiscale_batch_last_timing(a3d_multiscale_batch* b, float* out_total_ms,
                                            uint64_t* out_pixel_kernel_launches) {
  A3D_REQUIRE(b, A3D_INVALID_PARAMETER, "batch is null");
  A3D_TRY(batch_collect_timing(b));
  if (out_total_ms) *out_total_ms = b->last_total_ms;
  if (out_pixel_kernel_launches) *out_pixel_kernel_launches = b->last_kernel_launches;
  return A3D_OK;
}

// Sum of the per-pixel kernel's launch durations in the most recent batch_align (profiling on).
a3d_status a3d_multiscale_batch_last_kernel_ms(a3d_multiscale_batch* b, float* out_kernel_ms) {
  A3D_REQUIRE(b && out_kernel_ms, A3D_INVALID_PARAMETER, "null argument");
  A3D_TRY(batch_collect_timing(b));
  *out_kernel_ms = b->last_kernel_ms;
  return A3D_OK;
}

a3d_status a3d_multiscale_batch_free(a3d_multiscale_batch* b) {
  if (!b) return A3D_OK;
  hipStreamSynchronize(b->ctx->stream);
  delete b;
  return A3D_OK;
}

}  // extern "C"
