// ImageIcp::align (src/icp/image_icp.rs:43-164) and MultiscaleAlign (src/icp/multiscale.rs:26-67)
// for P independent frame pairs at once: per iteration ONE kernel per stream group of pairs (grid = tiles x
// pairs).  Every block stores its partial sums plainly; every block of the NEXT launch first finishes the previous
// iteration at its head (sums its pair's partials in a fixed order and runs the Gauss-Newton solve and the pose
// update redundantly: icp_engine.hpp, head_advance) — no atomics, no last block.  The whole coarse-to-fine
// sequence is enqueued without a host round trip.
#include <cstdlib>
#include <memory>

#include "icp_engine.hpp"

using namespace a3d;

namespace {

// What the per-pixel kernel needs to know about one (pair, level): the resident arrays, in the
// reference's own layout (14 B per source pixel, 25 B per target pixel, 4 B per map texel), + intrinsics.
struct LevelDesc {
  const float* src_points;         // [src_n][3]
  const uint8_t* src_mask;         // [src_n]
  const uint8_t* src_intensities;  // [src_n]
  const float* tgt_points;         // [th*tw][3]
  const float* tgt_normals;        // [th*tw][3]
  const uint8_t* tgt_mask;         // [th*tw]
  const float* imap;               // [(th+2)][(tw+2)]
  uint32_t src_n;
  uint32_t tw, th;
  float fx, fy, cx, cy;
  uint32_t flags;  // bit 0: both images carry mask_is_z
  uint32_t ppt;    // source pixels per thread at this level: the pair's tiles are ceil(src_n / (256 ppt)) blocks
  uint32_t pad;
};
static_assert(sizeof(LevelDesc) == 96, "LevelDesc layout");

struct Gates {
  float max_distance_sqr;
  float max_color_distance_sqr;
  float dot_reject_max;  // reject iff -1 <= p.n <= dot_reject_max  (== acos(p.n).abs() >= max_normal_angle)
};

// `u as usize` (Rust): NaN and negatives -> 0; the callers only see u < width.  That is what v_cvt_u32_f32 does by
// itself (it saturates: NaN -> 0, negative -> 0), but a C++ cast of such a value is undefined, so the portable form
// costs two v_max before the conversion; the instruction is named instead.
__device__ __forceinline__ uint32_t f32_as_usize(float x) {
  uint32_t r;
  asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
// `x as i32` (Rust): saturating, NaN -> 0 = v_cvt_i32_f32.
__device__ __forceinline__ int32_t f32_as_i32(float x) {
  int32_t r;
  asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// ---- loads ------------------------------------------------------------------------------------------
// Array bases come out of a descriptor (uniform, SGPRs); every access is base + 32-bit byte offset in
// the global address space, which selects the `global_load … v_off, s[base]` form: no 64-bit address
// arithmetic in the VALU.  (Images are limited to 2^28 pixels at upload, so offsets fit.)
typedef const char __attribute__((address_space(1)))* gptr_c;
template <typename T>
__device__ __forceinline__ T ld(const void* base, uint32_t byte_off) {
  return *(const T __attribute__((address_space(1)))*)((gptr_c)base + byte_off);
}
// One 12-byte Vector3<f32> is ONE global_load_dwordx3.  Three dword loads at a 12-byte lane stride make
// the L1 look up every 64-byte line of the wave's 768-byte span three times (measured: 5.3 tag lookups
// per pixel, which bounded the first versions of this kernel).
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef f32x3 __attribute__((aligned(4))) f32x3_u;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x2 __attribute__((aligned(4))) f32x2_u;
__device__ __forceinline__ V3 ld_v3(const float* base, uint32_t idx) {
  // idx * 12 as two shifts and an add: v_mul_lo_u32 is a quarter-rate instruction (pixel indices are < 2^28)
  uint32_t off;
  asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(off) : "v"(idx));  // 3 idx
  const f32x3 v = ld<f32x3_u>(base, off << 2);
  return {v.x, v.y, v.z};
}

// (IEEE division with a shared reciprocal: div_prepare / div_by / div_den_ok / div_num_ok live in devmath.hpp)

// Byte offset of texel (row, col) in an f32 image `width` texels wide.  Written as a 24-bit multiply-add
// (rows, columns and widths are far below 2^24): the generic 32-bit form compiles to v_mad_u64_u32 with a 64-bit
// addend whose unused high half the register allocator parks on a register that still has a load in flight,
// and the resulting s_waitcnt vmcnt(0) drains the whole software pipeline once per pixel.
__device__ __forceinline__ uint32_t texel_offset(uint32_t row, uint32_t width, uint32_t col) {
  return (__umul24(row, width) + col) * 4u;
}

// IntensityMap::bilinear (src/intensity_map.rs:150-169) from four already-loaded texels.
__device__ __forceinline__ float bilerp(float v00, float v10, float v01, float v11, float uf, float vf) {
  float u0 = v00 * (1.0f - uf) + v10 * uf;
  float u1 = v01 * (1.0f - uf) + v11 * uf;
  return u0 * (1.0f - vf) + u1 * vf;
}
__device__ __forceinline__ float bilinear_at(const float* imap, uint32_t mw, float u, float v) {
  const uint32_t ui = f32_as_usize(u), vi = f32_as_usize(v);
  const uint32_t o = texel_offset(vi, mw, ui);
  const f32x2 a = ld<f32x2_u>(imap, o), b = ld<f32x2_u>(imap, o + mw * 4u);
  return bilerp(a.x, a.y, b.x, b.y, u - (float)ui, v - (float)vi);
}

// ---- the reference's pixel loop (image_icp.rs:101-139), in four stages -----------------------------------
struct SrcPx {  // stage A: one source record
  V3 sp;
  uint8_t intensity;   // the raw byte; converted to f32 where it is used (stage D), two steps after its load,
                       // so that the conversion does not wait for the load inside the step that issued it
  bool live;
};
// ZMASK: both images of the pair were made by the device frame builder with a depth scale that maps every depth
// unit to a non-zero z, so a pixel's mask is 1 exactly when its point's z is not 0 (invalid pixels are stored as
// (0, 0, 0), frame.hip) — the two mask bytes per pixel need not be read.  Decided per launch on the host.
template <bool ZMASK = false>
__device__ __forceinline__ SrcPx stage_a(const LevelDesc& d, uint32_t i, bool in_range) {
  const uint32_t ii = in_range ? i : 0u;
  SrcPx s;
  s.sp = ld_v3(d.src_points, ii);
  s.intensity = ld<uint8_t>(d.src_intensities, ii);
  if (ZMASK) {
    s.live = in_range & (s.sp.z != 0.0f);
    return s;
  }
  // the mask byte is loaded unconditionally: `in_range && load != 0` compiles to a branch around the load with an
  // s_waitcnt vmcnt(0) behind it, which drains every load in flight (the whole software pipeline) once per pixel
  const uint8_t mask = ld<uint8_t>(d.src_mask, ii);
  s.live = in_range & (mask != 0);  // mask != 0 (image_icp.rs:102)
  return s;
}

struct ProjPx {  // stage B: transformed point, projection, the gathered target record
  V3 p;
  float u, v;
  V3 tp, tn;
  uint8_t tmask;  // consumed in stage C, so that stage B only ISSUES the gathers
  bool live;
};
template <bool ZMASK = false>
__device__ __forceinline__ ProjPx stage_b(const LevelDesc& d, const Pose& T, const SrcPx& s, float twf, float thf) {
  ProjPx o;
  o.p = transform_vector(T, s.sp);
  // CameraIntrinsics::project (src/camera.rs:64-70): x * fx / z + cx
  const float z = o.p.z, au = o.p.x * d.fx, av = o.p.y * d.fy;
  const DivBy dz = div_prepare(z);
  float qu = div_by(au, dz), qv = div_by(av, dz);
  // rare: some lane is outside the fast range -> the whole wave takes the plain IEEE divide (same values
  // where both apply); a wave-uniform branch, so the slow sequence is not speculated into the hot path
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(div_den_ok(z) & div_num_ok(au) & div_num_ok(av))) != 0ull, 0))
    qu = au / z, qv = av / z;
  o.u = qu + d.cx;
  o.v = qv + d.cy;
  // (u + 0.5) as i32 -> as usize -> get_point bounds test: in range iff -1 < x < dim (NaN casts to 0)
  const float ur = o.u + 0.5f, vr = o.v + 0.5f;
  o.live = s.live & !((ur <= -1.0f) | (ur >= twf) | (vr <= -1.0f) | (vr >= thf));
  const uint32_t col = (uint32_t)f32_as_i32(ur), row = (uint32_t)f32_as_i32(vr);  // (NaN casts to 0)
  const uint32_t tidx = o.live ? __umul24(row, d.tw) + col : 0u;
  o.tp = ld_v3(d.tgt_points, tidx);
  o.tn = ld_v3(d.tgt_normals, tidx);
  o.tmask = ZMASK ? (uint8_t)0 : ld<uint8_t>(d.tgt_mask, tidx);
  return o;
}

struct MapPx {  // stage C: gates passed, the intensity-map cell
  float t00, t10, t01, t11;
  uint32_t ui, vi;
};
template <bool ZMASK = false>
__device__ __forceinline__ MapPx stage_c(const LevelDesc& d, const Gates& gt, ProjPx& px, uint32_t mw) {
  const V3 diff = px.tp - px.p;
  // angle_between_normals(&p, &n) >= max_normal_angle on the POINT p; NaN (|p.n| > 1) passes (image_icp.rs:118-123)
  const float pn = dot(px.p, px.tn);
  // (bitwise on purpose: the short-circuit form compiles to three nested exec-mask branches per pixel)
  px.live = px.live & (ZMASK ? px.tp.z != 0.0f : px.tmask == 1)    // RangeImage::get_point: mask == 1 (structure.rs:176)
            & !(norm_squared(diff) > gt.max_distance_sqr)        // image_icp.rs:114
            & !((pn >= -1.0f) & (pn <= gt.dot_reject_max));
  MapPx m;
  m.ui = f32_as_usize(px.u), m.vi = f32_as_usize(px.v);
  const uint32_t o = px.live ? texel_offset(m.vi, mw, m.ui) : 0u;  // (rejected pixels read texel 0)
  const f32x2 a = ld<f32x2_u>(d.imap, o), b = ld<f32x2_u>(d.imap, o + mw * 4u);
  m.t00 = a.x, m.t10 = a.y, m.t01 = b.x, m.t11 = b.y;
  return m;
}

struct Terms {  // stage D: the two residuals and Jacobians of a live pixel
  float Jg[6], rg;
  float Jc[6], rc;
  bool color;
};
// EXACT = true (the cross-check kernel behind a3d_image_icp_accumulate_exact, never the product path): the Jacobians
// and the geometric residual in the reference's own operations too — no fused multiply-adds, IEEE divisions — so
// that every per-pixel value is the oracle's bit for bit and only the order of the sums differs.
template <bool EXACT = false>
__device__ __forceinline__ Terms stage_d(const LevelDesc& d, const Gates& gt, const ProjPx& px, const MapPx& m,
                                         uint8_t intensity, uint32_t mw) {
  // What decides whether a pixel counts — the transform, the projection, the gates above, and here the colour
  // residual with its gate — is the reference's arithmetic operation for operation (no fused multiply-adds, IEEE
  // division).  What follows a passed gate only feeds the sums: the two Jacobians and the geometric residual may use
  // a*b+c in one rounding and a refined reciprocal instead of a division (each term within 1-3 ulp of the
  // reference's; the sums are compared with an f64 oracle at 1e-6, where the order of summation already costs more).
  auto fms = [](float a, float b, float c, float e) { return __builtin_fmaf(a, b, -(c * e)); };  // a b - c e
  Terms t;
  const V3 P = px.p, n = px.tn;
  {  // PointPlaneDistance::jacobian (src/icp/cost_function.rs:33-41)
    const V3 df = px.tp - P;
    t.Jg[0] = n.x, t.Jg[1] = n.y, t.Jg[2] = n.z;
    if (EXACT) {
      t.rg = dot(df, n);
      const V3 tw = cross(P, n);
      t.Jg[3] = tw.x, t.Jg[4] = tw.y, t.Jg[5] = tw.z;
    } else {
      t.rg = __builtin_fmaf(df.z, n.z, __builtin_fmaf(df.y, n.y, df.x * n.x));
      t.Jg[3] = fms(P.y, n.z, P.z, n.y), t.Jg[4] = fms(P.z, n.x, P.x, n.z), t.Jg[5] = fms(P.x, n.y, P.y, n.x);
    }
  }
  // IntensityMap::bilinear_grad (src/intensity_map.rs:184-210), H = 0.005
  const float uf = px.u - (float)m.ui, vf = px.v - (float)m.vi;
  const float value = bilerp(m.t00, m.t10, m.t01, m.t11, uf, vf);  // exact: the colour gate reads it
  const float Hh = 0.005f, H_INV = 1.0f / 0.005f;
  const float u2 = px.u + Hh, v2 = px.v + Hh;
  // the shifted samples share the cell except within 0.005 of a texel boundary: those (rare) lanes resample
  // from memory, behind ONE wave-uniform branch so that the common path has no per-lane branches
  // (uh, vh like `value`, operation for operation: du = (uh - value) / H amplifies a last-bit difference between the
  // two by value / (H * slope), three to four decimal orders)
  float uh = bilerp(m.t00, m.t10, m.t01, m.t11, u2 - (float)m.ui, vf);
  float vh = bilerp(m.t00, m.t10, m.t01, m.t11, uf, v2 - (float)m.vi);
  const bool u_leaves = f32_as_usize(u2) != m.ui, v_leaves = f32_as_usize(v2) != m.vi;
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(u_leaves | v_leaves) != 0ull, 0)) {
    if (u_leaves) uh = bilinear_at(d.imap, mw, u2, px.v);
    if (v_leaves) vh = bilinear_at(d.imap, mw, px.u, v2);
  }
  const float du = (uh - value) * H_INV;
  const float dv = (vh - value) * H_INV;
  const float sc = (float)intensity * 0.003921569f;  // image_icp.rs:131 (u8 -> f32 is exact)
  // CameraIntrinsics::project_grad (src/camera.rs:82-89): fx / z, -x fx / zz, fy / z, -y fy / zz through the
  // refined reciprocal of z (rcp + one Newton step)
  const float z = P.z;
  t.rc = sc - value;
  t.color = t.rc * t.rc <= gt.max_color_distance_sqr;  // image_icp.rs:136
  if (EXACT) {
    const float zz = z * z;
    const float dfx = d.fx / z, dcx = -P.x * d.fx / zz, dfy = d.fy / z, dcy = -P.y * d.fy / zz;
    const V3 gr{du * dfx, dv * dfy, du * dcx + dv * dcy};
    const V3 tw = cross(P, gr);
    t.Jc[0] = gr.x, t.Jc[1] = gr.y, t.Jc[2] = gr.z, t.Jc[3] = tw.x, t.Jc[4] = tw.y, t.Jc[5] = tw.z;
    return t;
  }
  const float r0 = __builtin_amdgcn_rcpf(z);
  const float rz = __builtin_fmaf(__builtin_fmaf(-z, r0, 1.0f), r0, r0), rzz = rz * rz;
  const float dfx = d.fx * rz, dfy = d.fy * rz, dcx = (-P.x * d.fx) * rzz, dcy = (-P.y * d.fy) * rzz;
  const V3 gr{du * dfx, dv * dfy, __builtin_fmaf(du, dcx, dv * dcy)};
  t.Jc[0] = gr.x, t.Jc[1] = gr.y, t.Jc[2] = gr.z;
  t.Jc[3] = fms(P.y, gr.z, P.z, gr.y), t.Jc[4] = fms(P.z, gr.x, P.x, gr.z), t.Jc[5] = fms(P.x, gr.y, P.y, gr.x);
  return t;
}

#ifdef A3D_DIAGNOSTICS  // the round-2 last-block kernel (all its variants) and the exact-arithmetic cross-check kernel
// grid = (tiles, pairs); block = 256.  A thread visits PPT source pixels, 256 apart (coalesced), through a
// three-deep software pipeline: the source record of pixel k+2, the target gathers of pixel k+1 and the map cell of
// pixel k are in flight while pixel k is accumulated (58 per-thread f32 accumulators, wave reduce-scatter at the
// end).  G = 1 (the default) unrolls the pipeline by two with the buffers swapping roles; G = 2 / 4 handle G pixels
// per step (tuning options).  Reading the loop's s_waitcnt's in the ISA is part of maintaining this kernel: a
// short-circuit `&&` around a load, a u8 -> f32 conversion next to its load or a 64-bit multiply-add with a
// don't-care high half each cost a full drain of the pipeline per pixel before they were found.
//
// MERGED: the thread accumulates geom.add_weighted(color, w, cw) directly (31 sums from the weighted Jacobians,
// icp_engine.hpp) instead of the two 29-entry systems: same FMAs, 27 fewer live registers, one more wave per SIMD.
template <int G, bool MERGED>
__global__ void __launch_bounds__(256, MERGED ? 5 : 1)
    image_icp_kernel(const LevelDesc* __restrict__ descs, JobState* __restrict__ states, Gates gt,
                     float* __restrict__ partials, unsigned* __restrict__ counters, SolveArgs solve, int PPT) {
  const int pair = blockIdx.y;
  const uint32_t tile = blockIdx.x;
  constexpr int NACC = MERGED ? GN_MERGED : GN_PARTIAL;
  float acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;
  const float wg = solve.weight, wc = solve.color_weight;
  // one live pixel into the accumulators (the geometric term counts even when the colour term is rejected)
  auto accumulate = [&](const Terms& t) {
    if (MERGED) {
      float Jg[6], Jc[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Jg[k] = wg * t.Jg[k], Jc[k] = t.color ? wc * t.Jc[k] : 0.0f;
      gn_step_merged(acc, t.rg, Jg, t.color ? t.rc : 0.0f, Jc, t.color ? 1.0f : 0.0f);
    } else {
      gn_step(acc, t.rg, t.Jg);
      if (t.color) gn_step(acc + (MERGED ? 0 : GN_ACC), t.rc, t.Jc);
    }
  };
  JobState* st = &states[pair];
#ifdef A3D_TAIL_STAMPS
  if (threadIdx.x == 0 && pair == 0 && blockIdx.x == 0) g_tail_stamps[8] = __builtin_amdgcn_s_memrealtime();
#endif
  if (st->status == A3D_OK) {
    const LevelDesc d = descs[pair];
    const Pose T = st->pose;
    const uint32_t mw = d.tw + 2;
    const float twf = (float)d.tw, thf = (float)d.th;
    const uint32_t base = tile * (256u * (uint32_t)PPT) + threadIdx.x;
    // Software pipeline, three batches deep: while batch k is being accumulated (stage D) the target
    // gathers of batch k+1 and the source records of batch k+2 are in flight, so the L1 miss queue of the
    // CU stays occupied during the arithmetic.
    auto src_at = [&](int k0, int g) {
      const uint32_t i = base + (uint32_t)(k0 + g) * 256u;
      return stage_a(d, i, (k0 < PPT) && (i < d.src_n));
    };
    if (G == 1) {
      // Two pipeline steps per trip with the buffers swapping roles, so that the rotation cur <- nxt, s1 <- s2
      // costs no register copies (a rolled loop spent ~30 v_mov per pixel on it; the kernel is VALU-issue bound).
      SrcPx sa, sb;
      ProjPx pa, pb;
      uint8_t ia, ib;
      {
        const SrcPx s0 = src_at(0, 0);
        sa = src_at(1, 0);
        pa = stage_b(d, T, s0, twf, thf);
        ia = s0.intensity;
      }
      // one step: `cur` holds the projected pixel k, `s_next` the source record of pixel k+1 (consumed here);
      // leaves the projected pixel k+1 in `nxt` and the source record of pixel k+2 in `s_new`
      auto step = [&](ProjPx& cur, uint8_t cur_i, ProjPx& nxt, uint8_t& nxt_i, const SrcPx& s_next, SrcPx& s_new, int k0) {
        s_new = src_at(k0 + 2, 0);                                  // issue source record k+2
        const MapPx mp = stage_c(d, gt, cur, mw);                   // gathers(k) land; issue map cell(k)
        nxt = stage_b(d, T, s_next, twf, thf);                      // issue gathers(k+1)
        nxt_i = s_next.intensity;
        if (cur.live) accumulate(stage_d(d, gt, cur, mp, cur_i, mw));
      };
#pragma unroll 1
      for (int k0 = 0; k0 < PPT; k0 += 2) {  // PPT is even (batch_commit_descs); surplus pixels are out of range
        step(pa, ia, pb, ib, sa, sb, k0);
        step(pb, ib, pa, ia, sb, sa, k0 + 1);
      }
    } else {
    SrcPx s1[G];
    ProjPx cur[G];
    uint8_t cur_int[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const SrcPx s0 = src_at(0, g);
      s1[g] = src_at(G, g);
      cur[g] = stage_b(d, T, s0, twf, thf);
      cur_int[g] = s0.intensity;
    }
#pragma unroll 1
    for (int k0 = 0; k0 < PPT; k0 += G) {
      SrcPx s2[G];
      MapPx mp[G];
      ProjPx nxt[G];
      uint8_t nxt_int[G];
#pragma unroll
      for (int g = 0; g < G; ++g) s2[g] = src_at(k0 + 2 * G, g);  // issue source records of batch k+2
#pragma unroll
      for (int g = 0; g < G; ++g) mp[g] = stage_c(d, gt, cur[g], mw);  // gathers(k) land; issue map cells(k)
#pragma unroll
      for (int g = 0; g < G; ++g) {  // source(k+1) landed long ago; issue gathers(k+1)
        nxt[g] = stage_b(d, T, s1[g], twf, thf);
        nxt_int[g] = s1[g].intensity;
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (cur[g].live) accumulate(stage_d(d, gt, cur[g], mp[g], cur_int[g], mw));
      }
#pragma unroll
      for (int g = 0; g < G; ++g) cur[g] = nxt[g], cur_int[g] = nxt_int[g], s1[g] = s2[g];
    }
    }  // G != 1
  }
#ifdef A3D_TAIL_STAMPS
  if (threadIdx.x == 0 && pair == 0 && blockIdx.x == 0) g_tail_stamps[9] = __builtin_amdgcn_s_memrealtime();
#endif
  // a failed job stays frozen: its blocks contribute nothing and nobody runs its solve
  SolveArgs sa = solve;
  if (st->status != A3D_OK) sa.mode = SOLVE_NONE;
  block_finish<NACC>(acc, partials + (size_t)pair * gridDim.x * GN_PARTIAL, tile, gridDim.x,
                     counters + pair, st, sa, pair);
}

// Cross-check kernel (a3d_image_icp_accumulate_exact): the same four stages with stage_d<EXACT>, one pixel at a time
// and no software pipeline; the per-sample products of the accumulation are rounded separately (`a * b` then `+`),
// like GaussNewton::step.  Only the order of the additions differs from the reference.
__global__ void __launch_bounds__(256)
    image_icp_exact_kernel(const LevelDesc* __restrict__ descs, const JobState* __restrict__ states, Gates gt,
                           float* __restrict__ partials, int PPT) {
  float acc[GN_PARTIAL];
#pragma unroll
  for (int k = 0; k < GN_PARTIAL; ++k) acc[k] = 0.0f;
  const LevelDesc d = descs[0];
  const Pose T = states[0].pose;
  const uint32_t mw = d.tw + 2;
  const float twf = (float)d.tw, thf = (float)d.th;
  auto step_exact = [](float* a, float r, const float J[6]) {  // gaussnewton.rs:47-77, no fused multiply-adds
    int t = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) {
        const float m = J[i] * J[j];
        a[t] = a[t] + m;
        ++t;
      }
    for (int i = 0; i < 6; ++i) {
      const float m = J[i] * r;
      a[21 + i] = a[21 + i] + m;
    }
    const float rr = r * r;
    a[27] = a[27] + rr;
    a[28] += 1.0f;
  };
  for (int k0 = 0; k0 < PPT; ++k0) {
    const uint32_t i = blockIdx.x * (256u * (uint32_t)PPT) + threadIdx.x + (uint32_t)k0 * 256u;
    const SrcPx s0 = stage_a(d, i, i < d.src_n);
    ProjPx px = stage_b(d, T, s0, twf, thf);
    const MapPx mp = stage_c(d, gt, px, mw);
    if (px.live) {
      const Terms t = stage_d<true>(d, gt, px, mp, s0.intensity, mw);
      step_exact(acc, t.rg, t.Jg);
      if (t.color) step_exact(acc + GN_ACC, t.rc, t.Jc);
    }
  }
  block_reduce_store<GN_PARTIAL, false>(acc, partials + (size_t)blockIdx.x * GN_PARTIAL);
}

#endif  // A3D_DIAGNOSTICS

// ---- the pixel pass of one block (the reference's loop over its share of the source pixels) -----------------------
// grid = (tiles, pairs); block = 256.  A thread visits `ppt` source pixels, 256 apart (coalesced), through a
// three-deep software pipeline: the source record of pixel k+2, the target gathers of pixel k+1 and the map cell of
// pixel k are in flight while pixel k is accumulated (58 per-thread f32 accumulators, wave reduce-scatter at the end).
// The pipeline is unrolled by two with the buffers swapping roles, so that the rotation cur <- nxt, s1 <- s2 costs no
// register copies.  Reading the loop's s_waitcnt's in the ISA is part of maintaining this code: a short-circuit `&&`
// around a load, a u8 -> f32 conversion next to its load or a 64-bit multiply-add with a don't-care high half each
// cost a full drain of the pipeline per pixel before they were found.
// The first two source records do not depend on the pose: the callers issue them (pixel_source_at) BEFORE they wait
// for the pose of the iteration, so those loads are in flight during the head.
template <bool ZMASK>
__device__ __forceinline__ SrcPx pixel_source_at(const LevelDesc& d, uint32_t base, int ppt, int k0) {
  const uint32_t i = base + (uint32_t)k0 * 256u;
  return stage_a<ZMASK>(d, i, (k0 < ppt) && (i < d.src_n));
}
template <bool ZMASK>
__device__ __forceinline__ void pixel_pass(const LevelDesc& d, const Gates& gt, const Pose& T, uint32_t base, int ppt,
                                           const SrcPx& s0, SrcPx sa, float (&acc)[GN_PARTIAL]) {
  const uint32_t mw = d.tw + 2;
  const float twf = (float)d.tw, thf = (float)d.th;
  SrcPx sb;
  ProjPx pa = stage_b<ZMASK>(d, T, s0, twf, thf), pb;
  uint8_t ia = s0.intensity, ib;
  // one step: `cur` holds the projected pixel k, `s_next` the source record of pixel k+1 (consumed here); leaves the
  // projected pixel k+1 in `nxt` and the source record of pixel k+2 in `s_new`
  auto step = [&](ProjPx& cur, uint8_t cur_i, ProjPx& nxt, uint8_t& nxt_i, const SrcPx& s_next, SrcPx& s_new, int k0) {
    s_new = pixel_source_at<ZMASK>(d, base, ppt, k0 + 2);  // issue source record k+2
    const MapPx mp = stage_c<ZMASK>(d, gt, cur, mw);        // gathers(k) land; issue map cell(k)
    nxt = stage_b<ZMASK>(d, T, s_next, twf, thf);           // issue gathers(k+1)
    nxt_i = s_next.intensity;
    if (cur.live) {
      const Terms t = stage_d(d, gt, cur, mp, cur_i, mw);
      gn_step(acc, t.rg, t.Jg);  // the geometric term counts even when the colour term is rejected
      if (t.color) gn_step(acc + GN_ACC, t.rc, t.Jc);
    }
  };
#pragma unroll 1
  for (int k0 = 0; k0 < ppt; k0 += 2) {  // ppt is even (plan_tiling); surplus pixels are out of range
    step(pa, ia, pb, ib, sa, sb, k0);
    step(pb, ib, pa, ia, sb, sa, k0 + 1);
  }
}

// ---- one launch per iteration: head-solve hand-off (icp_engine.hpp, "ticketless hand-off") -----------------------
// The launch of iteration k first finishes iteration k - 1: every block sums its pair's partials of the previous
// launch and runs the solve (head_advance), then takes the pixel pass with the resulting pose and stores its own
// partial with plain stores.  State and partials alternate between two buffers (in / out).  A pair whose own tile
// count is below the grid's (a smaller image in a mixed batch) leaves the surplus blocks to store zero partials.
template <bool ZMASK>
__global__ void __launch_bounds__(256, 1)
    image_icp_head_kernel(const LevelDesc* __restrict__ descs, const JobState* __restrict__ states_in,
                          JobState* __restrict__ states_out, Gates gt, const float* __restrict__ partials_in,
                          float* __restrict__ partials_out, uint32_t job_stride, HeadArgs head) {
  __shared__ uint32_t s_state[JOB_WORDS];
  const int pair = blockIdx.y;
  const uint32_t tile = blockIdx.x;
  float acc[GN_PARTIAL];
#pragma unroll
  for (int k = 0; k < GN_PARTIAL; ++k) acc[k] = 0.0f;
#if defined(A3D_DIAGNOSTICS) && defined(A3D_TAIL_STAMPS)
  // s_memrealtime stamps (100 MHz) of the pair-0 block that owns the LAST tile (scripts/head_stamps.py): where one
  // iteration's launch spends its time.  Slot base alternates with the launch so that two consecutive launches survive.
  const bool stamping = threadIdx.x == 0 && pair == 0 && tile + 1 == gridDim.x;
  unsigned long long* stamp = g_tail_stamps + 16 + 8 * ((states_out > states_in) ? 1 : 0);
#define A3D_HEAD_STAMP(k) do { if (stamping) stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define A3D_HEAD_STAMP(k) do { } while (0)
#endif
  A3D_HEAD_STAMP(0);  // kernel entry
  const LevelDesc d = descs[pair];
  const int ppt = (int)d.ppt;
  const uint32_t base = tile * (256u * (uint32_t)ppt) + threadIdx.x;
  const SrcPx s0 = pixel_source_at<ZMASK>(d, base, ppt, 0), s1 = pixel_source_at<ZMASK>(d, base, ppt, 1);
  A3D_HEAD_STAMP(1);  // descriptor loaded, first source pixels requested
  head_advance(states_in + pair, tile == 0 ? states_out + pair : nullptr, partials_in + (size_t)pair * job_stride, head,
               pair, s_state, tile == 0);
  A3D_HEAD_STAMP(2);  // previous iteration finished: partials summed, solve, pose update
  if ((int)s_state[15] == A3D_OK) {  // a failed job stays frozen: its blocks contribute nothing
    auto uni = [&](int k) { return __uint_as_float(__builtin_amdgcn_readfirstlane(s_state[k])); };
    const Pose T{{uni(0), uni(1), uni(2)}, {uni(3), uni(4), uni(5), uni(6)}};
    pixel_pass<ZMASK>(d, gt, T, base, ppt, s0, s1, acc);
  }
  A3D_HEAD_STAMP(3);  // pixel pass done
  block_reduce_store<GN_PARTIAL, false>(acc, partials_out + (size_t)pair * job_stride + (size_t)tile * GN_PARTIAL);
  A3D_HEAD_STAMP(4);  // block partial stored
#undef A3D_HEAD_STAMP
}

#ifdef A3D_DIAGNOSTICS
// ---- many iterations in one launch: the persistent head-solve kernel (round 4; measured SLOWER than kernel
// boundaries on MI355X — DESIGN.md, ruled out — and therefore only in the diagnostics build: A3D_ICP_PERSIST=mask) ----
// What separates two dependent iterations (a 1.5 us kernel boundary + the ~4 us head that finishes the previous one) is more than
// the pixel pass of a coarse level or of a lone pair.  Here the blocks of a pair stay resident and run a whole schedule
// of (level, iterations) entries.  The hand-off between two iterations is ONE hop: a block stores its partials
// write-through, drains, and adds their number to the pair's counter; every block of the pair waits until the counter
// shows that all partials of the iteration are out, then sums them itself (same fixed order as head_advance: same
// bits) and runs the solve redundantly — there is no last block, no published state and no second hop back (the
// round-1/2 level kernel had both and lost to kernel boundaries).  The state of the pair lives in every block's LDS and
// never touches memory until one block stores it at the end; the partials alternate between the same two buffers as
// the per-iteration launches use, so a launch of image_icp_head_kernel can continue where this kernel stops (and
// does, for level 0 of a batch).  Pairs never wait for each other.
//   * forward progress without assuming that the whole grid is resident: the blocks of a pair REGISTER when they start
//     (one atomic add on the pair's roster word, which also hands out ranks).  The roster closes when all blocks of the
//     pair have arrived — at once on a GPU the process has to itself — or when a registered block has waited
//     ROSTER_WAIT_TICKS for the rest (the chip is shared and some blocks have no slot yet): the R blocks registered by
//     then share the pair's tiles (rank r takes tiles r, r + R, ...), a block that arrives later leaves at once.
//     Everybody a block ever waits for is therefore running.  Which block computes a tile does not change the tile's
//     partial, and the partials are summed in tile order: the result does not depend on R.
//   * control words per pair (PAIR_CTRL_WORDS, 256 bytes apart so that the pairs' polls spread over the memory
//     channels): [0] counter, monotonic — `counter_base` = its value when the launch starts (the host advances it by the
//     schedule's total, so nothing is reset between launches); [1 + parity] roster (count | closed bit), [3 + parity]
//     R, published by whoever closed the roster.  Launches alternate the parity; rank 0 of a launch zeroes the other
//     parity's words for the next one.
//   * a pair's own tile count below the level's (a smaller image in a mixed batch): the surplus tiles are zero partials.
struct PersistLevel {
  uint32_t desc_base;   // descs[desc_base + pair] describes this level
  uint32_t iterations;
  uint32_t tiles;       // partials per pair and iteration at this level (<= gridDim.x)
  float weight, color_weight;
  Gates gates;
};
constexpr int PERSIST_MAX_LEVELS = 16;
constexpr uint32_t PAIR_CTRL_WORDS = 64;  // 256 bytes per pair
constexpr uint32_t ROSTER_CLOSED = 0x80000000u;
// s_memrealtime ticks (100 MHz).  The roster wait only matters on a shared GPU; the spin bound is a safety net (a
// registered partner is running by construction, so only a fault elsewhere can make it this late).
constexpr unsigned long long ROSTER_WAIT_TICKS = 2000ull, PERSIST_SPIN_BOUND_TICKS = 200'000'000ull;
struct PersistPlan {
  uint32_t n_levels;       // schedule entries, in execution order (coarsest level first)
  uint32_t seq0;           // launches / iterations that ran before this kernel (selects the buffer parity)
  uint32_t counter_base;
  uint32_t finish;         // 1: apply the last iteration too and write the outputs (no job_finish launch needed)
  uint32_t parity;         // which roster words this launch uses
  int trace_index0;
  int trace_stride;
  HeadArgs prev;           // the iteration that ran just before this kernel (mode SOLVE_NONE: none)
  PersistLevel lv[PERSIST_MAX_LEVELS];
};
struct PersistOut {  // written by rank 0 when plan.finish (each nullable)
  Pose* poses;
  int32_t* status;
  float* matrices;
  unsigned long long* stamps;  // nullable: [pair][n_levels][2] s_memrealtime at the start / end of each level (rank 0)
};

template <bool ZMASK>
__global__ void __launch_bounds__(256, 4)  // four blocks per CU: what plan_tiling counts on for a 64-pair batch
    image_icp_persistent_kernel(const LevelDesc* __restrict__ descs, uint32_t n_pairs_total, JobState* __restrict__ states,
                                float* __restrict__ partials, uint32_t partials_half, uint32_t job_stride,
                                unsigned* __restrict__ ctrl, PersistPlan plan, PersistOut out) {
  __shared__ uint32_t s_state[JOB_WORDS];
  __shared__ uint32_t s_rank, s_members;
  __shared__ int s_go;
  const int pair = blockIdx.y;
  unsigned* const counter = ctrl + (size_t)pair * PAIR_CTRL_WORDS;
  unsigned* const roster = counter + 1 + plan.parity;
  unsigned* const members = counter + 3 + plan.parity;
  float* const job_partials = partials + (size_t)pair * job_stride;
  // ---- register; learn the rank and how many blocks share the pair ----
  if (threadIdx.x == 0) {
    const uint32_t old = __hip_atomic_fetch_add(roster, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t rank = old & 0xffffu, R = 0;
    if (old & ROSTER_CLOSED) {
      rank = 0xffffffffu;  // the pair started without this block
    } else {
      unsigned long long t0 = 0;
      for (uint32_t spin = 0;; ++spin) {
        R = __hip_atomic_load(members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (R) break;
        uint32_t cur = __hip_atomic_load(roster, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool close = !(cur & ROSTER_CLOSED) && (cur & 0xffffu) >= gridDim.x;
        if (!close && !(cur & ROSTER_CLOSED) && (spin & 15u) == 15u) {
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          if (!t0) t0 = now;
          close = now - t0 > ROSTER_WAIT_TICKS;
        }
        if (close && __hip_atomic_compare_exchange_strong(roster, &cur, cur | ROSTER_CLOSED, __ATOMIC_RELAXED,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          R = cur & 0xffffu;
          __hip_atomic_store(members, R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    s_rank = rank, s_members = R;
  }
  // the pair's state: every block carries its own copy through the schedule (lane k < 18 of wave 0 <-> word k)
  if (threadIdx.x < JOB_WORDS) s_state[threadIdx.x] = ((const uint32_t*)(states + (size_t)(plan.seq0 & 1u) * n_pairs_total + pair))[threadIdx.x];
  __syncthreads();
  const uint32_t rank = s_rank, R = s_members;
  if (rank == 0xffffffffu) return;
  HeadArgs prev = plan.prev;
  uint32_t seq = plan.seq0;
  uint32_t expected = plan.counter_base;  // the counter's value once every partial of the previous iteration is out
  int trace_index = plan.trace_index0;
  bool alive = true;
  // waits for the previous iteration's partials, sums them and advances the state in LDS
  auto head = [&](bool write_trace) {
    if (prev.mode != SOLVE_NONE) {
      if (threadIdx.x == 0) {
        int go = 1;
        unsigned long long t0 = 0;
        for (uint32_t spin = 0; (int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expected) < 0; ++spin) {
          __builtin_amdgcn_s_sleep(1);
          if ((spin & 255u) == 255u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (!t0) t0 = now;
            if (now - t0 > PERSIST_SPIN_BOUND_TICKS) {
              go = 0;
              break;
            }
          }
        }
        s_go = go;
      }
      __syncthreads();
      if (!s_go) {
        alive = false;
        return;
      }
    }
    const uint32_t state_bits = threadIdx.x < JOB_WORDS ? s_state[threadIdx.x] : 0u;
    head_sum_and_advance<true>(state_bits, job_partials + (size_t)((seq + 1u) & 1u) * partials_half, prev, pair, s_state,
                               write_trace);
  };
#pragma unroll 1
  for (uint32_t l = 0; l < plan.n_levels && alive; ++l) {
    const PersistLevel lv = plan.lv[l];
    const LevelDesc d = descs[lv.desc_base + pair];
    const int ppt = (int)d.ppt;
    const uint32_t own_tiles = (d.src_n + 256u * d.ppt - 1u) / (256u * d.ppt);
    const uint32_t mine = rank < lv.tiles ? (lv.tiles - rank + R - 1u) / R : 0u;  // tiles rank, rank + R, ...
    if (out.stamps && rank == 0 && threadIdx.x == 0) out.stamps[((size_t)pair * plan.n_levels + l) * 2] = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (uint32_t it = 0; it < lv.iterations; ++it) {
      const uint32_t base0 = rank * (256u * (uint32_t)ppt) + threadIdx.x;
      SrcPx s0{}, s1{};
      if (mine) s0 = pixel_source_at<ZMASK>(d, base0, ppt, 0), s1 = pixel_source_at<ZMASK>(d, base0, ppt, 1);
      head(rank == 0);
      if (!alive) break;
      const bool ok = (int)s_state[15] == A3D_OK;  // (a failed pair stays frozen: its blocks publish zeros)
      auto uni = [&](int k) { return __uint_as_float(__builtin_amdgcn_readfirstlane(s_state[k])); };
      const Pose T{{uni(0), uni(1), uni(2)}, {uni(3), uni(4), uni(5), uni(6)}};
#pragma unroll 1
      for (uint32_t tile = rank; tile < lv.tiles; tile += R) {
        float acc[GN_PARTIAL];  // (not live across the head: its partial loads want the registers)
#pragma unroll
        for (int k = 0; k < GN_PARTIAL; ++k) acc[k] = 0.0f;
        if (ok && tile < own_tiles) {
          const uint32_t base = tile * (256u * (uint32_t)ppt) + threadIdx.x;
          if (tile != rank) s0 = pixel_source_at<ZMASK>(d, base, ppt, 0), s1 = pixel_source_at<ZMASK>(d, base, ppt, 1);
          pixel_pass<ZMASK>(d, lv.gates, T, base, ppt, s0, s1, acc);
        }
        block_reduce_store<GN_PARTIAL, true>(acc, job_partials + (size_t)(seq & 1u) * partials_half + (size_t)tile * GN_PARTIAL);
        __syncthreads();  // (the reduction's LDS is reused by the next tile)
      }
      if (mine) {  // publish: write-through partials, drained by the storing wave, then one agent-scope add (Guideline 16, form R1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      expected += lv.tiles;
      prev.weight = lv.weight, prev.color_weight = lv.color_weight, prev.mode = SOLVE_IMAGE_ICP;
      prev.tiles = lv.tiles;
      prev.first_in_level = it == 0, prev.last_in_level = it + 1 == lv.iterations;
      prev.trace_index = trace_index++;
      ++seq;
    }
    if (out.stamps && rank == 0 && threadIdx.x == 0) out.stamps[((size_t)pair * plan.n_levels + l) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
  }
  if (alive && plan.finish) head(rank == 0);
  if (rank != 0) return;
  if (threadIdx.x == 0) {  // the other parity's roster words, for the next launch
    __hip_atomic_store(counter + 1 + (plan.parity ^ 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(counter + 3 + (plan.parity ^ 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!alive) {  // the spin bound was hit (a fault elsewhere): the pair fails loudly
    if (threadIdx.x == 0) {
      states[(size_t)(seq & 1u) * n_pairs_total + pair].status = A3D_HIP_ERROR;
      if (out.status) out.status[pair] = A3D_HIP_ERROR;
    }
    return;
  }
  // leave the state where the next launch (image_icp_head_kernel / job_finish_head) reads it
  if (threadIdx.x < JOB_WORDS) ((uint32_t*)(states + (size_t)(seq & 1u) * n_pairs_total + pair))[threadIdx.x] = s_state[threadIdx.x];
  if (plan.finish && threadIdx.x == 0) {
    const float* f = (const float*)s_state;
    const Pose p{{f[0], f[1], f[2]}, {f[3], f[4], f[5], f[6]}};
    if (out.poses) out.poses[pair] = p;
    if (out.status) out.status[pair] = (int32_t)s_state[15];
    if (out.matrices) {
      float m[16];
      pose_to_matrix(p, m);
      for (int k = 0; k < 16; ++k) out.matrices[(size_t)pair * 16 + k] = m[k];
    }
  }
}

#endif  // A3D_DIAGNOSTICS (persistent head-solve kernel)

#ifdef A3D_DIAGNOSTICS  // measured slower than the default path (ruled out): kept as cross-checks
// ---- one launch per pyramid level ------------------------------------------------------------------
// When the grid of a level is exactly the set of blocks the chip holds at once (choose_tiling, waves = 1),
// every block of every pair is resident for the whole launch, so the iterations of the level can run inside
// ONE launch: after publishing its partial a block waits for its pair's next pose instead of exiting.
//   * the pair's last block runs the solve and then publishes `epoch = epoch_base + it + 1` (state stored
//     with sc1 stores, drained, then the epoch word: CDNA guide Guideline 16, form R1);
//   * the pair's other blocks poll that word (one lane, relaxed agent-scope load, s_sleep), bounded, then read
//     the pose with sc1 loads; nothing handed off is ever read through the L1;
//   * pairs never wait for each other: there is no grid-wide barrier.
// No block can wait for one that is not resident because the host only uses this kernel when
// pairs x tiles <= resident blocks (occupancy query of THIS kernel); a spin that still exceeds its bound
// marks the job A3D_HIP_ERROR and leaves.
struct LevelPlan {
  uint32_t iterations;
  uint32_t epoch_base;  // iterations completed by the levels run before this one
  int trace_base;
};

template <int G>
__global__ void __launch_bounds__(256, 3)  // at most 168 VGPRs: three blocks per CU
    image_icp_level_kernel(const LevelDesc* __restrict__ descs, JobState* __restrict__ states, Gates gt,
                           float* __restrict__ partials, unsigned* __restrict__ counters,
                           unsigned* __restrict__ epochs, SolveArgs solve, LevelPlan plan, int PPT) {
  __shared__ float s_pose[8];
  __shared__ int s_go;
  const int pair = blockIdx.y;
  JobState* st = &states[pair];
  const LevelDesc d = descs[pair];
  const uint32_t mw = d.tw + 2;
  const float twf = (float)d.tw, thf = (float)d.th;
  const uint32_t base = blockIdx.x * (256u * (uint32_t)PPT) + threadIdx.x;
  float* job_partials = partials + (size_t)pair * gridDim.x * GN_PARTIAL;
#pragma unroll 1
  for (uint32_t it = 0; it < plan.iterations; ++it) {
    // ---- this iteration's pose: wait until the pair has completed `epoch_base + it` iterations ----------
    if (threadIdx.x == 0) {
      const unsigned want = plan.epoch_base + it;
      // a failed job stays frozen: its epoch stops advancing, so look at the status before (and while) waiting
      int go = load_status(st) == A3D_OK;
      unsigned spins = 0;
      while (go && __hip_atomic_load(&epochs[pair], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(4);
        ++spins;
        if ((spins & 63u) == 0 && load_status(st) != A3D_OK) go = 0;
        if (spins > (1u << 22)) {  // ~1 s: a resident partner can not be this late
          store_status(st, A3D_HIP_ERROR);
          go = 0;
        }
      }
      if (go && load_status(st) != A3D_OK) go = 0;
      const Pose T0 = load_pose(&st->pose);
      s_pose[0] = T0.t.x, s_pose[1] = T0.t.y, s_pose[2] = T0.t.z;
      s_pose[3] = T0.q.i, s_pose[4] = T0.q.j, s_pose[5] = T0.q.k, s_pose[6] = T0.q.w;
      s_go = go;
    }
    __syncthreads();
    if (!s_go) return;  // uniform over the block, and every block of the pair sees the same status
    // the pose is wave-uniform: keep it in scalar registers like a kernel argument
    auto uni = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    const Pose T{{uni(s_pose[0]), uni(s_pose[1]), uni(s_pose[2])},
                 {uni(s_pose[3]), uni(s_pose[4]), uni(s_pose[5]), uni(s_pose[6])}};
    float acc[GN_PARTIAL];
#pragma unroll
    for (int k = 0; k < GN_PARTIAL; ++k) acc[k] = 0.0f;
    {
      auto src_at = [&](int k0, int g) {
        const uint32_t i = base + (uint32_t)(k0 + g) * 256u;
        return stage_a(d, i, (k0 < PPT) && (i < d.src_n));
      };
      SrcPx s1[G];
      ProjPx cur[G];
      uint8_t cur_int[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const SrcPx s0 = src_at(0, g);
        s1[g] = src_at(G, g);
        cur[g] = stage_b(d, T, s0, twf, thf);
        cur_int[g] = s0.intensity;
      }
#pragma unroll 1
      for (int k0 = 0; k0 < PPT; k0 += G) {
        SrcPx s2[G];
        MapPx mp[G];
        ProjPx nxt[G];
        uint8_t nxt_int[G];
#pragma unroll
        for (int g = 0; g < G; ++g) s2[g] = src_at(k0 + 2 * G, g);
#pragma unroll
        for (int g = 0; g < G; ++g) mp[g] = stage_c(d, gt, cur[g], mw);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          nxt[g] = stage_b(d, T, s1[g], twf, thf);
          nxt_int[g] = s1[g].intensity;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (cur[g].live) {
            const Terms t = stage_d(d, gt, cur[g], mp[g], cur_int[g], mw);
            gn_step(acc, t.rg, t.Jg);
            if (t.color) gn_step(acc + GN_ACC, t.rc, t.Jc);
          }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) cur[g] = nxt[g], cur_int[g] = nxt_int[g], s1[g] = s2[g];
      }
    }
    __syncthreads();  // s_pose is rewritten next iteration; the reduction below also reuses LDS
    SolveArgs sa = solve;
    sa.first_in_level = it == 0;
    sa.last_in_level = it + 1 == plan.iterations;
    sa.trace_index = plan.trace_base + (int)it;
    float* out = job_partials + (size_t)blockIdx.x * GN_PARTIAL;
    block_reduce_store<GN_PARTIAL, true>(acc, out);
    const bool was_last = block_publish_and_finish(job_partials, gridDim.x, counters + pair, st, sa, pair);
    if (was_last && threadIdx.x == 0) {
      // the state stores above were sc1; drain them, then let the pair's other blocks go on
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&epochs[pair], plan.epoch_base + it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}

// ---- MFMA accumulation -----------------------------------------------------------------------------
// The per-pixel sums  H += J J^T, g += J r, ssq += r^2, count += 1  for the geometric and the colour
// term are all entries of X^T X, where row p of X holds pixel p's 16 "features"
//   [ Jg(6) | rg | Jc(6) | rc | live_g | live_c ].
// v_mfma_f32_16x16x4_f32 computes a 16x16 f32 tile of A B with K = 4, exact f32 fma chains; with
// A = X^T and B = X both operands are the SAME register: lane l supplies X[pixel l>>4][feature l&15]
// (CDNA guide §3).  A wave's 64 pixels therefore take 16 MFMAs, fed by a transpose through a 4 KiB LDS
// slab per wave: lane p writes its 16 features as one row, then reads X[4m + (l>>4)][l&15] for MFMA m.
// What it buys: the 58 per-thread accumulators (58 VGPRs) become two 16x16 tiles = 8 VGPRs, and the 54
// accumulate FMAs per pixel leave the VALU for the matrix pipe.  Measured on MI355X it does not beat the
// VALU path yet, so it is opt-in (A3D_ICP_ACCUM=mfma).
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

template <int G>
__global__ void __launch_bounds__(256)
    image_icp_mfma_kernel(const LevelDesc* __restrict__ descs, JobState* __restrict__ states, Gates gt,
                          float* __restrict__ partials, unsigned* __restrict__ counters, SolveArgs solve, int PPT) {
  __shared__ __attribute__((aligned(16))) float slab[4][64 * 16];  // per wave: 64 pixels x 16 features
  const int pair = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const my_slab = slab[wave];
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  JobState* st = &states[pair];
  if (st->status == A3D_OK) {
    const LevelDesc d = descs[pair];
    const Pose T = st->pose;
    const uint32_t mw = d.tw + 2;
    const float twf = (float)d.tw, thf = (float)d.th;
    const uint32_t base = blockIdx.x * (256u * (uint32_t)PPT) + threadIdx.x;
    // transposed-read offsets of this lane: row 4m + (lane >> 4), feature lane & 15
    const int rd_row0 = lane >> 4, rd_chunk = (lane & 15) >> 2, rd_word = lane & 3;
    // same three-deep software pipeline as the VALU kernel
    auto src_at = [&](int k0, int g) {
      const uint32_t i = base + (uint32_t)(k0 + g) * 256u;
      return stage_a(d, i, (k0 < PPT) && (i < d.src_n));
    };
    SrcPx s1[G];
    ProjPx cur[G];
    uint8_t cur_int[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const SrcPx s0 = src_at(0, g);
      s1[g] = src_at(G, g);
      cur[g] = stage_b(d, T, s0, twf, thf);
      cur_int[g] = s0.intensity;
    }
#pragma unroll 1
    for (int k0 = 0; k0 < PPT; k0 += G) {
      SrcPx s2[G];
      MapPx mp[G];
      ProjPx nxt[G];
      uint8_t nxt_int[G];
#pragma unroll
      for (int g = 0; g < G; ++g) s2[g] = src_at(k0 + 2 * G, g);
#pragma unroll
      for (int g = 0; g < G; ++g) mp[g] = stage_c(d, gt, cur[g], mw);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        nxt[g] = stage_b(d, T, s1[g], twf, thf);
        nxt_int[g] = s1[g].intensity;
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        // ---- features of this lane's pixel (all zero when it is gated out) ----------------------
        f32x4_t f0 = {0.f, 0.f, 0.f, 0.f}, f1 = f0, f2 = f0, f3 = f0;
        if (cur[g].live) {
          const Terms t = stage_d(d, gt, cur[g], mp[g], cur_int[g], mw);
          f0 = f32x4_t{t.Jg[0], t.Jg[1], t.Jg[2], t.Jg[3]};
          f1.x = t.Jg[4], f1.y = t.Jg[5], f1.z = t.rg;
          f3.z = 1.0f;
          if (t.color) {
            f1.w = t.Jc[0];
            f2 = f32x4_t{t.Jc[1], t.Jc[2], t.Jc[3], t.Jc[4]};
            f3.x = t.Jc[5], f3.y = t.rc, f3.w = 1.0f;
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
#pragma unroll
      for (int g = 0; g < G; ++g) cur[g] = nxt[g], cur_int[g] = nxt_int[g], s1[g] = s2[g];
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

#endif  // A3D_DIAGNOSTICS

// Device self-test of the shared-reciprocal division: counts pairs for which it differs from `/`.
__global__ void division_selftest_kernel(const float* __restrict__ num, const float* __restrict__ den, uint32_t n,
                                         unsigned* __restrict__ mismatches) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = num[i], z = den[i];
  if (!(div_den_ok(z) && div_num_ok(a))) return;
  const float q = div_by(a, div_prepare(z)), want = a / z;
  if (__float_as_uint(q) != __float_as_uint(want)) atomicAdd(mismatches, 1u);
}

}  // namespace

// Control words per pair of the diagnostics build's in-launch hand-offs (persistent kernel: counter + roster; the
// last-block forms use the first word of the array as a plain per-pair ticket counter).  The product has none.
#ifdef A3D_DIAGNOSTICS
constexpr uint32_t CTRL_WORDS_PER_PAIR = PAIR_CTRL_WORDS;
#else
constexpr uint32_t CTRL_WORDS_PER_PAIR = 0;
#endif

// P independent coarse-to-fine alignments.  Owns only small state; the images are borrowed.
struct a3d_multiscale_batch {
  a3d_context* ctx = nullptr;
  uint32_t n_pairs = 0, n_levels = 0;
  std::vector<a3d_icp_params> params;  // index 0 = finest
  std::vector<Gates> gates;
  // per level: blocks per pair of the grid (the largest of the pairs' own tile counts), pixels per thread of the
  // largest pair (each pair's own value is in its descriptor), pixels per pipeline step (1; diagnostics builds: 2, 4)
  std::vector<uint32_t> tiles, ppt, group;
  std::vector<LevelDesc> h_descs;      // [level][pair]
  void* d_block = nullptr;  // one allocation behind every small device array below
  LevelDesc* d_descs = nullptr;
  JobState* d_states = nullptr;    // [2][P]: the launches alternate between the two (icp_engine.hpp, head-solve hand-off)
  hipEvent_t level_ready[16] = {};  // a3d_multiscale_align_host: the launches of level l wait for this event (its upload)
  float* d_partials = nullptr;     // [2][P][max_tiles][58]
  size_t partials_capacity = 0;  // floats
  size_t partials_half = 0;      // floats per buffer
  uint32_t max_tiles = 1;        // largest tiles[level]: the per-pair stride of d_partials
  // Persistent head-solve kernel: the levels in `persist_mask` run as ONE launch per stream group (image_icp.hip,
  // image_icp_persistent_kernel); d_counters holds PAIR_CTRL_WORDS control words per pair (counter, roster): the
  // counter counts the partials the pair has published since the batch was created, counter_base is what every
  // pair's counter shows between two alignments, persist_parity which roster words the next launch uses.
  unsigned* d_counters = nullptr;
  uint32_t counter_base = 0;
  uint32_t persist_parity = 0;
  uint32_t persist_mask = 0;
  uint32_t persist_resident_blocks = 0;  // blocks of the persistent kernel the chip holds at once
  bool persist_disabled = false;  // a launch hit its spin bound once (the GPU is shared): per-iteration launches from now on
  bool persist_used = false;      // by the most recent enqueue
  std::vector<uint32_t> persist_levels;  // the levels its persistent launch ran, in execution order
  uint32_t last_groups = 1;              // stream groups of the most recent enqueue
  unsigned long long* d_stamps = nullptr;  // [P][2 L] level start / end stamps of the persistent kernel, when profiling
  Pose* d_poses = nullptr;
  Pose* d_init = nullptr;  // per-pair initial transforms when the caller supplies them
  int32_t* d_status = nullptr;
  double* d_readback = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> kev;  // per pixel-kernel launch: start/stop pairs, when profiling
  std::vector<uint8_t> kev_level;  // the pyramid level of each launch (the finest one of a persistent launch)
  float last_level_ms[16] = {0};
  uint32_t last_level_launches[16] = {0};
  bool profile_kernels = false;
  uint32_t resident_blocks = 1024;  // blocks of the per-iteration kernel the chip holds at once
  // every image of the batch was built on the device with masks that equal (z != 0): the kernels skip the two mask
  // bytes per pixel
  bool zmask = false;
  // Pair groups launched on separate streams: one group's launch ramp and head overlap the other groups' streaming
  // (pairs are independent, so the groups never synchronise until the final read-out).
  uint32_t n_streams = 1;
  hipStream_t aux_streams[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  float last_total_ms = 0.f, last_kernel_ms = 0.f;
  uint64_t last_kernel_launches = 0;
  // what the most recent enqueue was asked for (a rerun after a spin-bound exit repeats it)
  const Pose* last_init = nullptr;
  uint32_t last_levels = 0;
  float* last_matrices = nullptr;
  // "every launch this batch has enqueued so far": registered with the arenas of the images the batch reads, so
  // that an image freed after an enqueue-only align (no host synchronisation) is not recycled under the kernels
  std::shared_ptr<UseFence> fence = std::make_shared<UseFence>();
  std::shared_ptr<UseFence> descs_uploaded = std::make_shared<UseFence>();  // h_descs -> d_descs copy of the last rebind
#ifdef A3D_DIAGNOSTICS
  unsigned* d_epochs = nullptr;    // per pair: iterations completed (level kernel hand-off word)
  bool use_level_kernel = false;   // one launch per level, last-block form (A3D_ICP_PERSISTENT)
  uint32_t level_mask = 0;         // bit l: level l runs as ONE last-block-form launch per stream group
  uint32_t level_resident_blocks = 0;
  bool use_mfma = false, merged_accumulators = false;  // A3D_ICP_ACCUM
  bool head_solve = true;          // A3D_ICP_HANDOFF=ticket selects the last-block form
  bool exact_solve = false;        // A3D_ICP_SOLVE=exact
#endif

  ~a3d_multiscale_batch() {
    fence->retire();
    hipFree(d_block);  // descs, states, counters, epochs, poses, init, status, readback
    hipFree(d_partials);
    hipFree(d_stamps);
    if (ev0) hipEventDestroy(ev0);
    if (ev1) hipEventDestroy(ev1);
    for (auto e : kev) hipEventDestroy(e);
    for (int i = 0; i < 3; ++i) {
      if (ev_join[i]) hipEventDestroy(ev_join[i]);
    }
    if (ev_fork) hipEventDestroy(ev_fork);
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
  d->imap = target->imap;
  d->src_points = source->points, d->src_mask = source->mask, d->src_intensities = source->intensities;
  d->tgt_points = target->points, d->tgt_normals = target->normals, d->tgt_mask = target->mask;
  d->src_n = source->width * source->height;
  d->tw = target->width;
  d->th = target->height;
  d->fx = target->fx, d->fy = target->fy, d->cx = target->cx, d->cy = target->cy;
  d->flags = (target->mask_is_z && source->mask_is_z) ? 1u : 0u;
  d->ppt = 2, d->pad = 0;
  return A3D_OK;
}

Gates make_gates(const a3d_icp_params& p) {
  Gates g;
  g.max_distance_sqr = p.max_distance * p.max_distance;
  g.max_color_distance_sqr = p.max_color_distance * p.max_color_distance;
  g.dot_reject_max = acos_gate_threshold(p.max_normal_angle, /*strict=*/false);
  return g;
}

// Pixels per thread (even: the pipeline takes two steps per trip) for `n` pixels cut into at most `want` blocks, and
// the number of blocks that makes.
void tiling_for(uint32_t n, uint32_t want, uint32_t group, uint32_t* tiles, uint32_t* ppt) {
  const uint32_t step = std::max(2u, group);
  want = std::max(1u, std::min(want, std::max(1u, (n + 256 * step - 1) / (256 * step))));
  uint32_t p = (n + 256 * want - 1) / (256 * want);
  p = std::max(step, ((p + step - 1) / step) * step);
  *ppt = p;
  *tiles = std::max(1u, (n + 256 * p - 1) / (256 * p));
}

// Which levels of a batch run inside the persistent kernel, and how every (pair, level) is cut into blocks.
//  * throughput tiling (default): the block count follows the batch — a per-iteration launch fills the chip with whole
//    rounds of blocks (1.5 rounds over three stream groups measured best), a persistent launch fits all
//    its blocks on the chip at once; a handful of pairs is latency-bound and takes fewer, fatter blocks.
//  * pinned tiling (a3d_context_set_tiling): `tiles_per_pair` blocks for every pair and level whatever the batch, each
//    pair from its OWN size — the association of a pair's sums, hence every bit of its pose, no longer depends on the
//    batch it is in (the persistent kernel is used where the pinned grid still fits the chip: same bits either way).
void plan_tiling(a3d_multiscale_batch* b) {
  const uint32_t P = b->n_pairs, L = b->n_levels;
  const uint32_t pinned = b->ctx->tiles_per_pair;
  float waves = P >= 8 ? (b->n_streams > 2 ? 1.5f : 1.0f) : 0.25f;
  if (const char* env = A3D_DIAG_ENV("A3D_ICP_WAVES"))
    if (*env) waves = (float)atof(env);
  // levels that run inside the persistent kernel: none by default (it lost to kernel boundaries: DESIGN.md, ruled
  // out); diagnostics build: A3D_ICP_PERSIST=mask
  uint32_t mask = 0;
  if (const char* env = A3D_DIAG_ENV("A3D_ICP_PERSIST"))
    if (*env) mask = (uint32_t)strtoul(env, nullptr, 0) & ((1u << L) - 1u);
  if (!b->persist_resident_blocks) mask = 0;
  // (a batch that fell back to per-iteration launches keeps the tiling it had: same bits before and after)
  const uint32_t persist_tiles_cap = std::max(1u, b->persist_resident_blocks / std::max(1u, P));
  for (uint32_t l = 0; l < L; ++l) {
    uint32_t max_n = 0;
    for (uint32_t p = 0; p < P; ++p) max_n = std::max(max_n, b->h_descs[(size_t)l * P + p].src_n);
    b->group[l] = 1;
    float w = waves;
#ifdef A3D_DIAGNOSTICS
    if (const char* env = getenv("A3D_ICP_GROUP")) b->group[l] = atoi(env) == 2 ? 2 : 1;
    if (const char* env = getenv("A3D_ICP_GROUP_LEVELS")) {  // per-level "g0,g1,g2"
      unsigned gl[3] = {1, 1, 1};
      sscanf(env, "%u,%u,%u", &gl[0], &gl[1], &gl[2]);
      if (l < 3) b->group[l] = (gl[l] == 2 || gl[l] == 4) ? gl[l] : 1;
    }
    if (const char* env = getenv("A3D_ICP_WAVES_LEVELS")) {  // per-level "w0,w1,w2"
      float wl[3] = {waves, waves, waves};
      sscanf(env, "%f,%f,%f", &wl[0], &wl[1], &wl[2]);
      if (l < 3) w = wl[l];
    }
#endif
    uint32_t want = std::max(1u, (uint32_t)((float)b->resident_blocks * w / (float)P + 0.5f));
    if ((mask >> l) & 1u) want = std::min(want, persist_tiles_cap);  // all blocks of the launch resident at once
    if (pinned) want = pinned;
    uint32_t tiles = 0, ppt = 0;
    tiling_for(max_n, want, b->group[l], &tiles, &ppt);
#ifdef A3D_DIAGNOSTICS
    if (const char* env = getenv("A3D_ICP_VARIANT")) {  // "ppt,g"
      unsigned ep = 0, eg = 0;
      if (sscanf(env, "%u,%u", &ep, &eg) == 2 && ep && (eg == 1 || eg == 2)) {
        b->group[l] = eg;
        ppt = ((ep + 1) / 2) * 2;
        tiles = (max_n + 256 * ppt - 1) / (256 * ppt);
      }
    }
#endif
    for (uint32_t p = 0; p < P; ++p) {
      LevelDesc& d = b->h_descs[(size_t)l * P + p];
      d.ppt = ppt;
      if (pinned) {
        // From the pair's own size; the grid is the LARGEST own cut of the batch, not the cut of the largest pair:
        // tiling_for is not monotonic in n (pixels per thread are rounded up to even, so 512x384 at level 2 is cut into
        // 24 blocks where 640x480 takes 19), and a grid sized from max_n alone left a smaller pair's last blocks
        // unlaunched.  A pair with fewer own blocks than the grid stores zero partials from the surplus ones.
        uint32_t t_own = 0;
        tiling_for(d.src_n, pinned, 1, &t_own, &d.ppt);
        tiles = p == 0 ? t_own : std::max(tiles, t_own);
      }
    }
    b->tiles[l] = tiles, b->ppt[l] = ppt;
    // a pinned grid that does not fit the chip at once runs as per-iteration launches (same bits)
    if (((mask >> l) & 1u) && (uint64_t)tiles * P > b->persist_resident_blocks) mask &= ~(1u << l);
  }
  // the persistent levels are ONE launch: the coarsest contiguous run of the mask
  uint32_t run = 0;
  for (uint32_t l = L; l-- > 0;) {
    if (!((mask >> l) & 1u)) break;
    run |= 1u << l;
  }
  b->persist_mask = b->persist_disabled ? 0u : run;
}

#ifdef A3D_DIAGNOSTICS
// Launches pairs [p0, p0 + count) of one level on stream `s`: the round-2 last-block kernel and its variants.
a3d_status launch_pixel_kernel(a3d_multiscale_batch* b, uint32_t level, const SolveArgs& solve, uint32_t p0,
                               uint32_t count, hipStream_t s) {
  dim3 grid(b->tiles[level], count), block(256);
  const LevelDesc* descs = b->d_descs + (size_t)level * b->n_pairs + p0;
  JobState* states = b->d_states + p0;
  // A group's slice starts at p0 x (the LARGEST tile count of any level): groups on different streams may be at
  // different levels at the same time, so their slices must be disjoint for every combination of levels.
  float* partials = b->d_partials + (size_t)p0 * b->max_tiles * GN_PARTIAL;
  unsigned* counters = b->d_counters + p0;
  const int ppt = (int)b->ppt[level];
  if (b->use_mfma) {
    if (b->group[level] == 2)
      hipLaunchKernelGGL((image_icp_mfma_kernel<2>), grid, block, 0, s, descs, states, b->gates[level],
                         partials, counters, solve, ppt);
    else
      hipLaunchKernelGGL((image_icp_mfma_kernel<1>), grid, block, 0, s, descs, states, b->gates[level],
                         partials, counters, solve, ppt);
  } else {
    if (b->group[level] == 4)
      hipLaunchKernelGGL((image_icp_kernel<4, false>), grid, block, 0, s, descs, states, b->gates[level], partials,
                         counters, solve, ppt);
    else if (b->group[level] == 2)
      hipLaunchKernelGGL((image_icp_kernel<2, false>), grid, block, 0, s, descs, states, b->gates[level], partials,
                         counters, solve, ppt);
    else if (b->merged_accumulators && solve.mode == SOLVE_IMAGE_ICP) {
      SolveArgs merged = solve;  // the partials hold add_weighted(geom, color) already: tell the tail
      merged.mode = SOLVE_IMAGE_ICP_MERGED;
      hipLaunchKernelGGL((image_icp_kernel<1, true>), grid, block, 0, s, descs, states, b->gates[level], partials,
                         counters, merged, ppt);
    } else
      hipLaunchKernelGGL((image_icp_kernel<1, false>), grid, block, 0, s, descs, states, b->gates[level], partials,
                         counters, solve, ppt);
  }
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}
#endif  // A3D_DIAGNOSTICS

// One per-iteration launch (head-solve form) of pairs [p0, p0 + count) at `level`: launch number `seq` of the sequence.
a3d_status launch_head_kernel(a3d_multiscale_batch* b, uint32_t level, uint32_t seq, const HeadArgs& prev, uint32_t p0,
                              uint32_t count, hipStream_t s) {
  const uint32_t P = b->n_pairs;
  const uint32_t job_stride = b->max_tiles * GN_PARTIAL;
  const JobState* st_in = b->d_states + (size_t)(seq & 1u) * P + p0;
  JobState* st_out = b->d_states + (size_t)((seq + 1u) & 1u) * P + p0;
  const float* part_in = b->d_partials + (size_t)((seq + 1u) & 1u) * b->partials_half + (size_t)p0 * job_stride;
  float* part_out = b->d_partials + (size_t)(seq & 1u) * b->partials_half + (size_t)p0 * job_stride;
  const LevelDesc* descs = b->d_descs + (size_t)level * P + p0;
  if (b->zmask)
    hipLaunchKernelGGL(image_icp_head_kernel<true>, dim3(b->tiles[level], count), dim3(256), 0, s, descs, st_in, st_out,
                       b->gates[level], part_in, part_out, job_stride, prev);
  else
    hipLaunchKernelGGL(image_icp_head_kernel<false>, dim3(b->tiles[level], count), dim3(256), 0, s, descs, st_in, st_out,
                       b->gates[level], part_in, part_out, job_stride, prev);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

// (Re)derives tiling from h_descs and uploads the descriptors.
a3d_status batch_commit_descs(a3d_multiscale_batch* b) {
  const uint32_t P = b->n_pairs, L = b->n_levels;
  plan_tiling(b);
  size_t max_partials = 1;
  for (uint32_t l = 0; l < L; ++l) max_partials = std::max(max_partials, (size_t)P * b->tiles[l] * GN_PARTIAL);
  b->max_tiles = (uint32_t)(max_partials / ((size_t)P * GN_PARTIAL));
  const char* zenv = A3D_DIAG_ENV("A3D_ICP_ZMASK");
  b->zmask = !(zenv && atoi(zenv) == 0);
  for (const LevelDesc& dsc : b->h_descs) b->zmask = b->zmask && (dsc.flags & 1u);
  b->partials_half = max_partials;  // floats per buffer: the launches alternate between two
  if (b->partials_capacity < 2 * max_partials) {  // grow-only: a reused engine keeps its buffer
    if (b->d_partials) A3D_HIP_TRY(hipFree(b->d_partials));
    b->d_partials = nullptr;
    b->partials_capacity = 0;
    A3D_HIP_TRY(hipMalloc((void**)&b->d_partials, 2 * max_partials * sizeof(float)));
    b->partials_capacity = 2 * max_partials;
  }
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
  b->ppt.assign(n_levels, 2);
  b->group.assign(n_levels, 1);
  b->h_descs.resize((size_t)n_levels * n_pairs);
  auto resident = [&](auto kernel, int fallback_per_cu) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1)
      per_cu = fallback_per_cu;
    return (uint32_t)std::max(1, ctx->num_cus) * (uint32_t)std::max(0, per_cu);
  };
  b->resident_blocks = resident(image_icp_head_kernel<true>, 4);
#ifdef A3D_DIAGNOSTICS
  b->persist_resident_blocks = resident(image_icp_persistent_kernel<true>, 0);
  if (const char* env = getenv("A3D_ICP_ACCUM")) {
    b->use_mfma = strcmp(env, "mfma") == 0;
    b->merged_accumulators = strcmp(env, "merged") == 0;
  }
  if (const char* env = getenv("A3D_ICP_HANDOFF")) b->head_solve = strcmp(env, "ticket") != 0;
  if (const char* env = getenv("A3D_ICP_SOLVE")) b->exact_solve = strcmp(env, "exact") == 0;
  if (const char* env = getenv("A3D_ICP_PERSISTENT")) b->use_level_kernel = atoi(env) != 0;
  if (const char* env = getenv("A3D_ICP_PERSISTENT_LEVELS")) b->level_mask = (uint32_t)strtoul(env, nullptr, 0);
  if (!b->head_solve) b->resident_blocks = resident(image_icp_kernel<1, false>, 4);
  if (b->merged_accumulators) b->resident_blocks = resident(image_icp_kernel<1, true>, 4);
  b->level_resident_blocks = resident(image_icp_level_kernel<1>, 0);
  if (b->use_level_kernel && b->level_resident_blocks) b->resident_blocks = b->level_resident_blocks;
#endif
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  {  // every small device array of the batch in ONE allocation (each hipMalloc / hipFree synchronises the device:
     // nine of them made creating and destroying a batch cost ~12 ms, six alignments' worth)
    size_t total = 0;
    auto take = [&](size_t bytes) {
      const size_t at = total;
      total += ((bytes + 255) / 256) * 256;
      return at;
    };
    const size_t o_descs = take(b->h_descs.size() * sizeof(LevelDesc)), o_states = take(2 * n_pairs * sizeof(JobState)),
                 o_counters = take((size_t)n_pairs * CTRL_WORDS_PER_PAIR * sizeof(unsigned)), o_epochs = take(n_pairs * sizeof(unsigned)),
                 o_poses = take(n_pairs * sizeof(Pose)), o_init = take(n_pairs * sizeof(Pose)),
                 o_status = take(n_pairs * sizeof(int32_t)), o_readback = take(GN_PARTIAL * sizeof(double));
    A3D_HIP_TRY(hipMalloc((void**)&b->d_block, total));
    char* base = (char*)b->d_block;
    b->d_descs = (LevelDesc*)(base + o_descs), b->d_states = (JobState*)(base + o_states);
    b->d_counters = (unsigned*)(base + o_counters);
#ifdef A3D_DIAGNOSTICS
    b->d_epochs = (unsigned*)(base + o_epochs);
#else
    (void)o_epochs;
#endif
    b->d_poses = (Pose*)(base + o_poses), b->d_init = (Pose*)(base + o_init);
    b->d_status = (int32_t*)(base + o_status), b->d_readback = (double*)(base + o_readback);
  }
  if (CTRL_WORDS_PER_PAIR)
    A3D_HIP_TRY(hipMemsetAsync(b->d_counters, 0, (size_t)n_pairs * CTRL_WORDS_PER_PAIR * sizeof(unsigned), ctx->stream));
  // measured (scripts/sweep.sh): 3 groups best from 16 to 128 pairs (+14 % at 64, +24 % at 16 over one
  // stream), 2 groups at 8 pairs (+21 %); a handful of pairs stays on one stream
  b->n_streams = n_pairs >= 12 ? 3u : (n_pairs >= 8 ? 2u : 1u);
  if (const char* env = A3D_DIAG_ENV("A3D_ICP_STREAMS")) b->n_streams = (uint32_t)std::min(4, std::max(1, atoi(env)));
  b->n_streams = std::min(b->n_streams, n_pairs);
#ifdef A3D_DIAGNOSTICS
  if (b->use_level_kernel) b->n_streams = 1u;
#endif
  if (b->n_streams > 1) {
    A3D_HIP_TRY(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
    for (uint32_t g = 1; g < b->n_streams; ++g) {
      A3D_TRY(ctx_side_stream(ctx, g - 1, &b->aux_streams[g - 1]));  // shared by the context's batches
      A3D_HIP_TRY(hipEventCreateWithFlags(&b->ev_join[g - 1], hipEventDisableTiming));
    }
  }
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
  b->last_init = d_init, b->last_levels = levels_to_run, b->last_matrices = d_matrices;
  b->persist_used = false;
  A3D_HIP_TRY(hipEventRecord(b->ev0, s));
  A3D_TRY(launch_job_init(s, b->d_states, d_init, (int)P));
  size_t kidx = 0;
  int trace_index = 0;
  auto profile_begin = [&](hipStream_t ps) -> a3d_status {
    if (!b->profile_kernels) return A3D_OK;
    if (b->kev.size() < 2 * (kidx + 1)) {
      hipEvent_t e0, e1;
      A3D_HIP_TRY(hipEventCreate(&e0));
      A3D_HIP_TRY(hipEventCreate(&e1));
      b->kev.push_back(e0);
      b->kev.push_back(e1);
    }
    A3D_HIP_TRY(hipEventRecord(b->kev[2 * kidx], ps));
    return A3D_OK;
  };
  uint32_t profile_level = 0;  // the level of the launch being bracketed
  auto profile_end = [&](hipStream_t ps) -> a3d_status {
    if (b->profile_kernels) A3D_HIP_TRY(hipEventRecord(b->kev[2 * kidx + 1], ps));
    if (b->kev_level.size() <= kidx) b->kev_level.resize(kidx + 1);
    b->kev_level[kidx] = (uint8_t)profile_level;
    ++kidx;
    return A3D_OK;
  };
  bool head = true;
  uint32_t S = d_trace ? 1u : b->n_streams;  // (not with a trace: its rows are indexed by the pair number inside a launch)
#ifdef A3D_DIAGNOSTICS
  // one last-block-form launch per level is only legal when every block of the level is resident at once
  bool level_kernel = b->use_level_kernel && !b->use_mfma && b->level_resident_blocks > 0;
  for (uint32_t l = 0; l < levels_to_run; ++l)
    level_kernel = level_kernel && (uint64_t)b->tiles[l] * P <= b->level_resident_blocks && b->group[l] == 1;
  // hybrid: only the levels in level_mask run as one launch (per stream group); the others launch per iteration
  uint32_t mask = 0;
  if (!level_kernel && !b->use_mfma && !d_trace && b->level_resident_blocks > 0)
    for (uint32_t l = 0; l < levels_to_run; ++l)
      if (((b->level_mask >> l) & 1u) && (uint64_t)b->tiles[l] * P <= b->level_resident_blocks && b->group[l] == 1)
        mask |= 1u << l;
  if (level_kernel || mask) A3D_HIP_TRY(hipMemsetAsync(b->d_epochs, 0, P * sizeof(unsigned), s));
  if (level_kernel) S = 1u;
  head = b->head_solve && !level_kernel && !mask && !b->use_mfma && !b->merged_accumulators && !getenv("A3D_ICP_NOSOLVE");
  for (uint32_t l = 0; l < levels_to_run; ++l) head = head && b->group[l] == 1;
  uint32_t epoch_base = 0;
#endif
  if (S > 1) {
    A3D_HIP_TRY(hipEventRecord(b->ev_fork, s));
    for (uint32_t g = 1; g < S; ++g) A3D_HIP_TRY(hipStreamWaitEvent(b->aux_streams[g - 1], b->ev_fork, 0));
  }
  // head-solve hand-off: launch k finishes iteration k - 1 at its head (`prev` describes iteration k - 1)
  HeadArgs prev{};
  prev.mode = SOLVE_NONE;
  prev.trace = d_trace, prev.trace_stride = trace_stride;
#ifdef A3D_DIAGNOSTICS
  prev.exact_solve = b->exact_solve;
#endif
  uint32_t seq = 0;  // iterations so far: launch `seq` reads state / partial buffer seq & 1 ... writes the other
  const uint32_t job_stride = b->max_tiles * GN_PARTIAL;
  bool finished_in_kernel = false;
  // ---- the persistent kernel takes the coarsest levels in persist_mask (all of them for a handful of pairs) ----
  uint32_t first_per_iteration = levels_to_run;  // levels below this index run as per-iteration launches
#ifdef A3D_DIAGNOSTICS
  uint32_t pmask = head ? (b->persist_mask & ((1u << levels_to_run) - 1u)) : 0u;
  if (pmask && !((pmask >> (levels_to_run - 1)) & 1u)) pmask = 0;  // (a truncated pyramid: the run must start at its coarsest level)
  for (uint32_t l = 0; l < levels_to_run && l < 16; ++l)  // a3d_multiscale_align_host: levels still being uploaded run as
    if (b->level_ready[l]) pmask = 0;                     // per-iteration launches, each behind its level's event (advisor r5)
  if (pmask) {
    PersistPlan plan{};
    plan.seq0 = 0, plan.counter_base = b->counter_base, plan.trace_index0 = 0, plan.trace_stride = trace_stride;
    plan.parity = b->persist_parity;
    b->persist_parity ^= 1u;
    plan.prev = prev;
    uint32_t published = 0;
    for (uint32_t l = levels_to_run; l-- > 0;) {
      if (!((pmask >> l) & 1u) || plan.n_levels == PERSIST_MAX_LEVELS) break;
      const a3d_icp_params& prm = b->params[l];
      PersistLevel& lv = plan.lv[plan.n_levels++];
      lv.desc_base = l * P, lv.iterations = (uint32_t)prm.max_iterations, lv.tiles = b->tiles[l];
      lv.weight = prm.weight, lv.color_weight = prm.color_weight, lv.gates = b->gates[l];
      published += lv.iterations * lv.tiles;
      first_per_iteration = l;
    }
    plan.finish = first_per_iteration == 0 ? 1u : 0u;
    uint32_t grid_x = 1;
    for (uint32_t k = 0; k < plan.n_levels; ++k) grid_x = std::max(grid_x, plan.lv[k].tiles);
    if (b->profile_kernels && !b->d_stamps)
      A3D_HIP_TRY(hipMalloc((void**)&b->d_stamps, (size_t)P * 2 * PERSIST_MAX_LEVELS * sizeof(unsigned long long)));
    profile_level = 255;  // (a persistent launch spans levels: its per-level times come from the kernel's stamps)
    b->persist_levels.clear();
    for (uint32_t k = 0; k < plan.n_levels; ++k) b->persist_levels.push_back(plan.lv[k].desc_base / P);
    for (uint32_t g = 0; g < S; ++g) {
      const uint32_t p0 = (uint32_t)((uint64_t)P * g / S), p1 = (uint32_t)((uint64_t)P * (g + 1) / S);
      hipStream_t gs = g == 0 ? s : b->aux_streams[g - 1];
      PersistOut po{};
      if (plan.finish) po.poses = b->d_poses + p0, po.status = b->d_status + p0, po.matrices = d_matrices ? d_matrices + (size_t)p0 * 16 : nullptr;
      po.stamps = b->profile_kernels ? b->d_stamps + (size_t)p0 * 2 * plan.n_levels : nullptr;
      PersistPlan gp = plan;
      gp.prev.trace = d_trace;
      A3D_TRY(profile_begin(gs));
      if (b->zmask)
        hipLaunchKernelGGL(image_icp_persistent_kernel<true>, dim3(grid_x, p1 - p0), dim3(256), 0, gs, b->d_descs + p0, P,
                           b->d_states + p0, b->d_partials + (size_t)p0 * job_stride, (uint32_t)b->partials_half, job_stride,
                           b->d_counters + (size_t)p0 * PAIR_CTRL_WORDS, gp, po);
      else
        hipLaunchKernelGGL(image_icp_persistent_kernel<false>, dim3(grid_x, p1 - p0), dim3(256), 0, gs, b->d_descs + p0, P,
                           b->d_states + p0, b->d_partials + (size_t)p0 * job_stride, (uint32_t)b->partials_half, job_stride,
                           b->d_counters + (size_t)p0 * PAIR_CTRL_WORDS, gp, po);
      A3D_HIP_TRY(hipGetLastError());
      A3D_TRY(profile_end(gs));
    }
    b->persist_used = true;
    b->counter_base += published;
    for (uint32_t k = 0; k < plan.n_levels; ++k) {  // what the next launch has to know about the last iteration run
      const PersistLevel& lv = plan.lv[k];
      if (!lv.iterations) continue;
      prev.weight = lv.weight, prev.color_weight = lv.color_weight, prev.mode = SOLVE_IMAGE_ICP, prev.tiles = lv.tiles;
      prev.first_in_level = lv.iterations == 1, prev.last_in_level = 1;
      seq += lv.iterations, trace_index += (int)lv.iterations;
      prev.trace_index = trace_index - 1;
    }
    finished_in_kernel = plan.finish != 0;
  }
#endif  // A3D_DIAGNOSTICS
  for (uint32_t l = first_per_iteration; l-- > 0;) {  // .rev(): coarsest level first (multiscale.rs:54-60)
    const a3d_icp_params& prm = b->params[l];
    profile_level = l;
    if (l < 16 && b->level_ready[l])  // the level's source arrays are still on their way (a3d_multiscale_align_host)
      for (uint32_t g = 0; g < S; ++g) A3D_HIP_TRY(hipStreamWaitEvent(g == 0 ? s : b->aux_streams[g - 1], b->level_ready[l], 0));
    if (head) {
      for (uint64_t it = 0; it < prm.max_iterations; ++it) {
        for (uint32_t g = 0; g < S; ++g) {
          const uint32_t p0 = (uint32_t)((uint64_t)P * g / S), p1 = (uint32_t)((uint64_t)P * (g + 1) / S);
          hipStream_t gs = g == 0 ? s : b->aux_streams[g - 1];
          A3D_TRY(profile_begin(gs));
          A3D_TRY(launch_head_kernel(b, l, seq, prev, p0, p1 - p0, gs));
          A3D_TRY(profile_end(gs));
        }
        prev.weight = prm.weight, prev.color_weight = prm.color_weight, prev.mode = SOLVE_IMAGE_ICP;
        prev.tiles = b->tiles[l];
        prev.first_in_level = it == 0, prev.last_in_level = it + 1 == prm.max_iterations;
        prev.trace_index = trace_index;
        ++trace_index;
        ++seq;
      }
      continue;
    }
#ifdef A3D_DIAGNOSTICS
    SolveArgs sa{};
    sa.weight = prm.weight, sa.color_weight = prm.color_weight;
    sa.mode = getenv("A3D_ICP_NOSOLVE") ? SOLVE_NONE : SOLVE_IMAGE_ICP;  // times the body alone (poses meaningless)
    sa.trace = d_trace, sa.trace_stride = trace_stride;
    if (level_kernel || ((mask >> l) & 1u)) {
      if (prm.max_iterations == 0) continue;
      LevelPlan plan;
      plan.iterations = (uint32_t)prm.max_iterations;
      plan.epoch_base = epoch_base;
      plan.trace_base = trace_index;
      sa.first_in_level = sa.last_in_level = 0;
      sa.trace_index = 0;
      for (uint32_t g = 0; g < S; ++g) {
        const uint32_t p0 = (uint32_t)((uint64_t)P * g / S), p1 = (uint32_t)((uint64_t)P * (g + 1) / S);
        hipStream_t gs = g == 0 ? s : b->aux_streams[g - 1];
        A3D_TRY(profile_begin(gs));
        hipLaunchKernelGGL((image_icp_level_kernel<1>), dim3(b->tiles[l], p1 - p0), dim3(256), 0, gs,
                           b->d_descs + (size_t)l * P + p0, b->d_states + p0, b->gates[l],
                           b->d_partials + (size_t)p0 * b->max_tiles * GN_PARTIAL, b->d_counters + p0, b->d_epochs + p0,
                           sa, plan, (int)b->ppt[l]);
        A3D_HIP_TRY(hipGetLastError());
        A3D_TRY(profile_end(gs));
      }
      epoch_base += plan.iterations;
      trace_index += (int)plan.iterations;
      continue;
    }
    for (uint64_t it = 0; it < prm.max_iterations; ++it) {
      sa.first_in_level = it == 0, sa.last_in_level = it + 1 == prm.max_iterations;
      sa.trace_index = trace_index;
      for (uint32_t g = 0; g < S; ++g) {
        const uint32_t p0 = (uint32_t)((uint64_t)P * g / S), p1 = (uint32_t)((uint64_t)P * (g + 1) / S);
        hipStream_t gs = g == 0 ? s : b->aux_streams[g - 1];
        A3D_TRY(profile_begin(gs));
        A3D_TRY(launch_pixel_kernel(b, l, sa, p0, p1 - p0, gs));
        A3D_TRY(profile_end(gs));
      }
      ++trace_index;
    }
#endif
  }
  for (uint32_t g = 1; g < S; ++g) {
    A3D_HIP_TRY(hipEventRecord(b->ev_join[g - 1], b->aux_streams[g - 1]));
    A3D_HIP_TRY(hipStreamWaitEvent(s, b->ev_join[g - 1], 0));
  }
  if (finished_in_kernel) {
    // the persistent kernel applied the last iteration and wrote the outputs
  } else if (head) {  // the last iteration is still pending: the finish kernel applies it
    A3D_TRY(launch_job_finish_head(s, b->d_states + (size_t)(seq & 1u) * P,
                                   b->d_partials + (size_t)((seq + 1u) & 1u) * b->partials_half, job_stride, prev,
                                   b->d_poses, b->d_status, d_matrices, (int)P));
  } else {
    A3D_TRY(launch_job_finish(s, b->d_states, b->d_poses, b->d_status, d_matrices, (int)P));
  }
  A3D_HIP_TRY(hipEventRecord(b->ev1, s));
  b->last_kernel_launches = kidx;
  b->last_groups = S;
  return A3D_OK;
}

// batch_enqueue failed after some launches may already have been issued on the main and side streams: no fence
// covers those kernels, so wait for them here — the caller's error cleanup typically frees the images at once, and
// their arenas must not go back to the pool under running kernels.
void batch_drain_after_failure(a3d_multiscale_batch* b) {
  for (uint32_t g = 1; g < b->n_streams; ++g)
    if (b->aux_streams[g - 1]) (void)hipStreamSynchronize(b->aux_streams[g - 1]);
  (void)hipStreamSynchronize(b->ctx->stream);
  (void)hipGetLastError();
  // a persistent launch may not have run to its end: start the pairs' counters afresh
  if (CTRL_WORDS_PER_PAIR)
    (void)hipMemsetAsync(b->d_counters, 0, (size_t)b->n_pairs * CTRL_WORDS_PER_PAIR * sizeof(unsigned), b->ctx->stream);
  b->counter_base = 0, b->persist_parity = 0;
}

a3d_status batch_collect_timing(a3d_multiscale_batch* b) {
  A3D_HIP_TRY(hipEventSynchronize(b->ev1));
  A3D_HIP_TRY(hipEventElapsedTime(&b->last_total_ms, b->ev0, b->ev1));
  b->last_kernel_ms = 0.f;
  for (int l = 0; l < 16; ++l) b->last_level_ms[l] = 0.f, b->last_level_launches[l] = 0;
  if (b->profile_kernels) {
    for (size_t k = 0; k < b->last_kernel_launches; ++k) {
      float ms = 0.f;
      A3D_HIP_TRY(hipEventElapsedTime(&ms, b->kev[2 * k], b->kev[2 * k + 1]));
      b->last_kernel_ms += ms;
      const uint32_t l = k < b->kev_level.size() ? b->kev_level[k] : 0u;
      if (l < 16) b->last_level_ms[l] += ms, ++b->last_level_launches[l];
    }
    if (b->persist_used && b->d_stamps && !b->persist_levels.empty()) {
      // the levels that ran inside the persistent kernel: per stream group, first start to last end of the level over
      // the group's pairs (s_memrealtime: 100 MHz), summed over the groups like the launch durations above; the
      // "launches" of such a level are its iterations x groups
      const uint32_t P = b->n_pairs, nl = (uint32_t)b->persist_levels.size(), S = b->last_groups;
      std::vector<unsigned long long> st((size_t)P * 2 * nl);
      A3D_HIP_TRY(hipMemcpy(st.data(), b->d_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      for (uint32_t g = 0; g < S; ++g) {
        const uint32_t p0 = (uint32_t)((uint64_t)P * g / S), p1 = (uint32_t)((uint64_t)P * (g + 1) / S);
        for (uint32_t k = 0; k < nl; ++k) {
          unsigned long long lo = ~0ull, hi = 0;
          for (uint32_t p = p0; p < p1; ++p) {
            lo = std::min(lo, st[((size_t)p * nl + k) * 2]);
            hi = std::max(hi, st[((size_t)p * nl + k) * 2 + 1]);
          }
          const uint32_t l = b->persist_levels[k];
          if (l < 16 && hi >= lo) {
            b->last_level_ms[l] += (float)((double)(hi - lo) * 1e-5);
            b->last_level_launches[l] += (uint32_t)b->params[l].max_iterations;
          }
        }
      }
    }
  }
  return A3D_OK;
}

// `behind_fence`: read on the context's copy stream, ordered after the batch's own last enqueue only, so that work
// enqueued on the context stream since then (the next round of a pipelined stream of batches) is not waited for.
a3d_status read_results(a3d_multiscale_batch* b, a3d_pose* out_poses, int32_t* out_status, a3d_status* worst,
                        bool behind_fence = false) {
  const uint32_t P = b->n_pairs;
  std::vector<Pose> poses(P);
  std::vector<int32_t> status(P);
  for (int attempt = 0; attempt < 2; ++attempt) {
    hipStream_t s = b->ctx->stream;
    if (behind_fence) {
      s = b->ctx->copy_stream;
      A3D_REQUIRE(b->fence->wait_on(s), A3D_HIP_ERROR, "hipStreamWaitEvent failed");
    }
    A3D_HIP_TRY(hipMemcpyAsync(poses.data(), b->d_poses, P * sizeof(Pose), hipMemcpyDeviceToHost, s));
    A3D_HIP_TRY(hipMemcpyAsync(status.data(), b->d_status, P * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    A3D_HIP_TRY(hipStreamSynchronize(s));
    bool spin_bound = false;
    for (uint32_t p = 0; p < P; ++p) spin_bound = spin_bound || status[p] == A3D_HIP_ERROR;
    if (!(spin_bound && b->persist_used) || attempt == 1) break;
    // A pair's blocks waited for partners that never became resident (another process holds part of the chip): this
    // batch goes back to one launch per iteration for good — same tiling, same bits — and the pass is repeated.
    batch_drain_after_failure(b);
    b->persist_disabled = true;
    b->persist_mask = 0;
    a3d_status st = batch_enqueue(b, b->last_init, b->last_levels, b->last_matrices, nullptr, 0);
    if (st != A3D_OK) {
      batch_drain_after_failure(b);
      return st;
    }
    b->fence->record(b->ctx->stream);
  }
  *worst = A3D_OK;
  for (uint32_t p = 0; p < P; ++p) {
    if (out_poses) pose_to_c(poses[p], &out_poses[p]);
    if (out_status) out_status[p] = status[p];
    if (status[p] != A3D_OK) *worst = (a3d_status)status[p];
  }
  return A3D_OK;
}

// One pair, any number of levels, host-synchronous: shared by image_icp_align and multiscale_align.
// The context's single-pair engine: created on first use, kept until the context dies, re-parameterised per call
// (an alignment then allocates nothing; hipMalloc / hipFree synchronise the whole device).
a3d_status acquire_single_engine(a3d_context* ctx, const a3d_icp_params* params, uint32_t n_levels,
                                 a3d_multiscale_batch** out) {
  a3d_multiscale_batch* e = (a3d_multiscale_batch*)ctx->icp_engine;
  if (e && e->n_levels != n_levels) {
    A3D_HIP_TRY(hipStreamSynchronize(ctx->stream));
    delete e;
    e = nullptr;
    ctx->icp_engine = nullptr;
  }
  if (!e) {
    std::unique_ptr<a3d_multiscale_batch> b;
    A3D_TRY(batch_create(ctx, params, n_levels, 1, &b));
    e = b.release();
    ctx->icp_engine = e;
    ctx->icp_engine_free = [](void* p) { delete (a3d_multiscale_batch*)p; };
  } else {
    e->params.assign(params, params + n_levels);
    for (uint32_t l = 0; l < n_levels; ++l) e->gates[l] = make_gates(params[l]);
  }
  *out = e;
  return A3D_OK;
}

a3d_status align_single(a3d_context* ctx, const a3d_icp_params* params, uint32_t n_levels,
                        const a3d_device_image* const* targets, const a3d_device_image* const* sources,
                        const a3d_pose* init, a3d_pose* out_pose, float* host_trace,
                        const hipEvent_t* level_ready = nullptr) {
  A3D_HIP_TRY(hipSetDevice(ctx->device));  // the caller's thread may have another device current
  a3d_multiscale_batch* b = nullptr;
  A3D_TRY(acquire_single_engine(ctx, params, n_levels, &b));
  struct ReadyGuard {  // the engine is cached with the context: the events are this call's only
    a3d_multiscale_batch* b;
    ~ReadyGuard() {
      for (auto& e : b->level_ready) e = nullptr;
    }
  } ready_guard{b};
  for (uint32_t l = 0; l < n_levels && l < 16; ++l) b->level_ready[l] = level_ready ? level_ready[l] : nullptr;
  for (uint32_t l = 0; l < n_levels; ++l) A3D_TRY(fill_desc(targets[l], sources[l], &b->h_descs[l]));
  A3D_TRY(batch_commit_descs(b));
  Pose* d_init = nullptr;
  Pose h_init;
  if (init) {
    h_init = pose_from_c(init);
    d_init = b->d_init;
    A3D_HIP_TRY(hipMemcpyAsync(d_init, &h_init, sizeof(Pose), hipMemcpyHostToDevice, ctx->stream));
  }
  uint64_t total_iters = 0;
  for (uint32_t l = 0; l < n_levels; ++l) total_iters += params[l].max_iterations;
  float* d_trace = nullptr;
  if (host_trace && total_iters) {
    A3D_HIP_TRY(hipMalloc((void**)&d_trace, total_iters * 8 * sizeof(float)));
    A3D_HIP_TRY(hipMemsetAsync(d_trace, 0, total_iters * 8 * sizeof(float), ctx->stream));
  }
  a3d_status st = batch_enqueue(b, d_init, n_levels, nullptr, d_trace, (int)total_iters);
  if (st != A3D_OK) batch_drain_after_failure(b);
  a3d_status worst = A3D_OK;
  if (st == A3D_OK) st = read_results(b, out_pose, nullptr, &worst);
  if (st == A3D_OK && d_trace)
    if (hipMemcpy(host_trace, d_trace, total_iters * 8 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
      st = A3D_HIP_ERROR;
  if (d_trace) hipFree(d_trace);
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

// One pass of the pixel loop from `pose` (image_icp.rs:76-148): the two accumulators as the product kernel sums them
// (block partials added in f64).  `which`: 0 the product kernel; diagnostics builds: 1 the exact-arithmetic cross-check
// kernel, 2 the merged-accumulator kernel.
static a3d_status accumulate_pass(a3d_context* ctx, const a3d_icp_params* params, const a3d_device_image* target,
                                  const a3d_device_image* source, const a3d_pose* pose, int which, double sums[GN_PARTIAL],
                                  const char* what) {
  std::unique_ptr<a3d_multiscale_batch> b;
  A3D_TRY(batch_create(ctx, params, 1, 1, &b));
  A3D_TRY(fill_desc(target, source, &b->h_descs[0]));
  A3D_TRY(batch_commit_descs(b.get()));
  Pose h_pose = pose ? pose_from_c(pose) : pose_eye();
  Pose* d_pose = b->d_init;
  a3d_status st = A3D_OK;
  hipStream_t s = ctx->stream;
  if (hipMemcpyAsync(d_pose, &h_pose, sizeof(Pose), hipMemcpyHostToDevice, s) != hipSuccess) st = A3D_HIP_ERROR;
  if (st == A3D_OK) st = launch_job_init(s, b->d_states, d_pose, 1);
  if (st == A3D_OK && which == 0) {  // the per-iteration launch with nothing to finish at its head: partials in buffer 0
    HeadArgs none{};
    none.mode = SOLVE_NONE;
    st = launch_head_kernel(b.get(), 0, 0, none, 0, 1, s);
  }
#ifdef A3D_DIAGNOSTICS
  if (st == A3D_OK && which == 1) {
    hipLaunchKernelGGL(image_icp_exact_kernel, dim3(b->tiles[0]), dim3(256), 0, s, b->d_descs, b->d_states, b->gates[0],
                       b->d_partials, (int)b->ppt[0]);
    if (hipGetLastError() != hipSuccess) st = A3D_HIP_ERROR;
  }
  if (st == A3D_OK && which == 2) {  // the merged kernel, with its tail switched off: the partials stay as the blocks wrote them
    SolveArgs sa{};
    sa.weight = params->weight, sa.color_weight = params->color_weight, sa.mode = SOLVE_NONE;
    hipLaunchKernelGGL((image_icp_kernel<1, true>), dim3(b->tiles[0], 1), dim3(256), 0, s, b->d_descs, b->d_states, b->gates[0],
                       b->d_partials, b->d_counters, sa, (int)b->ppt[0]);
    if (hipGetLastError() != hipSuccess) st = A3D_HIP_ERROR;
  }
#endif
  if (st == A3D_OK) st = launch_gn_readback(s, b->d_partials, (int)b->tiles[0], b->d_readback);
  if (st == A3D_OK && hipMemcpyAsync(sums, b->d_readback, GN_PARTIAL * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess)
    st = A3D_HIP_ERROR;
  if (hipStreamSynchronize(s) != hipSuccess) st = A3D_HIP_ERROR;
  if (st == A3D_HIP_ERROR) set_error("%s: HIP failure: %s", what, hipGetErrorString(hipGetLastError()));
  return st;
}

a3d_status a3d_image_icp_accumulate(a3d_context* ctx, const a3d_icp_params* params, const a3d_device_image* target,
                                    const a3d_device_image* source, const a3d_pose* pose, a3d_gn_state* out_geom,
                                    a3d_gn_state* out_color) {
  A3D_REQUIRE(ctx && params, A3D_INVALID_PARAMETER, "null argument");
  double sums[GN_PARTIAL];
  A3D_TRY(accumulate_pass(ctx, params, target, source, pose, 0, sums, "a3d_image_icp_accumulate"));
  gn_states_from_sums(sums, out_geom, out_color);
  return A3D_OK;
}

#ifdef A3D_DIAGNOSTICS
// Diagnostics build only: the same pass through image_icp_exact_kernel (every per-pixel value in the reference's own
// unfused operations): guards the product kernel's fused Jacobians against a real error hiding inside their tolerance.
a3d_status a3d_image_icp_accumulate_exact(a3d_context* ctx, const a3d_icp_params* params, const a3d_device_image* target,
                                          const a3d_device_image* source, const a3d_pose* pose, a3d_gn_state* out_geom,
                                          a3d_gn_state* out_color) {
  A3D_REQUIRE(ctx && params, A3D_INVALID_PARAMETER, "null argument");
  double sums[GN_PARTIAL];
  A3D_TRY(accumulate_pass(ctx, params, target, source, pose, 1, sums, "a3d_image_icp_accumulate_exact"));
  gn_states_from_sums(sums, out_geom, out_color);
  return A3D_OK;
}

// Diagnostics build only: the merged-accumulator kernel (A3D_ICP_ACCUM=merged), one pass from `pose`, returning what it
// hands to the solve: geom.add_weighted(color, weight, color_weight) (gaussnewton.rs:115-121) — H, g, the weighted
// residual sum and the combined count.
a3d_status a3d_image_icp_accumulate_weighted(a3d_context* ctx, const a3d_icp_params* params,
                                             const a3d_device_image* target, const a3d_device_image* source,
                                             const a3d_pose* pose, a3d_gn_state* out_state) {
  A3D_REQUIRE(ctx && params && out_state, A3D_INVALID_PARAMETER, "null argument");
  double sums[GN_PARTIAL];
  A3D_TRY(accumulate_pass(ctx, params, target, source, pose, 2, sums, "a3d_image_icp_accumulate_weighted"));
  int t = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      const float v = (float)sums[t++];
      out_state->hessian[i * 6 + j] = out_state->hessian[j * 6 + i] = v;
    }
  for (int i = 0; i < 6; ++i) out_state->gradient[i] = (float)sums[21 + i];
  out_state->squared_residual_sum = (float)sums[27] * params->weight + (float)sums[29] * params->color_weight;
  out_state->count = (uint64_t)sums[28] + (uint64_t)sums[30];
  return A3D_OK;
}
#endif  // A3D_DIAGNOSTICS

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

a3d_status a3d_multiscale_align_host(a3d_multiscale* ms, const a3d_range_image_view* source_pyramid,
                                     uint64_t n_source_levels, a3d_pose* out_pose) {
  A3D_REQUIRE(ms && out_pose && (source_pyramid || n_source_levels == 0), A3D_INVALID_PARAMETER, "null argument");
  uint32_t n = (uint32_t)std::min<uint64_t>(ms->params.size(), n_source_levels);  // izip! (multiscale.rs:54-59)
  if (n == 0) {
    pose_to_c(pose_eye(), out_pose);
    return A3D_OK;
  }
  for (uint32_t l = 0; l < n; ++l)  // the reference's expect() on the source (image_icp.rs:52-57), before anything is copied
    A3D_REQUIRE(source_pyramid[l].intensities, A3D_MISSING_FIELD, "Please, the source image should have intensity colors.");
  A3D_REQUIRE(n <= 16, A3D_INVALID_PARAMETER, "a3d_multiscale_align_host handles pyramids of up to 16 levels (upload with "
              "a3d_range_image_upload_pyramid and call a3d_multiscale_align for more)");
  a3d_device_image* images[16] = {};
  hipEvent_t ready[16] = {};
  A3D_TRY(upload_pyramid(ms->ctx, source_pyramid, n, images, ready));
  const a3d_status st = align_single(ms->ctx, ms->params.data(), n, ms->targets.data(), images, nullptr, out_pose, nullptr, ready);
  // align_single is host-synchronous when it succeeds; after a failure wait for whatever still copies or runs
  if (st != A3D_OK && st != A3D_SOLVE_FAILED) {
    hipStreamSynchronize(ms->ctx->stream);
    if (ms->ctx->copy_stream) hipStreamSynchronize(ms->ctx->copy_stream);
  }
  for (uint32_t l = 0; l < n; ++l) {
    if (ready[l]) hipEventDestroy(ready[l]);
    a3d_range_image_free(images[l]);
  }
  return st;
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
  for (size_t k = 0; k < (size_t)n_pairs * n_levels; ++k)
    attach_fence(target_pyramids[k], b->fence), attach_fence(source_pyramids[k], b->fence);
  A3D_TRY(batch_commit_descs(b.get()));
  A3D_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *out = b.release();
  return A3D_OK;
}

a3d_status a3d_multiscale_batch_rebind(a3d_multiscale_batch* b, const a3d_device_image* const* target_pyramids,
                                       const a3d_device_image* const* source_pyramids) {
  A3D_REQUIRE(b && target_pyramids && source_pyramids, A3D_INVALID_PARAMETER, "null argument");
  A3D_HIP_TRY(hipSetDevice(b->ctx->device));
  // the descriptors may still be in use by THIS batch's launches that have not run yet (or by their upload): wait for
  // those only — another batch running on the same context keeps running (two batches alternating over a stream of
  // rounds, the next one rebound and enqueued while the previous one computes)
  b->fence->wait();
  b->descs_uploaded->wait();
  const uint32_t P = b->n_pairs, L = b->n_levels;
  for (uint32_t p = 0; p < P; ++p)
    for (uint32_t l = 0; l < L; ++l)
      A3D_TRY(fill_desc(target_pyramids[(size_t)p * L + l], source_pyramids[(size_t)p * L + l],
                        &b->h_descs[(size_t)l * P + p]));
  for (size_t k = 0; k < (size_t)P * L; ++k) attach_fence(target_pyramids[k], b->fence), attach_fence(source_pyramids[k], b->fence);
  A3D_TRY(batch_commit_descs(b));
  b->descs_uploaded->record(b->ctx->stream);
  return A3D_OK;
}

// Host-synchronous read of the most recent pass's results (after an enqueue-only batch_align): waits for that pass
// only as far as stream order requires (everything enqueued on the context before this call).
a3d_status a3d_multiscale_batch_results(a3d_multiscale_batch* b, a3d_pose* out_poses_host, int32_t* out_status_host) {
  A3D_REQUIRE(b && (out_poses_host || out_status_host), A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(b->fence->recorded, A3D_INVALID_PARAMETER, "a3d_multiscale_batch_results: no pass has been enqueued on this batch");
  A3D_HIP_TRY(hipSetDevice(b->ctx->device));
  a3d_status worst;
  return read_results(b, out_poses_host, out_status_host, &worst, true);
}

a3d_status a3d_multiscale_batch_align(a3d_multiscale_batch* b, a3d_pose* out_poses_host, float* out_matrices_device,
                                      int32_t* out_status_host) {
  A3D_REQUIRE(b, A3D_INVALID_PARAMETER, "batch is null");
  A3D_HIP_TRY(hipSetDevice(b->ctx->device));
  const a3d_status enq = batch_enqueue(b, nullptr, b->n_levels, out_matrices_device, nullptr, 0);
  if (enq != A3D_OK) {  // some launches may be running with no fence recorded behind them
    batch_drain_after_failure(b);
    return enq;
  }
  b->fence->record(b->ctx->stream);  // the aux streams have been joined into the context stream by now
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

// The same per pyramid level (profiling on): sum of the launch durations at `level` and how many launches that was.
a3d_status a3d_multiscale_batch_last_level_ms(a3d_multiscale_batch* b, uint32_t level, float* out_ms, uint32_t* out_launches) {
  A3D_REQUIRE(b && out_ms && level < 16, A3D_INVALID_PARAMETER, "bad argument");
  A3D_TRY(batch_collect_timing(b));
  *out_ms = b->last_level_ms[level];
  if (out_launches) *out_launches = b->last_level_launches[level];
  return A3D_OK;
}

a3d_status a3d_multiscale_batch_persistent_levels(a3d_multiscale_batch* b, uint32_t* out_mask) {
  A3D_REQUIRE(b && out_mask, A3D_INVALID_PARAMETER, "null argument");
  *out_mask = 0;
  if (b->persist_used)
    for (uint32_t l : b->persist_levels) *out_mask |= 1u << l;
  return A3D_OK;
}

a3d_status a3d_multiscale_batch_concurrency(a3d_multiscale_batch* b, uint32_t* out_streams) {
  A3D_REQUIRE(b && out_streams, A3D_INVALID_PARAMETER, "null argument");
  *out_streams = b->n_streams;
  return A3D_OK;
}

// Instrumentation: runs the kernels' shared-reciprocal division on n (numerator, denominator) pairs
// drawn by the caller and counts the results that differ from IEEE `/` (expected: 0).
a3d_status a3d_selftest_division(a3d_context* ctx, const float* numerators, const float* denominators, uint64_t n,
                                 uint64_t* out_mismatches) {
  A3D_REQUIRE(ctx && numerators && denominators && out_mismatches && n > 0 && n < (1ull << 31), A3D_INVALID_PARAMETER,
              "bad argument");
  float *d_a = nullptr, *d_z = nullptr;
  unsigned *d_m = nullptr, h_m = 0;
  hipStream_t s = ctx->stream;
  a3d_status st = A3D_OK;
  if (hipMalloc((void**)&d_a, n * 4) != hipSuccess || hipMalloc((void**)&d_z, n * 4) != hipSuccess ||
      hipMalloc((void**)&d_m, 4) != hipSuccess || hipMemsetAsync(d_m, 0, 4, s) != hipSuccess ||
      hipMemcpyAsync(d_a, numerators, n * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d_z, denominators, n * 4, hipMemcpyHostToDevice, s) != hipSuccess)
    st = A3D_HIP_ERROR;
  if (st == A3D_OK) {
    hipLaunchKernelGGL(division_selftest_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, d_a, d_z,
                       (uint32_t)n, d_m);
    if (hipMemcpyAsync(&h_m, d_m, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      st = A3D_HIP_ERROR;
  }
  hipFree(d_a);
  hipFree(d_z);
  hipFree(d_m);
  if (st != A3D_OK) {
    set_error("a3d_selftest_division: HIP failure: %s", hipGetErrorString(hipGetLastError()));
    return st;
  }
  *out_mismatches = h_m;
  return A3D_OK;
}

// Instrumentation: the pose arithmetic of the iteration tail on the device (Transform::exp, Mul, transform_vector,
// transform_normal; src/transform.rs:44-118,138-153,205-220) for n items: out_composed[i] = exp(Se3(update_i)) * pose_i,
// out_points[i] = out_composed[i] . point_i, out_normals[i] = its rotation . point_i.  Host arrays in and out.
a3d_status a3d_selftest_transform(a3d_context* ctx, const float* updates6, const a3d_pose* poses, const float* points3,
                                  uint64_t n, a3d_pose* out_composed, float* out_points3, float* out_normals3) {
  A3D_REQUIRE(ctx && updates6 && points3 && out_composed && out_points3 && out_normals3 && n > 0 && n < (1ull << 24),
              A3D_INVALID_PARAMETER, "bad argument");
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  std::vector<Pose> h_poses(n), h_out(n);
  if (poses)
    for (uint64_t i = 0; i < n; ++i) h_poses[i] = pose_from_c(&poses[i]);
  char* d = nullptr;
  const size_t o_u = 0, o_p = o_u + n * 24, o_x = o_p + n * sizeof(Pose), o_c = o_x + n * 12, o_a = o_c + n * sizeof(Pose),
               o_b = o_a + n * 12, total = o_b + n * 12;
  A3D_HIP_TRY(hipMalloc((void**)&d, total));
  bool ok = hipMemcpyAsync(d + o_u, updates6, n * 24, hipMemcpyHostToDevice, s) == hipSuccess &&
            hipMemcpyAsync(d + o_x, points3, n * 12, hipMemcpyHostToDevice, s) == hipSuccess &&
            (!poses || hipMemcpyAsync(d + o_p, h_poses.data(), n * sizeof(Pose), hipMemcpyHostToDevice, s) == hipSuccess);
  if (ok)
    ok = launch_transform_selftest(s, (const float*)(d + o_u), poses ? (const Pose*)(d + o_p) : nullptr,
                                   (const float*)(d + o_x), (int)n, (Pose*)(d + o_c), (float*)(d + o_a),
                                   (float*)(d + o_b)) == A3D_OK;
  ok = ok && hipMemcpyAsync(h_out.data(), d + o_c, n * sizeof(Pose), hipMemcpyDeviceToHost, s) == hipSuccess &&
       hipMemcpyAsync(out_points3, d + o_a, n * 12, hipMemcpyDeviceToHost, s) == hipSuccess &&
       hipMemcpyAsync(out_normals3, d + o_b, n * 12, hipMemcpyDeviceToHost, s) == hipSuccess;
  if (hipStreamSynchronize(s) != hipSuccess) ok = false;
  hipFree(d);
  if (!ok) {
    set_error("a3d_selftest_transform: HIP failure: %s", hipGetErrorString(hipGetLastError()));
    return A3D_HIP_ERROR;
  }
  for (uint64_t i = 0; i < n; ++i) pose_to_c(h_out[i], &out_composed[i]);
  return A3D_OK;
}

#if defined(A3D_DIAGNOSTICS) && defined(A3D_TAIL_STAMPS)
extern "C" int a3d_debug_tail_stamps(unsigned long long out[16]) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(a3d::g_tail_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int a3d_debug_head_stamps(unsigned long long out[48]) {  // [0, 16): two launches' kernel stamps, [32, 40): inside the head, [40, 48): the same in shader-clock cycles
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(a3d::g_tail_stamps), 48 * sizeof(unsigned long long), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

a3d_status a3d_multiscale_batch_free(a3d_multiscale_batch* b) {
  if (!b) return A3D_OK;
  hipStreamSynchronize(b->ctx->stream);
  delete b;
  return A3D_OK;
}

}  // extern "C"
