// Device-side Gauss-Newton state shared by the image ICP and point-cloud ICP engines:
// per-job pose/best/status, the block-partial layout and the solve kernel
// (GaussNewton::add_weighted / weight / mean_squared_residual / solve, src/optim/gaussnewton.rs:84-133;
// exp * T and best tracking, src/icp/image_icp.rs:150-161, src/icp/pcl_icp.rs:94-103).
#pragma once
#include "common.hpp"

namespace a3d {

// One accumulator = 21 upper-triangle JtJ + 6 Jtr + sum r^2 + count.
constexpr int GN_ACC = 29;
// Block partial of the image ICP kernel: geometric accumulator then colour accumulator.
constexpr int GN_PARTIAL = 2 * GN_ACC;  // 58 floats

struct JobState {
  Pose pose;            // optim_transform
  Pose best;            // best_transform
  float best_residual;  // best_residual
  int32_t status;       // a3d_status of this job (A3D_OK or A3D_SOLVE_FAILED)
  float last_residual;  // residual of the most recent iteration (trace / tests)
  uint32_t pad;
};

enum SolveMode : int {
  SOLVE_IMAGE_ICP = 0,  // geom.add_weighted(color, w, cw); residual = weighted mean (image_icp.rs:150-151)
  SOLVE_PCL_ICP = 1,    // residual = mean, then weight(w)  (pcl_icp.rs:94-95)
  SOLVE_NONE = 2,       // leave the block partials alone (test hook: read the accumulators back)
  // The partials already hold geom.add_weighted(color, w, cw): H and g were accumulated from the weighted
  // Jacobians (GN_MERGED layout below); only the residual's two sums and counts are still separate.
  SOLVE_IMAGE_ICP_MERGED = 3
};

// Merged accumulator of the image ICP kernel: the solve only ever sees H = Hg w^2 + Hc cw^2 and
// g = gg w + gc cw (GaussNewton::add_weighted, gaussnewton.rs:115-121), so a thread accumulates those directly
// from J' = w J instead of two separate 6x6 systems: 31 running sums instead of 58.
//   [0, 21) H   [21, 27) g   27 sum rg^2   28 count_g   29 sum rc^2   30 count_c
constexpr int GN_MERGED = 31;

// What the last block of a job needs in order to finish the iteration on the device.
struct SolveArgs {
  float weight, color_weight;
  int mode;
  int first_in_level, last_in_level;
  int trace_stride, trace_index;
  float* trace;  // nullable: [job][trace_stride][8] = residual, t, q
  int reverse;   // image ICP: this launch walks pairs and tiles from the last to the first (cache reuse, image_icp.hip)
};

// Host launchers (kernels live in icp_engine.hip); all enqueue on `stream` and return immediately.
// Sums the block partials of job 0 in f64 (test hook): out58 is device memory, GN_PARTIAL doubles.
a3d_status launch_gn_readback(hipStream_t stream, const float* partials, int tiles, double* out58);
// states[j] = {init_poses[j] (identity when null), same, +inf, A3D_OK}
a3d_status launch_job_init(hipStream_t stream, JobState* states, const Pose* init_poses, int n_jobs);
// poses_out[j] = states[j].pose ; status_out[j] ; matrices_out[j] = 4x4 row-major (each nullable)
a3d_status launch_job_finish(hipStream_t stream, const JobState* states, Pose* poses_out, int32_t* status_out,
                             float* matrices_out, int n_jobs);
// Converts the 58 f64 sums of launch_gn_readback into the ABI's two a3d_gn_state.
void gn_states_from_sums(const double sums[GN_PARTIAL], a3d_gn_state* geom, a3d_gn_state* color);

// ---- block reduction of per-thread accumulators --------------------------------------------------
// A wave-level reduce-scatter: after six exchange steps lane L holds the wave total of accumulator L.
// Each step halves the number of values a lane carries, so the whole wave reduction costs ~2N lane
// exchanges instead of 6N (v_permlane32_swap / v_permlane16_swap across rows, DPP inside a row).
template <int CTRL>
__device__ __forceinline__ float dpp_recv(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}

// Returns, in lane L (L < N, N <= 64), the sum over the wave's 64 lanes of acc[L].
template <int N>
__device__ __forceinline__ float wave_reduce_scatter(const float (&acc)[N]) {
  static_assert(N <= 64, "at most one value per lane");
  const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  float b[32];
  if constexpr (N > 32) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {  // lanes 0-31 keep index i, lanes 32-63 keep index i + 32
      const float lo = acc[i], hi = (i + 32 < N) ? acc[i + 32] : 0.0f;
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
      b[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) {  // both halves end up with the same totals
      const float v = (i < N) ? acc[i] : 0.0f;
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      b[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
  }
  float c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {  // even rows (of 16 lanes) keep i, odd rows keep i + 16
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(b[i]), __float_as_uint(b[i + 16]), false, false);
    c[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
  float d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)  // partner lane ^ 8 (row_ror:8)
    d[i] = (b3 ? c[i + 8] : c[i]) + dpp_recv<0x128>(b3 ? c[i] : c[i + 8]);
  float e[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)  // partner 7 - lane within each group of 8 (row_half_mirror): opposite bit 2
    e[i] = (b2 ? d[i + 4] : d[i]) + dpp_recv<0x141>(b2 ? d[i] : d[i + 4]);
  float f[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)  // partner lane ^ 2 (quad_perm [2,3,0,1])
    f[i] = (b1 ? e[i + 2] : e[i]) + dpp_recv<0x4E>(b1 ? e[i] : e[i + 2]);
  // partner lane ^ 1 (quad_perm [1,0,3,2])
  return (b0 ? f[1] : f[0]) + dpp_recv<0xB1>(b0 ? f[0] : f[1]);
}

// Wave reduce-scatter + LDS across the block's 4 waves -> one block partial of N floats.
template <int N, bool WRITE_THROUGH = false, int WAVES = 4>
__device__ __forceinline__ void block_reduce_store(const float (&acc)[N], float* __restrict__ out) {
  __shared__ float red[WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  red[wave][lane] = wave_reduce_scatter<N>(acc);
  __syncthreads();
  if (threadIdx.x < N) {
    float s = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) s += red[w][threadIdx.x];
    if (WRITE_THROUGH)  // global_store_dword sc1: leaves this CU's L1 and the XCD's L2 right away
      __hip_atomic_store((unsigned*)out + threadIdx.x, __float_as_uint(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      out[threadIdx.x] = s;
  }
}

#ifdef A3D_TAIL_STAMPS
__device__ unsigned long long g_tail_stamps[16];
#define A3D_STAMP(k)                                                                       \
  do {                                                                                     \
    if (threadIdx.x == 0 && job == 0) g_tail_stamps[k] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define A3D_STAMP(k) \
  do {               \
  } while (0)
#endif

// ---- job state accessors ---------------------------------------------------------------------------
// Inside the persistent level kernel the state is handed from block to block within one launch, so it is
// always read and written with agent-scope (sc1) accesses that bypass the CU's L1.
__device__ __forceinline__ float ld_coherent(const float* p) {
  return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_coherent(float* p, float v) {
  __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ Pose load_pose(const Pose* p) {
  const float* f = (const float*)p;
  Pose r;
  r.t = {ld_coherent(f), ld_coherent(f + 1), ld_coherent(f + 2)};
  r.q = {ld_coherent(f + 3), ld_coherent(f + 4), ld_coherent(f + 5), ld_coherent(f + 6)};
  return r;
}
__device__ __forceinline__ void store_pose(Pose* p, const Pose& v) {
  float* f = (float*)p;
  st_coherent(f, v.t.x), st_coherent(f + 1, v.t.y), st_coherent(f + 2, v.t.z);
  st_coherent(f + 3, v.q.i), st_coherent(f + 4, v.q.j), st_coherent(f + 5, v.q.k), st_coherent(f + 6, v.q.w);
}
__device__ __forceinline__ int load_status(const JobState* st) {
  return __hip_atomic_load(&st->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_status(JobState* st, int v) {
  __hip_atomic_store(&st->status, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- finishing an iteration on the device ---------------------------------------------------------
// Called by every thread of the job's last block.  sums = GN_PARTIAL f64 totals in LDS.
//   GaussNewton::add_weighted / weight / mean_squared_residual   src/optim/gaussnewton.rs:115-133
//   GaussNewton::solve (f64 Cholesky, nalgebra's update order)    src/optim/gaussnewton.rs:84-93
//   optim_transform = exp(update) * optim_transform, best tracking   src/icp/image_icp.rs:150-161,
//                                                                   src/icp/pcl_icp.rs:94-103
// The 6x6 factorisation runs one matrix element per thread out of LDS (column k: sqrt, scale the
// column, rank-1 update of the trailing columns — the same operations on the same operands, in the
// same order per element, as nalgebra's left-looking loop), which keeps the solve out of the
// accumulate kernel's register budget; the substitutions and the pose update are one lane.
__device__ __forceinline__ void gn_finish_block(JobState* st, const double* sums, const SolveArgs& a, int job) {
  __shared__ double Lm[36];
  const int tid = threadIdx.x;
  if (tid >= 64) return;  // ONE wave finishes the iteration: everything below is wave-synchronous, no block barrier
  const int r = tid / 6, c = tid % 6;
  const bool cell = tid < 36;  // lane (r, c) owns matrix element [r][c]
  const bool image_mode = a.mode == SOLVE_IMAGE_ICP;
  const bool merged = a.mode == SOLVE_IMAGE_ICP_MERGED;
  // the job state (pose 0..6, best 7..13, best_residual 14): one float per lane, in flight during the solve
  const float state_word = tid < 15 ? ld_coherent((const float*)st + tid) : 0.0f;
  double v = 0.0;
  if (cell) {
    const int t = tri6(r < c ? r : c, r < c ? c : r);
    const float hg = (float)sums[t], hc = (float)sums[GN_ACC + t];
    // add_weighted: H = Hg w1^2 + Hc w2^2 ; weight(): H *= w^2   (all f32)
    const float h = merged ? hg
                           : image_mode ? hg * (a.weight * a.weight) + hc * (a.color_weight * a.color_weight)
                                        : hg * (a.weight * a.weight);
    v = (double)h;
  }
  A3D_STAMP(3);
  // Cholesky, column by column, operands exchanged with wave shuffles: sqrt of the pivot, scale the column,
  // rank-1 update of the trailing columns.
  int ok = 1;
  const int src_c = cell ? c * 6 : 0, src_r = cell ? r * 6 : 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double diag = __shfl(v, k * 7, 64);
    if (diag == 0.0 || !(diag >= 0.0)) ok = 0;  // zero, negative or NaN pivot: Cholesky::new() == None (wave-uniform)
    const double sq = sqrt(diag);
    if (tid == k * 7) v = sq;
    if (cell && c == k && r > k) v = v / sq;
    const double lck = __shfl(v, src_c + k, 64), lrk = __shfl(v, src_r + k, 64);
    if (cell && c > k && r >= c) v = (-lck) * lrk + v;
  }
  A3D_STAMP(4);
  if (cell) Lm[tid] = v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the LDS writes above are visible to lane 0 below
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // From here on every lane computes the same values (all inputs are wave-uniform); only lane 0 stores.
  float residual;
  {
    const float ssq_g = (float)sums[27], ssq_c = (float)sums[merged ? 29 : GN_ACC + 27];
    const double cnt_g = sums[28], cnt_c = sums[merged ? 30 : GN_ACC + 28];
    // ImageIcp: weighted sum / combined count ; Icp: plain mean taken before weight()
    const double count = (image_mode || merged) ? cnt_g + cnt_c : cnt_g;
    const float ssq = (image_mode || merged) ? ssq_g * a.weight + ssq_c * a.color_weight : ssq_g;
    residual = ssq / (float)count;
    if (!(count != 0.0)) ok = 0;  // solve(): None if count == 0
  }
  if (!ok) {  // the reference's unwrap() panics here
    if (tid == 0) {
      st_coherent(&st->last_residual, residual);
      store_status(st, A3D_SOLVE_FAILED);
    }
    return;
  }
  // The two substitutions in registers on one lane (the accumulators are dead here, so this fits the kernel's
  // register budget): 12 divisions and 45 multiply / add pairs in one dependent chain.
  double Lr[21], bvec[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float gg = (float)sums[21 + i], gc = (float)sums[GN_ACC + 21 + i];
    bvec[i] = (double)(merged ? gg : image_mode ? gg * a.weight + gc * a.color_weight : gg * a.weight);
#pragma unroll
    for (int j = 0; j <= i; ++j) Lr[i * (i + 1) / 2 + j] = Lm[i * 6 + j];  // lower triangle, row-major packed
  }
#define A3D_L(row, col) Lr[(row) * ((row) + 1) / 2 + (col)]
#pragma unroll
  for (int i = 0; i < 6; ++i) {  // solve_lower_triangular (column oriented)
    const double coeff = bvec[i] / A3D_L(i, i);
    bvec[i] = coeff;
#pragma unroll
    for (int rr = i + 1; rr < 6; ++rr) bvec[rr] = -coeff * A3D_L(rr, i) + bvec[rr];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {  // ad_solve_lower_triangular: L^T x = b
    double d = 0.0;
#pragma unroll
    for (int rr = i + 1; rr < 6; ++rr) d += A3D_L(rr, i) * bvec[rr];
    bvec[i] = (bvec[i] - d) / A3D_L(i, i);
  }
#undef A3D_L
  float update[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) update[i] = (float)bvec[i];
  A3D_STAMP(5);
  auto word = [&](int lane) {
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(state_word), lane));
  };
  Pose pose{{word(0), word(1), word(2)}, {word(3), word(4), word(5), word(6)}};
  Pose best{{word(7), word(8), word(9)}, {word(10), word(11), word(12), word(13)}};
  float best_residual = word(14);
  if (a.first_in_level) {  // ImageIcp::align starts every level with best = initial, +inf
    best_residual = __builtin_inff();
    best = pose;
  }
  // sin and cos of theta / 2 (even lanes) and of theta (odd lanes) in one pass
  Se3Trig tg;
  tg.theta = se3_theta(update);
  {
    const float x = (tid & 1) ? tg.theta : 0.5f * tg.theta;
    float sx, cx;
    if (tg.theta <= 0.78539816f) {
      // |x| <= pi/4 (every ICP update in practice): no range reduction needed, the single-precision minimax
      // kernels of the Cephes library (sinf / cosf, < 1 ulp on this interval) — the device libm spends ~1 us of
      // this single-lane tail in its general-argument path
      const float z = x * x;
      sx = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
      cx = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    } else {
      sx = sin_f32(x), cx = cos_f32(x);
    }
    auto lane_of = [](float v, int lane) {
      return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
    };
    tg.sin_half = lane_of(sx, 0), tg.cos_half = lane_of(cx, 0);
    tg.sin_theta = lane_of(sx, 1), tg.cos_theta = lane_of(cx, 1);
  }
  pose = compose(exp_se3_trig(update, tg), pose);  // Transform::exp(Se3(update)) * optim_transform
  if (residual < best_residual) {         // stores the transform AFTER the update (image_icp.rs:158-161)
    best_residual = residual;
    best = pose;
  }
  if (a.trace && tid == 0) {
    float* tr = a.trace + ((size_t)job * a.trace_stride + a.trace_index) * 8;
    tr[0] = residual;
    tr[1] = pose.t.x, tr[2] = pose.t.y, tr[3] = pose.t.z;
    tr[4] = pose.q.i, tr[5] = pose.q.j, tr[6] = pose.q.k, tr[7] = pose.q.w;
  }
  if (a.last_in_level) pose = best;  // align() returns best_transform; the next level starts from it
  // one store instruction for the whole state: lane k writes float k of JobState (status, word 15, stays)
  {
    const float w[17] = {pose.t.x, pose.t.y, pose.t.z, pose.q.i, pose.q.j, pose.q.k, pose.q.w,
                         best.t.x, best.t.y, best.t.z, best.q.i, best.q.j, best.q.k, best.q.w,
                         best_residual, 0.0f, residual};
    float mine = w[0];
#pragma unroll
    for (int k = 1; k < 17; ++k) mine = tid == k ? w[k] : mine;
    if (tid < 17 && tid != 15) st_coherent((float*)st + tid, mine);
  }
  A3D_STAMP(6);
}

// Tail of an accumulate kernel (all 256 threads call it): reduce the block's accumulators to one
// partial, publish it, take a ticket on the job's counter; the block that arrives last sums all the
// job's partials in f64 and runs the solve, while blocks of other jobs are still accumulating.
// Hand-off (CDNA guide, Guideline 16, form R1): the partial is stored write-through (sc1), the storing
// wave drains its stores, a barrier, then ONE lane adds to the counter (agent-scope atomic) — no
// release fence, which would write back the whole L2 once per block.  The last block does one agent
// acquire, a barrier, and reads the partials with sc1 loads.  `job_partials` = [tiles][GN_PARTIAL].
// Second half of the tail: the block's partial has been stored write-through by its wave 0.
__device__ __forceinline__ bool block_publish_and_finish(float* __restrict__ job_partials, uint32_t tiles,
                                                         unsigned* __restrict__ counter, JobState* st,
                                                         const SolveArgs& args, int job);

template <int N, int WAVES = 4>
__device__ __forceinline__ void block_finish(const float (&acc)[N], float* __restrict__ job_partials, uint32_t tile,
                                             uint32_t tiles, unsigned* __restrict__ counter, JobState* st,
                                             const SolveArgs& args, int job) {
  float* out = job_partials + (size_t)tile * GN_PARTIAL;
  block_reduce_store<N, true, WAVES>(acc, out);
  if (N < GN_PARTIAL && threadIdx.x >= N && threadIdx.x < GN_PARTIAL)
    __hip_atomic_store((unsigned*)out + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  block_publish_and_finish(job_partials, tiles, counter, st, args, job);
}

// Returns true in the block that came last and ran the solve.
__device__ __forceinline__ bool block_publish_and_finish(float* __restrict__ job_partials, uint32_t tiles,
                                                         unsigned* __restrict__ counter, JobState* st,
                                                         const SolveArgs& args, int job) {
  if (args.mode == SOLVE_NONE) return false;
  __shared__ unsigned s_is_last;
  __shared__ double s_sums[8][64];
  A3D_STAMP(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_is_last = ticket == tiles - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_is_last) return false;
  A3D_STAMP(1);
  // ---- last block of this job ----
  // Every load of the handed-off partials below is an sc1 (agent-scope) load that bypasses this CU's L1,
  // so no acquire fence (an L1 invalidate) is needed.  Thread (pair of components cg, slice s) sums tiles
  // s, s + 8, ... in tile order with 8-byte loads, up to eight in flight; then the 8 slices are added in a
  // fixed order: the totals do not depend on which block came last.
  {
    const int cg = threadIdx.x & 31, slice = threadIdx.x >> 5;
    double sum0 = 0.0, sum1 = 0.0;
    if (cg < GN_PARTIAL / 2 && slice < 8) {  // the first 256 threads of the block (blocks may be larger)
      const unsigned long long* base = (const unsigned long long*)job_partials + cg;
      uint32_t t = slice;
      // a single pair runs on ~256 fat blocks: 32 partials per slice, all of them in flight at once (one memory
      // round trip instead of four; the accumulators are dead here, so the registers are free)
      for (; t + 248 < tiles; t += 256) {
        unsigned long long v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k)
          v[k] = __hip_atomic_load(base + (size_t)(t + 8 * k) * (GN_PARTIAL / 2), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          sum0 += (double)__uint_as_float((unsigned)v[k]);
          sum1 += (double)__uint_as_float((unsigned)(v[k] >> 32));
        }
      }
      for (; t + 56 < tiles; t += 64) {
        unsigned long long v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          v[k] = __hip_atomic_load(base + (size_t)(t + 8 * k) * (GN_PARTIAL / 2), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          sum0 += (double)__uint_as_float((unsigned)v[k]);
          sum1 += (double)__uint_as_float((unsigned)(v[k] >> 32));
        }
      }
      unsigned long long v[8];
      int n = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t tt = t + 8 * k;
        v[k] = tt < tiles ? __hip_atomic_load(base + (size_t)tt * (GN_PARTIAL / 2), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT)
                          : 0ull;  // +0.0f, +0.0f: adds nothing
        n += tt < tiles;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sum0 += (double)__uint_as_float((unsigned)v[k]);
        sum1 += (double)__uint_as_float((unsigned)(v[k] >> 32));
      }
      (void)n;
      s_sums[slice][2 * cg] = sum0;
      s_sums[slice][2 * cg + 1] = sum1;
    }
  }
  __syncthreads();
  if (threadIdx.x < GN_PARTIAL) {
    const int c = threadIdx.x;
    s_sums[0][c] = ((s_sums[0][c] + s_sums[1][c]) + (s_sums[2][c] + s_sums[3][c])) +
                   ((s_sums[4][c] + s_sums[5][c]) + (s_sums[6][c] + s_sums[7][c]));
  }
  __syncthreads();
  A3D_STAMP(2);
  if (threadIdx.x == 0)  // ready for the next launch (ordered by the kernel boundary)
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  gn_finish_block(st, s_sums[0], args, job);
  return true;
}

// acc[0..21) += J J^T (upper triangle), acc[21..27) += J r, acc[27] += r^2, acc[28] += 1
// (GaussNewton::step, src/optim/gaussnewton.rs:47-77).  The sums use fused multiply-adds: the sum over
// samples is re-associated on the GPU anyway, and an fma only removes one rounding per term.
__device__ __forceinline__ void gn_step(float* __restrict__ acc, float r, const float J[6]) {
  int t = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) {
      acc[t] = __builtin_fmaf(J[i], J[j], acc[t]);
      ++t;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[21 + i] = __builtin_fmaf(J[i], r, acc[21 + i]);
  acc[27] = __builtin_fmaf(r, r, acc[27]);
  acc[28] += 1.0f;
}

// One pixel into the merged accumulator: Jg, Jc already multiplied by their weights (Jc and rc zero when the colour
// term is rejected), rg / rc the plain residuals.  Same FMA count as two gn_step calls, 27 fewer live registers.
__device__ __forceinline__ void gn_step_merged(float* __restrict__ acc, float rg, const float Jg[6], float rc,
                                               const float Jc[6], float color_live) {
  int t = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) {
      acc[t] = __builtin_fmaf(Jc[i], Jc[j], __builtin_fmaf(Jg[i], Jg[j], acc[t]));
      ++t;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[21 + i] = __builtin_fmaf(Jc[i], rc, __builtin_fmaf(Jg[i], rg, acc[21 + i]));
  acc[27] = __builtin_fmaf(rg, rg, acc[27]);
  acc[28] += 1.0f;
  acc[29] = __builtin_fmaf(rc, rc, acc[29]);
  acc[30] += color_live;
}

}  // namespace a3d
