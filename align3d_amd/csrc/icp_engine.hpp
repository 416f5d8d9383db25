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
};

// Host launchers (kernels live in icp_engine.hip); all enqueue on `stream` and return immediately.
// Sums the block partials of job 0 in f64 (test hook): out58 is device memory, GN_PARTIAL doubles.
a3d_status launch_gn_readback(hipStream_t stream, const float* partials, int tiles, double* out58);
// states[j] = {init_poses[j] (identity when null), same, +inf, A3D_OK}
a3d_status launch_job_init(hipStream_t stream, JobState* states, const Pose* init_poses, int n_jobs);
// poses_out[j] = states[j].pose ; status_out[j] ; matrices_out[j] = 4x4 row-major (each nullable)
a3d_status launch_job_finish(hipStream_t stream, const JobState* states, Pose* poses_out, int32_t* status_out,
                             float* matrices_out, int n_jobs);
// Head-solve form (HeadArgs below): applies the last iteration (partials of the last launch) and writes the outputs.
struct HeadArgs;
a3d_status launch_job_finish_head(hipStream_t stream, const JobState* states_in, const float* partials_in,
                                  uint32_t partials_job_stride, const HeadArgs& head, Pose* poses_out,
                                  int32_t* status_out, float* matrices_out, int n_jobs);
// Converts the 58 f64 sums of launch_gn_readback into the ABI's two a3d_gn_state.
void gn_states_from_sums(const double sums[GN_PARTIAL], a3d_gn_state* geom, a3d_gn_state* color);
// Device self-test of the pose arithmetic (a3d_selftest_transform): device arrays, n items.
a3d_status launch_transform_selftest(hipStream_t stream, const float* updates6, const Pose* poses, const float* points3,
                                     int n, Pose* out_composed, float* out_points3, float* out_normals3);

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

#if defined(A3D_DIAGNOSTICS) && defined(A3D_TAIL_STAMPS)
__device__ unsigned long long g_tail_stamps[64];  // [0, 16): last-block tail; [16 + 8 * (launch parity), +8): head kernel
#define A3D_STAMP(k)                                                                       \
  do {                                                                                     \
    if (threadIdx.x == 0 && job == 0) g_tail_stamps[k] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
// the head-solve path: the block of the job's LAST tile (scripts/head_stamps.py), slots 48 ..
#define A3D_HSTAMP(k)                                                                                                  \
  do {                                                                                                                 \
    if (threadIdx.x == 0 && job == 0 && blockIdx.x + 1 == gridDim.x)                                                   \
      g_tail_stamps[48 + (k)] = __builtin_amdgcn_s_memrealtime(), g_tail_stamps[56 + (k)] = __builtin_amdgcn_s_memtime(); /* (100 MHz; shader clock) */ \
  } while (0)
#else
#define A3D_STAMP(k) \
  do {               \
  } while (0)
#define A3D_HSTAMP(k) \
  do {                \
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
//   GaussNewton::add_weighted / weight / mean_squared_residual   src/optim/gaussnewton.rs:115-133
//   GaussNewton::solve (f64 Cholesky, nalgebra's update order)    src/optim/gaussnewton.rs:84-93
//   optim_transform = exp(update) * optim_transform, best tracking   src/icp/image_icp.rs:150-161,
//                                                                   src/icp/pcl_icp.rs:94-103
// The 6x6 factorisation runs one matrix element per lane with the operands exchanged by wave shuffles (column k:
// sqrt, scale the column, rank-1 update of the trailing columns — the same operations on the same operands, in the
// same order per element, as nalgebra's left-looking loop); the substitutions and the pose update are computed by
// every lane alike (all inputs are wave-uniform).
// ---- ticketless hand-off: the NEXT launch finishes the iteration ("head solve") ---------------------------------
// The per-iteration chain above is  partial stores (write-through) -> drain -> agent-scope ticket -> last block:
// partial loads -> solve -> agent-scope state stores -> kernel boundary -> state loads.  In the head form a launch
// only stores its block partials (plain stores; the kernel boundary publishes them) and exits; every block of the
// NEXT launch of the job loads the job's partials (tiles x 58 floats, L2 hits), sums them in the same fixed order and
// runs the same solve redundantly before its pixel pass: no atomic, no drain, no last-block serialisation, and the
// solve of every job runs at once instead of one after the other as jobs finish.  One block per job stores the new
// state, into the OTHER of two state buffers (the blocks of this launch are still reading the current one); the
// partials alternate between two buffers for the same reason.  job_finish_head_kernel applies the last iteration.
struct HeadArgs {
  float weight, color_weight;  // of the iteration being finished (the previous launch's level)
  int mode;                    // SOLVE_IMAGE_ICP, or SOLVE_NONE: no previous iteration, the state is used as it is
  uint32_t tiles;              // partials per job that the previous launch wrote
  int first_in_level, last_in_level;
  int trace_stride, trace_index;
  float* trace;  // nullable: [job][trace_stride][8]
#ifdef A3D_DIAGNOSTICS
  int exact_solve;  // A3D_ICP_SOLVE=exact: IEEE sqrt and divisions in the 6x6 solve, as nalgebra (cross-check of the rsqrt form)
#endif
};
__host__ __device__ inline HeadArgs head_args_of(const SolveArgs& a) {
  HeadArgs h{};
  h.weight = a.weight, h.color_weight = a.color_weight, h.mode = a.mode, h.tiles = 0;
  h.first_in_level = a.first_in_level, h.last_in_level = a.last_in_level;
  h.trace_stride = a.trace_stride, h.trace_index = a.trace_index, h.trace = a.trace;
  return h;
}
constexpr int JOB_WORDS = 18;
static_assert(sizeof(JobState) == JOB_WORDS * 4, "JobState is 18 words");

// sqrt(x) and 1 / sqrt(x) of a positive f64: hardware estimate, two Goldschmidt steps, one correction of the root
// (1-2 ulp; the compiler's sqrt + divide sequences cost ~3x the dependent operations).
__device__ __forceinline__ double rsqrt_f64(double x, double* root) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g), h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g), h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  *root = __builtin_fma(d, h, g);
  return h + h;
}

// sin(x), cos(x) for x = theta or theta / 2 in the iteration tail.  theta <= pi / 4 (every ICP update in practice): no
// range reduction needed, the single-precision minimax kernels of the Cephes library (sinf / cosf, < 1 ulp on this
// interval) — the device libm spends ~1 us of the single-lane tail in its general-argument path; above: the device libm.
__device__ __forceinline__ void tail_sincos(float x, float theta, float* sx, float* cx) {
  if (theta <= 0.78539816f) {
    const float z = x * x;
    *sx = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
    *cx = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  } else {
    *sx = sin_f32(x), *cx = cos_f32(x);
  }
}
// Transform::exp(Se3(update)) as the iteration tail evaluates it (tail_sincos + exp_se3_trig).
__device__ __forceinline__ Pose tail_exp_se3(const float update[6]) {
  Se3Trig tg;
  tg.theta = se3_theta(update);
  tail_sincos(0.5f * tg.theta, tg.theta, &tg.sin_half, &tg.cos_half);
  tail_sincos(tg.theta, tg.theta, &tg.sin_theta, &tg.cos_theta);
  return exp_se3_trig(update, tg);
}

// One wave (threads 0..63 of the block).  Lane k < 18 holds word k of the job's state in `state_bits`; `sums` = the
// 58 f64 totals of the iteration being finished (LDS).  Leaves the job's new state in s_state[0..18) (LDS) and, when
// st_out is given, in global memory.  Same operations as gn_finish_block, except that the Cholesky's column scaling
// and the substitutions multiply by the refined 1 / sqrt(pivot) instead of dividing (f64 results within 1-2 ulp,
// the f32 update they round to is the same but for a last-bit tie).
// COHERENT: st_out is read by other blocks of a running launch (the last-block forms): only the words that change are
// stored, with agent-scope write-through stores.  write_trace: this caller owns the job's trace row.
template <bool COHERENT = false>
__device__ __forceinline__ void gn_advance_wave(uint32_t state_bits, const double* sums, const HeadArgs& a, int job,
                                                uint32_t* s_state, JobState* st_out, bool write_trace) {
  const int tid = threadIdx.x & 63;  // lane: the solving wave need not be the block's first (head_sum_and_advance)
  auto wordf = [&](int lane) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)state_bits, lane)); };
  const int status_in = __builtin_amdgcn_readlane((int)state_bits, 15);
  uint32_t mine = state_bits;  // a frozen job, or nothing to apply: the state goes on unchanged
  if (a.mode != SOLVE_NONE && status_in == A3D_OK) {
    const bool image_mode = a.mode == SOLVE_IMAGE_ICP;
    const bool merged = a.mode == SOLVE_IMAGE_ICP_MERGED;
    // The lower triangle of H in registers, every lane alike (all inputs are wave-uniform): add_weighted in f32 as the
    // reference, then the f64 factorisation column by column — sqrt of the pivot, scale the column, rank-1 update of the
    // trailing columns: the same operations on the same operands, in the same order per element, as nalgebra's
    // left-looking loop.  (Until round 5 one lane held one element and the operands travelled by wave shuffles: three
    // ds_bpermute round trips per column and the factor read back from LDS by the substitutions made this the longest
    // stretch of a lone pair's iteration — 1.5 + 0.9 us of 6.6; in registers the dependent chain is ~70 + ~40 f64
    // operations.  Same bits: the arithmetic per element did not change.)
    double A[6][6];  // [r][c], r >= c used
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        const int t = tri6(c, r);
        const float hg = (float)sums[t], hc = (float)sums[GN_ACC + t];
        // add_weighted: H = Hg w1^2 + Hc w2^2 ; weight(): H *= w^2   (all f32)
        const float h = merged ? hg
                               : image_mode ? hg * (a.weight * a.weight) + hc * (a.color_weight * a.color_weight)
                                            : hg * (a.weight * a.weight);
        A[r][c] = (double)h;
      }
    int ok = 1;
    double rinv[6];
#ifdef A3D_DIAGNOSTICS
    double lii[6] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
#endif
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double diag = A[k][k];
      if (diag == 0.0 || !(diag >= 0.0)) ok = 0;  // zero, negative or NaN pivot: Cholesky::new() == None (wave-uniform)
      double sq;
      rinv[k] = rsqrt_f64(diag, &sq);
#ifdef A3D_DIAGNOSTICS
      if (a.exact_solve) sq = __builtin_sqrt(diag), lii[k] = sq;
#endif
#pragma unroll
      for (int r = k + 1; r < 6; ++r) {
#ifdef A3D_DIAGNOSTICS
        if (a.exact_solve) {
          A[r][k] = A[r][k] / sq;  // nalgebra: col /= sqrt(pivot)
          continue;
        }
#endif
        A[r][k] = A[r][k] * rinv[k];
      }
      A[k][k] = sq;
#pragma unroll
      for (int c = k + 1; c < 6; ++c)
#pragma unroll
        for (int r = c; r < 6; ++r) {
#ifdef A3D_DIAGNOSTICS
          if (a.exact_solve) {
            A[r][c] = A[r][c] - A[c][k] * A[r][k];
            continue;
          }
#endif
          A[r][c] = __builtin_fma(-A[c][k], A[r][k], A[r][c]);
        }
    }
    A3D_HSTAMP(3);  // factorisation done
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
    if (!ok) {  // the reference's unwrap() panics here: the job freezes with A3D_SOLVE_FAILED
      mine = tid == 15 ? (uint32_t)A3D_SOLVE_FAILED : tid == 16 ? __float_as_uint(residual) : mine;
    } else {
      double bvec[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float gg = (float)sums[21 + i], gc = (float)sums[GN_ACC + 21 + i];
        bvec[i] = (double)(merged ? gg : image_mode ? gg * a.weight + gc * a.color_weight : gg * a.weight);
      }
#define A3D_L(row, col) A[row][col]
#pragma unroll
      for (int i = 0; i < 6; ++i) {  // solve_lower_triangular (column oriented)
#ifdef A3D_DIAGNOSTICS
        const double coeff = a.exact_solve ? bvec[i] / lii[i] : bvec[i] * rinv[i];
#else
        const double coeff = bvec[i] * rinv[i];
#endif
        bvec[i] = coeff;
#pragma unroll
        for (int rr = i + 1; rr < 6; ++rr) {
#ifdef A3D_DIAGNOSTICS
          if (a.exact_solve) {
            bvec[rr] = bvec[rr] - coeff * A3D_L(rr, i);
            continue;
          }
#endif
          bvec[rr] = __builtin_fma(-coeff, A3D_L(rr, i), bvec[rr]);
        }
      }
#pragma unroll
      for (int i = 5; i >= 0; --i) {  // ad_solve_lower_triangular: L^T x = b
        double d = 0.0;
#pragma unroll
        for (int rr = i + 1; rr < 6; ++rr) {
#ifdef A3D_DIAGNOSTICS
          if (a.exact_solve) {
            d = d + A3D_L(rr, i) * bvec[rr];
            continue;
          }
#endif
          d = __builtin_fma(A3D_L(rr, i), bvec[rr], d);
        }
#ifdef A3D_DIAGNOSTICS
        bvec[i] = a.exact_solve ? (bvec[i] - d) / lii[i] : (bvec[i] - d) * rinv[i];
#else
        bvec[i] = (bvec[i] - d) * rinv[i];
#endif
      }
#undef A3D_L
      float update[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) update[i] = (float)bvec[i];
      A3D_HSTAMP(4);  // substitutions done
      Pose pose{{wordf(0), wordf(1), wordf(2)}, {wordf(3), wordf(4), wordf(5), wordf(6)}};
      Pose best{{wordf(7), wordf(8), wordf(9)}, {wordf(10), wordf(11), wordf(12), wordf(13)}};
      float best_residual = wordf(14);
      if (a.first_in_level) {  // ImageIcp::align starts every level with best = initial, +inf
        best_residual = __builtin_inff();
        best = pose;
      }
      Se3Trig tg;
      tg.theta = se3_theta(update);
      {  // sin and cos of theta / 2 (even lanes) and of theta (odd lanes) in one pass
        const float x = (tid & 1) ? tg.theta : 0.5f * tg.theta;
        float sx, cx;
        tail_sincos(x, tg.theta, &sx, &cx);
        auto lane_of = [](float v, int lane) {
          return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
        };
        tg.sin_half = lane_of(sx, 0), tg.cos_half = lane_of(cx, 0);
        tg.sin_theta = lane_of(sx, 1), tg.cos_theta = lane_of(cx, 1);
      }
      pose = compose(exp_se3_trig(update, tg), pose);  // Transform::exp(Se3(update)) * optim_transform
      if (residual < best_residual) {                  // stores the transform AFTER the update (image_icp.rs:158-161)
        best_residual = residual;
        best = pose;
      }
      if (a.trace && write_trace && tid == 0) {
        float* tr = a.trace + ((size_t)job * a.trace_stride + a.trace_index) * 8;
        tr[0] = residual;
        tr[1] = pose.t.x, tr[2] = pose.t.y, tr[3] = pose.t.z;
        tr[4] = pose.q.i, tr[5] = pose.q.j, tr[6] = pose.q.k, tr[7] = pose.q.w;
      }
      if (a.last_in_level) pose = best;  // align() returns best_transform; the next level starts from it
      const float w[17] = {pose.t.x, pose.t.y, pose.t.z, pose.q.i, pose.q.j, pose.q.k, pose.q.w,
                           best.t.x, best.t.y, best.t.z, best.q.i, best.q.j, best.q.k, best.q.w,
                           best_residual, 0.0f, residual};
#pragma unroll
      for (int k = 0; k < 17; ++k)
        if (k != 15) mine = tid == k ? __float_as_uint(w[k]) : mine;
    }
  }
  A3D_HSTAMP(5);  // pose updated
  if (tid < JOB_WORDS) {
    if (s_state) s_state[tid] = mine;
    if (st_out) {
      if (!COHERENT)
        ((uint32_t*)st_out)[tid] = mine;
      else if (mine != state_bits)
        __hip_atomic_store((uint32_t*)st_out + tid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#ifdef A3D_DIAGNOSTICS  // the last-block (ticket) hand-off: kept as the cross-check of the head-solve form
// The last-block forms (block_publish_and_finish, the level kernel): called by every thread of the job's last block
// with the 58 f64 totals in LDS; the state is read and written in place with agent-scope accesses.
__device__ __forceinline__ void gn_finish_block(JobState* st, const double* sums, const SolveArgs& a, int job) {
  const int tid = threadIdx.x;
  if (tid >= 64) return;  // ONE wave finishes the iteration: wave-synchronous, no block barrier
  const uint32_t state_bits =
      tid < JOB_WORDS ? __hip_atomic_load((const uint32_t*)st + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  gn_advance_wave<true>(state_bits, sums, head_args_of(a), job, nullptr, st, true);
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

#endif  // A3D_DIAGNOSTICS

// ---- head-solve hand-off (the HeadArgs comment above): every block of the NEXT launch finishes the iteration ----
// Every thread of the block calls it (blocks of >= 256 threads).  Sums the job's `h.tiles` partials of the previous
// launch — thread (component pair cg, slice s) adds tiles s, s + 8, ... in tile order, then the eight slices in a
// fixed order: the same order as the last-block form above, so both forms give the same bits — runs the solve on
// wave 0 and leaves the job's state in s_state (LDS, JOB_WORDS words) behind a block barrier.
// COHERENT: the partials were written by other blocks of the SAME launch (the persistent kernel): every load is an
// agent-scope (sc1) load that bypasses this CU's L1.  Otherwise a kernel boundary published them: plain loads.
template <bool COHERENT>
__device__ __forceinline__ unsigned long long ld_partial_pair(const unsigned long long* p) {
  if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
// `state_bits`: lane k < JOB_WORDS of wave 0 holds word k of the job's current state.
template <bool COHERENT>
__device__ __forceinline__ void head_sum_and_advance(uint32_t state_bits, const float* prev_partials, const HeadArgs& h,
                                                     int job, uint32_t* s_state, bool write_trace,
                                                     JobState* st_out = nullptr, int solver_wave = 0) {
  __shared__ double s_sums[8][64];
  const int tid = threadIdx.x;
  A3D_HSTAMP(0);  // head entered
  if (h.mode != SOLVE_NONE) {
    const int cg = tid & 31, slice = tid >> 5;
    const uint32_t tiles = h.tiles;
    if (cg < GN_PARTIAL / 2 && slice < 8) {
      double sum0 = 0.0, sum1 = 0.0;
      const unsigned long long* base = (const unsigned long long*)prev_partials + cg;
      uint32_t t = slice;
      for (; t + 248 < tiles; t += 256) {
        unsigned long long v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = ld_partial_pair<COHERENT>(base + (size_t)(t + 8 * k) * (GN_PARTIAL / 2));
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          sum0 += (double)__uint_as_float((unsigned)v[k]);
          sum1 += (double)__uint_as_float((unsigned)(v[k] >> 32));
        }
      }
      // what is left (at most 256 tiles for this launch geometry) in ONE round of loads: up to 32 in flight per thread,
      // masked — a lone pair's 150-200 tiles took four dependent rounds of eight; tile order as before, so the same bits
      if (t + 56 < tiles) {
        unsigned long long v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const uint32_t tt = t + 8 * k;
          v[k] = tt < tiles ? ld_partial_pair<COHERENT>(base + (size_t)tt * (GN_PARTIAL / 2)) : 0ull;  // +0.0f, +0.0f
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          sum0 += (double)__uint_as_float((unsigned)v[k]);
          sum1 += (double)__uint_as_float((unsigned)(v[k] >> 32));
        }
      } else {
        unsigned long long v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t tt = t + 8 * k;
          v[k] = tt < tiles ? ld_partial_pair<COHERENT>(base + (size_t)tt * (GN_PARTIAL / 2)) : 0ull;  // +0.0f, +0.0f
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          sum0 += (double)__uint_as_float((unsigned)v[k]);
          sum1 += (double)__uint_as_float((unsigned)(v[k] >> 32));
        }
      }
      s_sums[slice][2 * cg] = sum0;
      s_sums[slice][2 * cg + 1] = sum1;
    }
    A3D_HSTAMP(1);  // this thread's partials loaded and added
    __syncthreads();
    if (tid < GN_PARTIAL) {
      const int c = tid;
      s_sums[0][c] = ((s_sums[0][c] + s_sums[1][c]) + (s_sums[2][c] + s_sums[3][c])) +
                     ((s_sums[4][c] + s_sums[5][c]) + (s_sums[6][c] + s_sums[7][c]));
    }
    __syncthreads();
  }
  A3D_HSTAMP(2);  // the 58 totals in LDS
  if ((tid >> 6) == solver_wave) gn_advance_wave<false>(state_bits, s_sums[0], h, job, s_state, st_out, write_trace);
  __syncthreads();
  A3D_HSTAMP(6);  // state in LDS for every thread
}

// WHICH wave solves (round 6).  The solve is ~1 250 wave-uniform instructions on ONE wave (every lane alike) while the
// block's other three waves wait: it runs at one wave's issue rate.  A CU holds up to four blocks of the batch kernel, and
// the runtime places wave w of every block on SIMD w: with the solve always on wave 0, four resident blocks' solves queued
// on one SIMD while three SIMDs idled.  The solving wave now rotates with the block's position in the grid.
#ifndef A3D_HEAD_ROTATE
#define A3D_HEAD_ROTATE 1
#endif
__device__ __forceinline__ void head_advance(const JobState* st_in, JobState* st_out, const float* prev_partials,
                                             const HeadArgs& h, int job, uint32_t* s_state, bool write_trace) {
  const int solver_wave = A3D_HEAD_ROTATE ? (int)((blockIdx.x + blockIdx.y) & 3u) : 0;
  uint32_t state_bits = 0;
  const int lane = (int)threadIdx.x - 64 * solver_wave;
  if (lane >= 0 && lane < (int)JOB_WORDS) state_bits = ((const uint32_t*)st_in)[lane];
  head_sum_and_advance<false>(state_bits, prev_partials, h, job, s_state, write_trace, st_out, solver_wave);
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

#ifdef A3D_DIAGNOSTICS
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

#endif  // A3D_DIAGNOSTICS

}  // namespace a3d
