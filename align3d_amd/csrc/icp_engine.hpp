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
  SOLVE_PCL_ICP = 1     // residual = mean, then weight(w)  (pcl_icp.rs:94-95)
};

// Host launchers (kernels live in icp_engine.hip); all enqueue on `stream` and return immediately.
// partials: [job][tiles][GN_PARTIAL].  trace (nullable): [job][trace_stride][8] = residual, t, q.
a3d_status launch_gn_solve(hipStream_t stream, JobState* states, const float* partials, int n_jobs, int tiles,
                           float weight, float color_weight, SolveMode mode, bool first_in_level,
                           bool last_in_level, float* trace, int trace_stride, int trace_index);
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
template <int N>
__device__ __forceinline__ void block_reduce_store(const float (&acc)[N], float* __restrict__ out) {
  constexpr int WAVES = 4;  // 256-thread blocks
  __shared__ float red[WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  red[wave][lane] = wave_reduce_scatter<N>(acc);
  __syncthreads();
  if (threadIdx.x < N) {
    float s = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) s += red[w][threadIdx.x];
    out[threadIdx.x] = s;
  }
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

}  // namespace a3d
