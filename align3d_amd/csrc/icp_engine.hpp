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

// Wave64 + LDS reduction of N per-thread accumulators to one block partial (called by all threads).
template <int N>
__device__ __forceinline__ void block_reduce_store(float (&acc)[N], float* __restrict__ out) {
  constexpr int WAVES = 4;  // 256-thread blocks
  __shared__ float red[WAVES][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float v = acc[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
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
