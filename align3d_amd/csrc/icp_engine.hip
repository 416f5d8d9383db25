// Gauss-Newton reduce + solve + pose update on the device, one wave per job, so that an ICP
// level runs its iterations without a host round trip.
#include "icp_engine.hpp"

namespace a3d {

namespace {

__global__ void __launch_bounds__(64)
    gn_readback_kernel(const float* __restrict__ partials, int tiles, double* __restrict__ out58) {
  const int lane = threadIdx.x;
  if (lane < GN_PARTIAL) {
    double s = 0.0;
    for (int t = 0; t < tiles; ++t) s += (double)partials[(size_t)t * GN_PARTIAL + lane];
    out58[lane] = s;
  }
}

__global__ void job_init_kernel(JobState* __restrict__ states, const Pose* __restrict__ init_poses, int n_jobs) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  JobState s;
  s.pose = init_poses ? init_poses[j] : pose_eye();
  s.best = s.pose;
  s.best_residual = __builtin_inff();
  s.status = A3D_OK;
  s.last_residual = 0.0f;
  s.pad = 0;
  states[j] = s;
}

__global__ void job_finish_kernel(const JobState* __restrict__ states, Pose* __restrict__ poses_out,
                                  int32_t* __restrict__ status_out, float* __restrict__ matrices_out, int n_jobs) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  Pose p = states[j].pose;
  if (poses_out) poses_out[j] = p;
  if (status_out) status_out[j] = states[j].status;
  if (matrices_out) {
    float m[16];
    pose_to_matrix(p, m);
    for (int k = 0; k < 16; ++k) matrices_out[(size_t)j * 16 + k] = m[k];
  }
}

// Head-solve form: one block per job applies the job's last iteration (the partials of its last launch) and writes
// the outputs from the resulting state.
__global__ void __launch_bounds__(256)
    job_finish_head_kernel(const JobState* __restrict__ states_in, const float* __restrict__ partials_in,
                           uint32_t job_stride, HeadArgs head, Pose* __restrict__ poses_out,
                           int32_t* __restrict__ status_out, float* __restrict__ matrices_out) {
  __shared__ uint32_t s_state[JOB_WORDS];
  const int j = blockIdx.x;
  head_advance(states_in + j, nullptr, partials_in + (size_t)j * job_stride, head, j, s_state, true);
  if (threadIdx.x == 0) {
    const float* f = (const float*)s_state;
    const Pose p{{f[0], f[1], f[2]}, {f[3], f[4], f[5], f[6]}};
    if (poses_out) poses_out[j] = p;
    if (status_out) status_out[j] = (int32_t)s_state[15];
    if (matrices_out) {
      float m[16];
      pose_to_matrix(p, m);
      for (int k = 0; k < 16; ++k) matrices_out[(size_t)j * 16 + k] = m[k];
    }
  }
}

// a3d_selftest_transform: item i -> T = exp(Se3(update_i)) * pose_i with the tail's own device functions
// (tail_exp_se3 = the polynomial / libm trig switch + exp_se3_trig, compose), then T . point_i (transform_vector, as the
// pixel kernels call it) and R . point_i (transform_normal, as Icp calls it on source normals).
__global__ void transform_selftest_kernel(const float* __restrict__ updates6, const Pose* __restrict__ poses,
                                          const float* __restrict__ points3, int n, Pose* __restrict__ out_composed,
                                          float* __restrict__ out_points3, float* __restrict__ out_normals3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float u[6];
  for (int k = 0; k < 6; ++k) u[k] = updates6[6 * i + k];
  const Pose T = compose(tail_exp_se3(u), poses ? poses[i] : pose_eye());
  out_composed[i] = T;
  const V3 p{points3[3 * i], points3[3 * i + 1], points3[3 * i + 2]};
  const V3 a = transform_vector(T, p), b = transform_normal(T, p);
  out_points3[3 * i] = a.x, out_points3[3 * i + 1] = a.y, out_points3[3 * i + 2] = a.z;
  out_normals3[3 * i] = b.x, out_normals3[3 * i + 1] = b.y, out_normals3[3 * i + 2] = b.z;
}

}  // namespace

a3d_status launch_transform_selftest(hipStream_t stream, const float* updates6, const Pose* poses, const float* points3,
                                     int n, Pose* out_composed, float* out_points3, float* out_normals3) {
  hipLaunchKernelGGL(transform_selftest_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, updates6, poses, points3, n,
                     out_composed, out_points3, out_normals3);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

a3d_status launch_job_finish_head(hipStream_t stream, const JobState* states_in, const float* partials_in,
                                  uint32_t partials_job_stride, const HeadArgs& head, Pose* poses_out,
                                  int32_t* status_out, float* matrices_out, int n_jobs) {
  hipLaunchKernelGGL(job_finish_head_kernel, dim3(n_jobs), dim3(256), 0, stream, states_in, partials_in,
                     partials_job_stride, head, poses_out, status_out, matrices_out);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

a3d_status launch_gn_readback(hipStream_t stream, const float* partials, int tiles, double* out58) {
  hipLaunchKernelGGL(gn_readback_kernel, dim3(1), dim3(64), 0, stream, partials, tiles, out58);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

a3d_status launch_job_init(hipStream_t stream, JobState* states, const Pose* init_poses, int n_jobs) {
  hipLaunchKernelGGL(job_init_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, stream, states, init_poses, n_jobs);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

a3d_status launch_job_finish(hipStream_t stream, const JobState* states, Pose* poses_out, int32_t* status_out,
                             float* matrices_out, int n_jobs) {
  hipLaunchKernelGGL(job_finish_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, stream, states, poses_out,
                     status_out, matrices_out, n_jobs);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

void gn_states_from_sums(const double sums[GN_PARTIAL], a3d_gn_state* geom, a3d_gn_state* color) {
  a3d_gn_state* outs[2] = {geom, color};
  for (int a = 0; a < 2; ++a) {
    if (!outs[a]) continue;
    const double* s = sums + a * GN_ACC;
    int t = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) {
        float v = (float)s[t++];
        outs[a]->hessian[i * 6 + j] = v;
        outs[a]->hessian[j * 6 + i] = v;
      }
    for (int i = 0; i < 6; ++i) outs[a]->gradient[i] = (float)s[21 + i];
    outs[a]->squared_residual_sum = (float)s[27];
    outs[a]->count = (uint64_t)s[28];
  }
}

}  // namespace a3d
