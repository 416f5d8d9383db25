// Gauss-Newton reduce + solve + pose update on the device, one wave per job, so that an ICP
// level runs its iterations without a host round trip.
#include "icp_engine.hpp"

namespace a3d {

namespace {

__device__ __forceinline__ void unpack_acc(const double* s, float H[36], float g[6], float* ssq, double* count) {
  int t = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      float v = (float)s[t++];
      H[i * 6 + j] = v;
      H[j * 6 + i] = v;
    }
  for (int i = 0; i < 6; ++i) g[i] = (float)s[21 + i];
  *ssq = (float)s[27];
  *count = s[28];
}

__global__ void __launch_bounds__(64)
    gn_solve_kernel(JobState* __restrict__ states, const float* __restrict__ partials, int tiles, float weight,
                    float color_weight, int mode, int first_in_level, int last_in_level, float* __restrict__ trace,
                    int trace_stride, int trace_index) {
  const int job = blockIdx.x, lane = threadIdx.x;
  JobState* st = &states[job];
  if (st->status != A3D_OK) return;  // a failed job stays frozen (the reference panicked here)
  __shared__ double sums[GN_PARTIAL];
  if (lane < GN_PARTIAL) {
    const float* p = partials + (size_t)job * tiles * GN_PARTIAL + lane;
    double s = 0.0;
    for (int t = 0; t < tiles; ++t) s += (double)p[(size_t)t * GN_PARTIAL];
    sums[lane] = s;
  }
  __syncthreads();
  if (lane != 0) return;

  float Hg[36], gg[6], ssq_g, Hc[36], gc[6], ssq_c;
  double cnt_g, cnt_c;
  unpack_acc(sums, Hg, gg, &ssq_g, &cnt_g);
  unpack_acc(sums + GN_ACC, Hc, gc, &ssq_c, &cnt_c);

  float H[36], g[6], residual;
  double count;
  if (mode == SOLVE_IMAGE_ICP) {
    // GaussNewton::add_weighted (gaussnewton.rs:115-121): w^2 on H, w on g and on sum r^2, counts add
    const float w1s = weight * weight, w2s = color_weight * color_weight;
    for (int k = 0; k < 36; ++k) H[k] = Hg[k] * w1s + Hc[k] * w2s;
    for (int k = 0; k < 6; ++k) g[k] = gg[k] * weight + gc[k] * color_weight;
    float ssq = ssq_g * weight + ssq_c * color_weight;
    count = cnt_g + cnt_c;
    residual = ssq / (float)count;  // mean_squared_residual (:131-133)
  } else {
    // Icp: residual first, then GaussNewton::weight (pcl_icp.rs:94-95, gaussnewton.rs:124-128)
    count = cnt_g;
    residual = ssq_g / (float)count;
    const float ws = weight * weight;
    for (int k = 0; k < 36; ++k) H[k] = Hg[k] * ws;
    for (int k = 0; k < 6; ++k) g[k] = gg[k] * weight;
  }

  Pose pose = st->pose;
  float best_residual = st->best_residual;
  Pose best = st->best;
  if (first_in_level) {  // ImageIcp::align starts every level with best = initial, +inf
    best_residual = __builtin_inff();
    best = pose;
  }
  float update[6];
  if (count == 0.0 || !gn_solve6(H, g, update)) {  // solve() == None -> unwrap() panics
    st->status = A3D_SOLVE_FAILED;
    st->last_residual = residual;
    return;
  }
  pose = compose(exp_se3(update), pose);  // Transform::exp(Se3(update)) * optim_transform
  if (residual < best_residual) {         // stores the transform AFTER the update (image_icp.rs:158-161)
    best_residual = residual;
    best = pose;
  }
  if (trace) {
    float* tr = trace + ((size_t)job * trace_stride + trace_index) * 8;
    tr[0] = residual;
    tr[1] = pose.t.x, tr[2] = pose.t.y, tr[3] = pose.t.z;
    tr[4] = pose.q.i, tr[5] = pose.q.j, tr[6] = pose.q.k, tr[7] = pose.q.w;
  }
  if (last_in_level) pose = best;  // align() returns best_transform; the next level starts from it
  st->pose = pose;
  st->best = best;
  st->best_residual = best_residual;
  st->last_residual = residual;
}

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

}  // namespace

a3d_status launch_gn_solve(hipStream_t stream, JobState* states, const float* partials, int n_jobs, int tiles,
                           float weight, float color_weight, SolveMode mode, bool first_in_level,
                           bool last_in_level, float* trace, int trace_stride, int trace_index) {
  hipLaunchKernelGGL(gn_solve_kernel, dim3(n_jobs), dim3(64), 0, stream, states, partials, tiles, weight,
                     color_weight, (int)mode, first_in_level ? 1 : 0, last_in_level ? 1 : 0, trace, trace_stride,
                     trace_index);
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
