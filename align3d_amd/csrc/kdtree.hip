// R3dTree::new / nearest (src/kdtree.rs:28-105) and Icp (src/icp/pcl_icp.rs:15-108).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <memory>

#include "icp_engine.hpp"
#include "kdtree.hpp"

using namespace a3d;

namespace a3d {

// Shape pass: the (start, len) recursion is data independent, so the depth of the deepest leaf is known
// before any sorting.
void kdtree_shape(uint32_t n, uint32_t* max_depth_out, uint64_t* n_leaves_out, uint64_t* n_internal_out) {
  uint32_t max_depth = 0;
  uint64_t n_leaves = 0, n_internal = 0;
  struct Item { uint32_t len, depth; };
  std::vector<Item> stack{{n, 0}};
  while (!stack.empty()) {
    Item it = stack.back();
    stack.pop_back();
    if (it.len <= 16) {
      max_depth = std::max(max_depth, it.depth);
      ++n_leaves;
      continue;
    }
    ++n_internal;
    uint32_t mid = it.len / 2;
    stack.push_back({mid, it.depth + 1});
    stack.push_back({it.len - mid, it.depth + 1});
  }
  *max_depth_out = max_depth, *n_leaves_out = n_leaves, *n_internal_out = n_internal;
}

a3d_status kdtree_build_host(const float* points, uint32_t n, std::vector<float>* split,
                             std::vector<float4>* leaves, std::vector<uint32_t>* slot_of_point,
                             uint32_t* max_depth_out, uint64_t* n_leaves_out, uint64_t* n_internal_out) {
  uint32_t max_depth = 0;
  {
    uint64_t a, b;
    kdtree_shape(n, &max_depth, &a, &b);
  }
  A3D_REQUIRE(max_depth <= 23, A3D_INVALID_PARAMETER, "point cloud too large for the implicit kd-tree layout (leaf byte offsets are 32-bit)");
  const uint64_t n_split = (1ull << max_depth) - 1, n_slots = (1ull << max_depth) * 16;
  split->assign(n_split, 0.0f);
  const float inf = std::numeric_limits<float>::infinity();
  leaves->assign(n_slots, make_float4(inf, inf, inf, 0.0f));
  slot_of_point->assign(n, 0);
  std::vector<uint32_t> idx(n);
  for (uint32_t i = 0; i < n; ++i) idx[i] = i;
  uint64_t n_leaves = 0, n_internal = 0;
  bool nan_seen = false;

  struct Seg { uint32_t start, len, node, path, depth; };
  std::vector<Seg> stack{{0, n, 0, 0, 0}};
  while (!stack.empty()) {
    Seg s = stack.back();
    stack.pop_back();
    if (s.len <= 16) {  // Leaf: points gathered in the current (parent-sorted) order (kdtree.rs:32-37)
      const uint64_t base = ((uint64_t)s.path << (max_depth - s.depth)) * 16;
      for (uint32_t k = 0; k < s.len; ++k) {
        const uint32_t pi = idx[s.start + k];
        float bits;
        memcpy(&bits, &pi, 4);
        (*leaves)[base + k] = make_float4(points[3 * pi], points[3 * pi + 1], points[3 * pi + 2], bits);
        (*slot_of_point)[pi] = (uint32_t)(base + k);
      }
      ++n_leaves;
      continue;
    }
    const int k = (int)(s.depth % 3);
    auto first = idx.begin() + s.start, last = first + s.len;
    for (auto it = first; it != last; ++it)
      if (std::isnan(points[3 * (*it) + k])) nan_seen = true;  // partial_cmp().unwrap() (kdtree.rs:43)
    if (nan_seen) break;
    // slice::sort_by is stable; partial_cmp treats -0.0 == +0.0
    std::stable_sort(first, last, [&](uint32_t a, uint32_t b) { return points[3 * a + k] < points[3 * b + k]; });
    const uint32_t mid = s.len / 2;
    (*split)[s.node] = points[3 * idx[s.start + mid] + k];
    ++n_internal;
    stack.push_back({s.start + mid, s.len - mid, 2 * s.node + 2, 2 * s.path + 1, s.depth + 1});
    stack.push_back({s.start, mid, 2 * s.node + 1, 2 * s.path, s.depth + 1});
  }
  A3D_REQUIRE(!nan_seen, A3D_NAN_IN_INPUT, "NaN coordinate in kd-tree input (the reference panics in partial_cmp().unwrap())");
  *max_depth_out = max_depth;
  *n_leaves_out = n_leaves;
  *n_internal_out = n_internal;
  return A3D_OK;
}

}  // namespace a3d

namespace {

// Launch geometry of the two query kernels.  The split table's top `lds_levels` heap levels are staged in LDS by
// every block (2^levels - 1 floats), so fat blocks amortise the staging.  Default: one 1024-thread block per CU with
// 15 levels (128 KiB of the CU's 160 KiB): a 500k-point tree then descends entirely out of LDS, and each thread
// serves two or more queries, which staggers the waves' descent and scan phases (measured on MI355X, 500k x 500k:
// 17.3 us against 19.4 us for 512 threads / 12 levels / 4 blocks per CU; scripts/kd_sweep.sh).
struct KdLaunch {
  uint32_t block, lds_levels, blocks;
  size_t lds_bytes;
};

KdLaunch kd_launch_config(const a3d_kdtree* t, uint64_t m, const char* env_prefix, uint32_t def_block,
                          uint32_t def_levels, uint32_t blocks_per_cu_cap) {
  auto env_u = [&](const char* suffix, uint32_t def) {
#ifdef A3D_DIAGNOSTICS  // A3D_KD_* / A3D_PCL_* launch-geometry knobs (scripts/kd_sweep.sh)
    std::string name = std::string(env_prefix) + suffix;
    if (const char* v = getenv(name.c_str())) return (uint32_t)atoi(v);
#endif
    (void)suffix, (void)env_prefix;
    return def;
  };
  KdLaunch L;
  L.block = env_u("_BLOCK", def_block);
  if (L.block != 256 && L.block != 512 && L.block != 1024) L.block = def_block;
  L.lds_levels = std::min<uint32_t>(env_u("_LDS_LEVELS", def_levels), 15u);  // 15 levels = 128 KiB of the CU's 160
  L.lds_levels = std::min<uint32_t>(L.lds_levels, t->max_depth);             // deeper levels do not exist
  const uint64_t want = (1ull << L.lds_levels) - 1;
  L.lds_bytes = (size_t)std::min<uint64_t>(want, t->n_split) * sizeof(float);
  const uint32_t per_cu = std::max<uint32_t>(1, env_u("_BLOCKS_PER_CU", blocks_per_cu_cap));
  L.blocks = (uint32_t)std::min<uint64_t>((m + L.block - 1) / L.block, (uint64_t)std::max(1, t->ctx->num_cus) * per_cu);
  L.blocks = std::max<uint32_t>(1, L.blocks);
  return L;
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
    kdtree_nearest_kernel(const float* __restrict__ split, const float4* __restrict__ leaves, uint32_t n,
                          uint32_t n_split, uint32_t max_depth, uint32_t lds_levels,
                          const float* __restrict__ queries, uint32_t m, uint32_t* __restrict__ out_idx,
                          float* __restrict__ out_dist) {
  extern __shared__ __attribute__((aligned(16))) float kd_lds[];
  typedef float f32x3 __attribute__((ext_vector_type(3)));
  typedef f32x3 __attribute__((aligned(4))) f32x3_u;
  kd_stage_splits(split, n_split, lds_levels, kd_lds);
  const KdSplits sp{split, kd_lds, lds_levels};
  // grid-stride over queries; every lane stays in the loop (the cooperative scan needs the whole wave)
  const uint32_t rounds = (m + gridDim.x * BLOCK - 1) / (gridDim.x * BLOCK);
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t i = (r * gridDim.x + blockIdx.x) * BLOCK + threadIdx.x;
    const uint32_t ii = i < m ? i : m - 1;
    const f32x3 qv = *(const f32x3_u*)(queries + 3 * (size_t)ii);  // one dwordx3
    const V3 q{qv.x, qv.y, qv.z};
#if defined(A3D_DIAGNOSTICS) && defined(A3D_KD_PROBE) && A3D_KD_PROBE == 2  // phase probe (scripts/build_variant.sh): scan only, pseudo-random leaf
    const uint32_t base = ((i * 2654435761u) >> (32u - max_depth)) * 16u;
#else
    const uint32_t base = kdtree_descend(sp, n, max_depth, q);
#endif
#if defined(A3D_DIAGNOSTICS) && defined(A3D_KD_PROBE) && A3D_KD_PROBE == 1  // phase probe: descent only
    if (i < m) out_idx[i] = base, out_dist[i] = 0.0f;
    continue;
#endif
    uint32_t slot;
    float dist;
    kdtree_scan_leaves_coop(leaves, base, q, &slot, &dist);
    const uint32_t idx = ((const uint32_t*)leaves)[(size_t)slot * 4 + 3];  // the winner's original index
    if (i < m) {
      out_idx[i] = idx;
      out_dist[i] = dist;
    }
  }
}

struct PclGates {
  float max_distance_sqr;
  float dot_reject_max;  // reject iff -1 <= sn.tn <= dot_reject_max  (== acos(sn.tn).abs() > max_normal_angle)
};

// The body of Icp::align's point loop (src/icp/pcl_icp.rs:68-92); grid-stride.
template <int BLOCK>
__device__ __forceinline__ void pcl_point_loop(const KdSplits& sp, const float4* __restrict__ leaves,
                                               const float4* __restrict__ leaf_normals, uint32_t n, uint32_t max_depth,
                                               const float* __restrict__ src_points, const float* __restrict__ src_normals,
                                               uint32_t m, const Pose& T, const PclGates& gates, float (&acc)[GN_ACC]) {
  typedef float f32x3 __attribute__((ext_vector_type(3)));
  typedef f32x3 __attribute__((aligned(4))) f32x3_u;
  // every lane stays in the loop: the cooperative leaf scan needs the whole wave
  const uint32_t rounds = (m + gridDim.x * BLOCK - 1) / (gridDim.x * BLOCK);
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t i = (r * gridDim.x + blockIdx.x) * BLOCK + threadIdx.x;
    const uint32_t ii = i < m ? i : m - 1;
    const f32x3 pv = *(const f32x3_u*)(src_points + 3 * (size_t)ii), nv = *(const f32x3_u*)(src_normals + 3 * (size_t)ii);
    const V3 p = transform_vector(T, V3{pv.x, pv.y, pv.z});
    const uint32_t base = kdtree_descend(sp, n, max_depth, p);
    uint32_t slot;
    float d2;
    kdtree_scan_leaves_coop(leaves, base, p, &slot, &d2);
    // the winner's record and its normal: one 16-byte gather each (the point's line was just scanned)
    const float4 win = leaves[slot], tn4 = leaf_normals[slot];
    const V3 sn = transform_normal(T, V3{nv.x, nv.y, nv.z});
    const V3 tn{tn4.x, tn4.y, tn4.z};
    const float c = dot(sn, tn);
    const bool keep = i < m && !(d2 > gates.max_distance_sqr) && !(c >= -1.0f && c <= gates.dot_reject_max);
    if (keep) {
      const V3 tp{win.x, win.y, win.z};
      const float rr = dot(tp - p, tn);
      const V3 tw = cross(p, tn);
      const float J[6] = {tn.x, tn.y, tn.z, tw.x, tw.y, tw.z};
      gn_step(acc, rr, J);
    }
  }
}

#ifdef A3D_DIAGNOSTICS
// Last-block form: the block that publishes the last partial finishes the iteration (icp_engine.hpp).
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
    pcl_icp_kernel(const float* __restrict__ split, const float4* __restrict__ leaves,
                   const float4* __restrict__ leaf_normals, uint32_t n, uint32_t max_depth, uint32_t lds_levels,
                   const float* __restrict__ src_points, const float* __restrict__ src_normals, uint32_t m,
                   JobState* __restrict__ states, PclGates gates, float* __restrict__ partials,
                   unsigned* __restrict__ counter, SolveArgs solve) {
  extern __shared__ __attribute__((aligned(16))) float kd_lds[];
  const uint32_t n_split = (1u << max_depth) - 1u;
  kd_stage_splits(split, n_split, lds_levels, kd_lds);
  const KdSplits sp{split, kd_lds, lds_levels};
  float acc[GN_ACC];
#pragma unroll
  for (int k = 0; k < GN_ACC; ++k) acc[k] = 0.0f;
  const int status = states->status;
  if (status == A3D_OK) {
    const Pose T = states->pose;
    pcl_point_loop<BLOCK>(sp, leaves, leaf_normals, n, max_depth, src_points, src_normals, m, T, gates, acc);
  }
  SolveArgs sa = solve;
  if (status != A3D_OK) sa.mode = SOLVE_NONE;
  block_finish<GN_ACC, BLOCK / 64>(acc, partials, blockIdx.x, gridDim.x, counter, states, sa, 0);  // no colour term
}

#endif  // A3D_DIAGNOSTICS

// Head-solve form (icp_engine.hpp): the launch of iteration k first finishes iteration k - 1 — every block sums the
// previous launch's partials and runs the solve while its split table is still arriving in LDS — then takes the point
// loop with the resulting pose and stores its partial with plain stores.  State and partials alternate between two
// buffers; job_finish_head applies the last iteration.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
    pcl_icp_head_kernel(const float* __restrict__ split, const float4* __restrict__ leaves,
                        const float4* __restrict__ leaf_normals, uint32_t n, uint32_t max_depth, uint32_t lds_levels,
                        const float* __restrict__ src_points, const float* __restrict__ src_normals, uint32_t m,
                        const JobState* __restrict__ state_in, JobState* __restrict__ state_out, PclGates gates,
                        const float* __restrict__ partials_in, float* __restrict__ partials_out, HeadArgs head) {
  extern __shared__ __attribute__((aligned(16))) float kd_lds[];
  __shared__ uint32_t s_state[JOB_WORDS];
  const uint32_t n_split = (1u << max_depth) - 1u;
  kd_stage_splits(split, n_split, lds_levels, kd_lds);
  const KdSplits sp{split, kd_lds, lds_levels};
  head_advance(state_in, blockIdx.x == 0 ? state_out : nullptr, partials_in, head, 0, s_state, blockIdx.x == 0);
  float acc[GN_ACC];
#pragma unroll
  for (int k = 0; k < GN_ACC; ++k) acc[k] = 0.0f;
  if ((int)s_state[15] == A3D_OK) {  // a failed job stays frozen
    const float* f = (const float*)s_state;
    const Pose T{{f[0], f[1], f[2]}, {f[3], f[4], f[5], f[6]}};
    pcl_point_loop<BLOCK>(sp, leaves, leaf_normals, n, max_depth, src_points, src_normals, m, T, gates, acc);
  }
  float* out = partials_out + (size_t)blockIdx.x * GN_PARTIAL;
  block_reduce_store<GN_ACC, false, BLOCK / 64>(acc, out);
  if (threadIdx.x >= GN_ACC && threadIdx.x < GN_PARTIAL) out[threadIdx.x] = 0.0f;  // no colour term
}

template <typename K>
a3d_status kd_allow_big_lds(K kernel, size_t lds_bytes) {
  if (lds_bytes > 48 * 1024)
    A3D_HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  return A3D_OK;
}

a3d_status check_finite_query_count(uint64_t m) {
  A3D_REQUIRE(m < (1ull << 31), A3D_INVALID_PARAMETER, "too many queries for one call");
  return A3D_OK;
}

}  // namespace

struct a3d_pcl_icp {
  a3d_context* ctx = nullptr;
  a3d_icp_params params;
  a3d_kdtree* tree = nullptr;
  bool target_has_normals = false;
  uint32_t blocks = 0;  // block partials (= grid size of the iteration kernel)
  KdLaunch launch{};
  void* d_block = nullptr;         // one device block (ctx_block_alloc) behind the five pointers below
  size_t block_bytes = 0;
  JobState* d_state = nullptr;     // two buffers: the head-solve form alternates between them
  float* d_partials = nullptr;    // [2][blocks][GN_PARTIAL]
  Pose* d_out_pose = nullptr;     // what job_finish_head leaves for the host (one allocation with d_out_status)
  int32_t* d_out_status = nullptr;
  unsigned* d_counter = nullptr;
  double* d_readback = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // bracket the iteration launches of the last align
  float last_device_ms = 0.f;
};

extern "C" {

static a3d_status kdtree_new_impl(a3d_context* ctx, const float* points, uint64_t n, bool on_device, a3d_kdtree** out) {
  A3D_REQUIRE(ctx && out && points, A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(n > 0 && n < (1ull << 31), A3D_INVALID_PARAMETER,
              "kd-tree needs 1 <= n < 2^31 points (the reference indexes an empty leaf and panics)");
  auto t = std::make_unique<a3d_kdtree>();
  t->ctx = ctx;
  t->n = (uint32_t)n;
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  const char* mode = A3D_DIAG_ENV("A3D_KDTREE_BUILD");  // diagnostics build: "host" = std::stable_sort build (cross-check)
  std::vector<float> host_copy;
  if (mode && !strcmp(mode, "host") && on_device) {  // (cross-check of the device-pointer form: through host memory)
    host_copy.resize((size_t)n * 3);
    A3D_HIP_TRY(hipMemcpy(host_copy.data(), points, (size_t)n * 12, hipMemcpyDeviceToHost));
    points = host_copy.data();
    on_device = false;
  }
  if (!(mode && !strcmp(mode, "host"))) {  // device build: upload the points (unless resident), build on the GPU
    kdtree_shape(t->n, &t->max_depth, &t->n_leaves, &t->n_internal);
    A3D_REQUIRE(t->max_depth <= 23, A3D_INVALID_PARAMETER, "point cloud too large for the implicit kd-tree layout (leaf byte offsets are 32-bit)");
    t->n_split = (uint32_t)((1ull << t->max_depth) - 1);
    t->n_leaf_slots = (1ull << t->max_depth) * 16;
    // the points are staged at the head of the context's kd-tree scratch region (grow-only, reused by every build on
    // this context: no hipMalloc / hipFree — each a device-wide synchronisation — for temporaries)
    void* region = nullptr;
    A3D_TRY(ctx_scratch(ctx, 2, kdtree_build_scratch_bytes(t->n, t->max_depth, ctx->stream), &region));
    const float* d_points = on_device ? points : (const float*)region;
    a3d_status st = A3D_OK;
    if (!on_device && hipMemcpyAsync(region, points, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      set_error("a3d_kdtree_new: upload failed: %s", hipGetErrorString(hipGetLastError()));
      st = A3D_HIP_ERROR;
    }
    if (st == A3D_OK) st = kdtree_build_device(t.get(), d_points);
    hipStreamSynchronize(ctx->stream);
    if (st != A3D_OK) {
      a3d_kdtree_free(t.release());
      return st;
    }
    *out = t.release();
    return A3D_OK;
  }
  std::vector<float4> leaves;
  A3D_TRY(kdtree_build_host(points, t->n, &t->h_split, &leaves, &t->h_slot_of_point, &t->max_depth, &t->n_leaves,
                            &t->n_internal));
  t->n_split = (uint32_t)t->h_split.size();
  t->n_leaf_slots = leaves.size();
  A3D_HIP_TRY(hipMalloc((void**)&t->d_split, std::max<size_t>(1, t->h_split.size()) * sizeof(float)));
  A3D_HIP_TRY(hipMalloc((void**)&t->d_leaves, leaves.size() * sizeof(float4)));
  if (!t->h_split.empty())
    A3D_HIP_TRY(hipMemcpyAsync(t->d_split, t->h_split.data(), t->h_split.size() * sizeof(float),
                               hipMemcpyHostToDevice, ctx->stream));
  A3D_HIP_TRY(hipMemcpyAsync(t->d_leaves, leaves.data(), leaves.size() * sizeof(float4), hipMemcpyHostToDevice,
                             ctx->stream));
  A3D_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *out = t.release();
  return A3D_OK;
}

a3d_status a3d_kdtree_new(a3d_context* ctx, const float* points, uint64_t n, a3d_kdtree** out) {
  return kdtree_new_impl(ctx, points, n, /*on_device=*/false, out);
}

a3d_status a3d_kdtree_new_device(a3d_context* ctx, const void* d_points, uint64_t n, a3d_kdtree** out) {
  return kdtree_new_impl(ctx, (const float*)d_points, n, /*on_device=*/true, out);
}

a3d_status a3d_kdtree_build_path(a3d_kdtree* t, int32_t* out_path) {
  A3D_REQUIRE(t && out_path, A3D_INVALID_PARAMETER, "null argument");
  *out_path = t->built_by;
  return A3D_OK;
}

a3d_status a3d_kdtree_build_ms(a3d_kdtree* t, float* out_ms) {
  A3D_REQUIRE(t && out_ms, A3D_INVALID_PARAMETER, "null argument");
  *out_ms = t->build_ms;
  return A3D_OK;
}

a3d_status a3d_kdtree_nearest_device(a3d_kdtree* t, const void* d_queries, uint64_t m, void* d_indices,
                                     void* d_sqr) {
  A3D_REQUIRE(t && (m == 0 || (d_queries && d_indices && d_sqr)), A3D_INVALID_PARAMETER, "null argument");
  A3D_TRY(check_finite_query_count(m));
  if (m == 0) return A3D_OK;
  A3D_HIP_TRY(hipSetDevice(t->ctx->device));
  const KdLaunch L = kd_launch_config(t, m, "A3D_KD", 1024, 15, 1);
#define A3D_KD_LAUNCH(B)                                                                                      \
  {                                                                                                           \
    A3D_TRY(kd_allow_big_lds(kdtree_nearest_kernel<B>, L.lds_bytes));                                         \
    hipLaunchKernelGGL(kdtree_nearest_kernel<B>, dim3(L.blocks), dim3(B), L.lds_bytes, t->ctx->stream, t->d_split, \
                       t->d_leaves, t->n, t->n_split, t->max_depth, L.lds_levels, (const float*)d_queries,   \
                       (uint32_t)m, (uint32_t*)d_indices, (float*)d_sqr);                                     \
  }
  if (L.block == 256) A3D_KD_LAUNCH(256) else if (L.block == 512) A3D_KD_LAUNCH(512) else A3D_KD_LAUNCH(1024)
#undef A3D_KD_LAUNCH
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

a3d_status a3d_kdtree_nearest(a3d_kdtree* t, const float* queries, uint64_t m, uint64_t* out_indices,
                              float* out_sqr) {
  A3D_REQUIRE(t && (m == 0 || (queries && out_indices && out_sqr)), A3D_INVALID_PARAMETER, "null argument");
  A3D_TRY(check_finite_query_count(m));
  if (m == 0) return A3D_OK;
  A3D_HIP_TRY(hipSetDevice(t->ctx->device));
  hipStream_t s = t->ctx->stream;
  float *d_q = nullptr, *d_d = nullptr;
  uint32_t* d_i = nullptr;
  std::vector<uint32_t> idx32(m);
  a3d_status st = A3D_OK;
  if (hipMalloc((void**)&d_q, m * 12) != hipSuccess || hipMalloc((void**)&d_i, m * 4) != hipSuccess ||
      hipMalloc((void**)&d_d, m * 4) != hipSuccess)
    st = A3D_HIP_ERROR;
  if (st == A3D_OK && hipMemcpyAsync(d_q, queries, m * 12, hipMemcpyHostToDevice, s) != hipSuccess) st = A3D_HIP_ERROR;
  if (st == A3D_OK) st = a3d_kdtree_nearest_device(t, d_q, m, d_i, d_d);
  if (st == A3D_OK && hipMemcpyAsync(idx32.data(), d_i, m * 4, hipMemcpyDeviceToHost, s) != hipSuccess) st = A3D_HIP_ERROR;
  if (st == A3D_OK && hipMemcpyAsync(out_sqr, d_d, m * 4, hipMemcpyDeviceToHost, s) != hipSuccess) st = A3D_HIP_ERROR;
  if (hipStreamSynchronize(s) != hipSuccess) st = A3D_HIP_ERROR;
  hipFree(d_q);
  hipFree(d_i);
  hipFree(d_d);
  if (st == A3D_HIP_ERROR) set_error("a3d_kdtree_nearest: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  if (st != A3D_OK) return st;
  for (uint64_t i = 0; i < m; ++i) out_indices[i] = idx32[i];
  return A3D_OK;
}

// Test hook: tree shape {leaves, internal nodes, max depth}.
a3d_status a3d_kdtree_stats(a3d_kdtree* t, uint64_t out3[3]) {
  A3D_REQUIRE(t && out3, A3D_INVALID_PARAMETER, "null argument");
  out3[0] = t->n_leaves, out3[1] = t->n_internal, out3[2] = t->max_depth;
  return A3D_OK;
}

a3d_status a3d_kdtree_download(a3d_kdtree* t, float* out_split, float* out_leaves, uint64_t out_counts[2]) {
  A3D_REQUIRE(t && out_counts, A3D_INVALID_PARAMETER, "null argument");
  out_counts[0] = t->n_split, out_counts[1] = t->n_leaf_slots;
  hipStream_t s = t->ctx->stream;
  if (out_split && t->n_split)
    A3D_HIP_TRY(hipMemcpyAsync(out_split, t->d_split, (size_t)t->n_split * 4, hipMemcpyDeviceToHost, s));
  if (out_leaves)
    A3D_HIP_TRY(hipMemcpyAsync(out_leaves, t->d_leaves, t->n_leaf_slots * sizeof(float4), hipMemcpyDeviceToHost, s));
  A3D_HIP_TRY(hipStreamSynchronize(s));
  return A3D_OK;
}

a3d_status a3d_kdtree_free(a3d_kdtree* t) {
  if (!t) return A3D_OK;
  if (t->d_block) {  // device build: blocks of the context's (stream-ordered reuse: no synchronisation needed)
    ctx_block_release(t->ctx, t->d_block, t->block_bytes);
    ctx_block_release(t->ctx, t->d_normals_block, t->normals_block_bytes);
  } else {
    hipStreamSynchronize(t->ctx->stream);
    hipFree(t->d_split);
    hipFree(t->d_leaves);
    hipFree(t->d_slot_of_point);
    hipFree(t->d_leaf_normals);
  }
  delete t;
  return A3D_OK;
}

static a3d_status pcl_icp_new_impl(a3d_context* ctx, const a3d_icp_params* params, const a3d_point_cloud_view* target,
                                   bool on_device, a3d_pcl_icp** out) {
  A3D_REQUIRE(ctx && params && target && out && target->points, A3D_INVALID_PARAMETER, "null argument");
  auto icp = std::make_unique<a3d_pcl_icp>();
  icp->ctx = ctx;
  icp->params = *params;
  A3D_TRY(kdtree_new_impl(ctx, target->points, target->len, on_device, &icp->tree));
  a3d_kdtree* t = icp->tree;
  a3d_status st = A3D_OK;
  std::vector<float> host_normals;
  const float* normals = target->normals;
  if (normals && on_device && !t->d_slot_of_point) {  // (diagnostics: host build of a resident cloud)
    host_normals.resize((size_t)t->n * 3);
    if (hipMemcpy(host_normals.data(), normals, (size_t)t->n * 12, hipMemcpyDeviceToHost) != hipSuccess) st = A3D_HIP_ERROR;
    normals = host_normals.data();
  }
  if (normals && t->d_slot_of_point) {  // device build: scatter on the device
    // staged where the build staged the points (dead by now): the head of the context's kd-tree scratch region
    const float* d_n = on_device ? normals : (const float*)ctx->scratch[2];
    if (!on_device && (!d_n || ctx->scratch_size[2] < (size_t)t->n * 12 ||
                       hipMemcpyAsync(ctx->scratch[2], normals, (size_t)t->n * 12, hipMemcpyHostToDevice, ctx->stream) != hipSuccess))
      st = A3D_HIP_ERROR;
    if (st == A3D_OK) st = kdtree_scatter_normals_device(t, d_n);
    hipStreamSynchronize(ctx->stream);
    icp->target_has_normals = true;
  } else if (normals) {  // scatter the target normals into the leaf slots of their points
    std::vector<float4> ln(t->n_leaf_slots, make_float4(0.f, 0.f, 0.f, 0.f));
    for (uint32_t i = 0; i < t->n; ++i)
      ln[t->h_slot_of_point[i]] = make_float4(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2], 0.f);
    if (hipMalloc((void**)&t->d_leaf_normals, ln.size() * sizeof(float4)) != hipSuccess ||
        hipMemcpy(t->d_leaf_normals, ln.data(), ln.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess)
      st = A3D_HIP_ERROR;
    icp->target_has_normals = true;
  }
  // few, fat blocks: the last block sums one partial per block, so the tail grows with the block count; the grid
  // is fixed per Icp object (sized for a source cloud as large as the target) so the partials buffer is too
  icp->launch = kd_launch_config(t, 1ull << 31, "A3D_PCL", 1024, 15, 1);
  icp->blocks = icp->launch.blocks;
  if (st == A3D_OK) {  // the object's device state: one block of the context's (no hipMalloc once a previous Icp was freed)
    auto pad = [](size_t b) { return ((b + 255) / 256) * 256; };
    const size_t state_b = pad(2 * sizeof(JobState)), part_b = pad(2 * (size_t)icp->blocks * GN_PARTIAL * sizeof(float)),
                 pose_b = 256, read_b = pad(GN_PARTIAL * sizeof(double)), count_b = 256;
    char* blk = nullptr;
    st = ctx_block_alloc(ctx, state_b + part_b + pose_b + read_b + count_b, (void**)&blk, &icp->block_bytes);
    if (st == A3D_OK) {
      icp->d_block = blk;
      icp->d_state = (JobState*)blk;
      icp->d_partials = (float*)(blk + state_b);
      icp->d_out_pose = (Pose*)(blk + state_b + part_b);
      icp->d_readback = (double*)(blk + state_b + part_b + pose_b);
      icp->d_counter = (unsigned*)(blk + state_b + part_b + pose_b + read_b);
      if (hipMemsetAsync(icp->d_counter, 0, sizeof(unsigned), ctx->stream) != hipSuccess) st = A3D_HIP_ERROR;
    }
  }
  if (st != A3D_OK) {
    set_error("a3d_pcl_icp_new: HIP failure: %s", hipGetErrorString(hipGetLastError()));
    a3d_pcl_icp_free(icp.release());
    return st;
  }
  *out = icp.release();
  return A3D_OK;
}

a3d_status a3d_pcl_icp_new(a3d_context* ctx, const a3d_icp_params* params, const a3d_point_cloud_view* target,
                           a3d_pcl_icp** out) {
  return pcl_icp_new_impl(ctx, params, target, /*on_device=*/false, out);
}

a3d_status a3d_pcl_icp_new_device(a3d_context* ctx, const a3d_icp_params* params, const a3d_point_cloud_view* d_target,
                                  a3d_pcl_icp** out) {
  return pcl_icp_new_impl(ctx, params, d_target, /*on_device=*/true, out);
}

static a3d_status pcl_upload_source(a3d_pcl_icp* icp, const a3d_point_cloud_view* source, bool on_device, float** d_pts,
                                    float** d_nrm) {
  A3D_HIP_TRY(hipSetDevice(icp->ctx->device));
  // the reference `expect`s both normal sets at align time (pcl_icp.rs:50-58)
  A3D_REQUIRE(icp->target_has_normals, A3D_MISSING_FIELD, "Please, the target point cloud should have normals.");
  A3D_REQUIRE(source->normals, A3D_MISSING_FIELD, "Please, the source point cloud should have normals.");
  A3D_REQUIRE(source->points && source->len > 0 && source->len < (1ull << 31), A3D_INVALID_PARAMETER,
              "bad source cloud");
  // staged in the context's grow-only kd-tree scratch region (idle between tree builds; calls on one context are
  // serialised and this one is host-synchronous): no hipMalloc / hipFree pair — two device-wide synchronisations — per align
  if (on_device) {  // resident source: nothing to stage
    *d_pts = const_cast<float*>(source->points);
    *d_nrm = const_cast<float*>(source->normals);
    return A3D_OK;
  }
  const size_t bytes = source->len * 12, stride = ((bytes + 255) / 256) * 256;
  void* region = nullptr;
  A3D_TRY(ctx_scratch(icp->ctx, 2, 2 * stride, &region));
  *d_pts = (float*)region;
  *d_nrm = (float*)((char*)region + stride);
  A3D_HIP_TRY(hipMemcpyAsync(*d_pts, source->points, bytes, hipMemcpyHostToDevice, icp->ctx->stream));
  A3D_HIP_TRY(hipMemcpyAsync(*d_nrm, source->normals, bytes, hipMemcpyHostToDevice, icp->ctx->stream));
  return A3D_OK;
}

#ifdef A3D_DIAGNOSTICS  // the last-block (ticket) form: cross-check of the head-solve form
static a3d_status pcl_launch_pass(a3d_pcl_icp* icp, const float* d_pts, const float* d_nrm, uint32_t m,
                                  const SolveArgs& solve) {
  PclGates g;
  g.max_distance_sqr = icp->params.max_distance * icp->params.max_distance;
  g.dot_reject_max = acos_gate_threshold(icp->params.max_normal_angle, /*strict=*/true);
  a3d_kdtree* t = icp->tree;
  const KdLaunch& L = icp->launch;
#define A3D_PCL_LAUNCH(B)                                                                                      \
  {                                                                                                            \
    A3D_TRY(kd_allow_big_lds(pcl_icp_kernel<B>, L.lds_bytes));                                                 \
    hipLaunchKernelGGL(pcl_icp_kernel<B>, dim3(L.blocks), dim3(B), L.lds_bytes, icp->ctx->stream, t->d_split,  \
                       t->d_leaves, t->d_leaf_normals, t->n, t->max_depth, L.lds_levels, d_pts, d_nrm, m,      \
                       icp->d_state, g, icp->d_partials, icp->d_counter, solve);                               \
  }
  if (L.block == 256) A3D_PCL_LAUNCH(256) else if (L.block == 512) A3D_PCL_LAUNCH(512) else A3D_PCL_LAUNCH(1024)
#undef A3D_PCL_LAUNCH
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

#endif  // A3D_DIAGNOSTICS

static a3d_status pcl_launch_head_pass(a3d_pcl_icp* icp, const float* d_pts, const float* d_nrm, uint32_t m, uint32_t seq,
                                       const HeadArgs& head) {
  PclGates g;
  g.max_distance_sqr = icp->params.max_distance * icp->params.max_distance;
  g.dot_reject_max = acos_gate_threshold(icp->params.max_normal_angle, /*strict=*/true);
  a3d_kdtree* t = icp->tree;
  const KdLaunch& L = icp->launch;
  const size_t half = (size_t)icp->blocks * GN_PARTIAL;
  const JobState* st_in = icp->d_state + (seq & 1u);
  JobState* st_out = icp->d_state + ((seq + 1u) & 1u);
  const float* part_in = icp->d_partials + (size_t)((seq + 1u) & 1u) * half;  // written by launch seq - 1
  float* part_out = icp->d_partials + (size_t)(seq & 1u) * half;
#define A3D_PCL_LAUNCH(B)                                                                                          \
  {                                                                                                                \
    A3D_TRY(kd_allow_big_lds(pcl_icp_head_kernel<B>, L.lds_bytes));                                                \
    hipLaunchKernelGGL(pcl_icp_head_kernel<B>, dim3(L.blocks), dim3(B), L.lds_bytes, icp->ctx->stream, t->d_split, \
                       t->d_leaves, t->d_leaf_normals, t->n, t->max_depth, L.lds_levels, d_pts, d_nrm, m, st_in,   \
                       st_out, g, part_in, part_out, head);                                                        \
  }
  if (L.block == 256) A3D_PCL_LAUNCH(256) else if (L.block == 512) A3D_PCL_LAUNCH(512) else A3D_PCL_LAUNCH(1024)
#undef A3D_PCL_LAUNCH
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

static a3d_status pcl_icp_align_impl(a3d_pcl_icp* icp, const a3d_point_cloud_view* source, bool on_device, a3d_pose* out_pose) {
  A3D_REQUIRE(icp && source && out_pose, A3D_INVALID_PARAMETER, "null argument");
  float *d_pts = nullptr, *d_nrm = nullptr;
  a3d_status st = pcl_upload_source(icp, source, on_device, &d_pts, &d_nrm);
  hipStream_t s = icp->ctx->stream;
  const uint32_t m = (uint32_t)source->len;
  // Icp::align starts from Transform::eye(): initial_transform is ignored (pcl_icp.rs:59)
  if (!icp->ev0) {
    hipEventCreate(&icp->ev0);
    hipEventCreate(&icp->ev1);
  }
  if (st == A3D_OK) hipEventRecord(icp->ev0, s);
  if (st == A3D_OK) st = launch_job_init(s, icp->d_state, nullptr, 1);
  const char* handoff = A3D_DIAG_ENV("A3D_ICP_HANDOFF");
  const bool head_form = !(handoff && !strcmp(handoff, "ticket"));  // diagnostics build: the last-block form
  icp->d_out_status = (int32_t*)((char*)icp->d_out_pose + 128);
  JobState h;
  Pose h_pose{};
  int32_t h_status = A3D_OK;
  if (head_form) {
    HeadArgs prev{};
    prev.mode = SOLVE_NONE;
    uint32_t seq = 0;
    for (uint64_t it = 0; st == A3D_OK && it < icp->params.max_iterations; ++it, ++seq) {
      st = pcl_launch_head_pass(icp, d_pts, d_nrm, m, seq, prev);
      prev.weight = icp->params.weight, prev.color_weight = 0.0f, prev.mode = SOLVE_PCL_ICP;
      prev.tiles = icp->blocks;
      prev.first_in_level = it == 0, prev.last_in_level = it + 1 == icp->params.max_iterations;
    }
    if (st == A3D_OK)  // the last iteration is still pending: the finish kernel applies it
      st = launch_job_finish_head(s, icp->d_state + (seq & 1u),
                                  icp->d_partials + (size_t)((seq + 1u) & 1u) * icp->blocks * GN_PARTIAL, 0, prev,
                                  icp->d_out_pose, icp->d_out_status, nullptr, 1);
    if (st == A3D_OK) hipEventRecord(icp->ev1, s);
    if (st == A3D_OK && (hipMemcpyAsync(&h_pose, icp->d_out_pose, sizeof(Pose), hipMemcpyDeviceToHost, s) != hipSuccess ||
                         hipMemcpyAsync(&h_status, icp->d_out_status, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess))
      st = A3D_HIP_ERROR;
  } else {
#ifdef A3D_DIAGNOSTICS
    for (uint64_t it = 0; st == A3D_OK && it < icp->params.max_iterations; ++it) {
      SolveArgs sa{};
      sa.weight = icp->params.weight, sa.color_weight = 0.0f;
      sa.mode = SOLVE_PCL_ICP;
      sa.first_in_level = it == 0, sa.last_in_level = it + 1 == icp->params.max_iterations;
      st = pcl_launch_pass(icp, d_pts, d_nrm, m, sa);
    }
    if (st == A3D_OK) hipEventRecord(icp->ev1, s);
    if (st == A3D_OK && hipMemcpyAsync(&h, icp->d_state, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess)
      st = A3D_HIP_ERROR;
#endif
  }
  if (hipStreamSynchronize(s) != hipSuccess && st == A3D_OK) st = A3D_HIP_ERROR;
  if (!head_form) h_pose = h.pose, h_status = h.status;
  if (st == A3D_OK) hipEventElapsedTime(&icp->last_device_ms, icp->ev0, icp->ev1);
  if (st == A3D_HIP_ERROR) set_error("a3d_pcl_icp_align: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  if (st != A3D_OK) return st;
  pose_to_c(h_pose, out_pose);
  if (h_status == A3D_SOLVE_FAILED) set_error("GaussNewton::solve() returned None (count == 0 or Cholesky failed)");
  return (a3d_status)h_status;
}

a3d_status a3d_pcl_icp_align(a3d_pcl_icp* icp, const a3d_point_cloud_view* source, a3d_pose* out_pose) {
  return pcl_icp_align_impl(icp, source, /*on_device=*/false, out_pose);
}

a3d_status a3d_pcl_icp_align_device(a3d_pcl_icp* icp, const a3d_point_cloud_view* d_source, a3d_pose* out_pose) {
  return pcl_icp_align_impl(icp, d_source, /*on_device=*/true, out_pose);
}

a3d_status a3d_pcl_icp_accumulate(a3d_pcl_icp* icp, const a3d_point_cloud_view* source, const a3d_pose* pose,
                                  a3d_gn_state* out_state) {
  A3D_REQUIRE(icp && source && out_state, A3D_INVALID_PARAMETER, "null argument");
  float *d_pts = nullptr, *d_nrm = nullptr;
  a3d_status st = pcl_upload_source(icp, source, /*on_device=*/false, &d_pts, &d_nrm);
  hipStream_t s = icp->ctx->stream;
  Pose h_pose = pose ? pose_from_c(pose) : pose_eye();
  Pose* d_pose = nullptr;
  double sums[GN_PARTIAL];
  if (st == A3D_OK && (hipMalloc((void**)&d_pose, sizeof(Pose)) != hipSuccess ||
                       hipMemcpyAsync(d_pose, &h_pose, sizeof(Pose), hipMemcpyHostToDevice, s) != hipSuccess))
    st = A3D_HIP_ERROR;
  if (st == A3D_OK) st = launch_job_init(s, icp->d_state, d_pose, 1);
  HeadArgs none{};  // the per-iteration launch with nothing to finish at its head: partials in buffer 0
  none.mode = SOLVE_NONE;
  if (st == A3D_OK) st = pcl_launch_head_pass(icp, d_pts, d_nrm, (uint32_t)source->len, 0, none);
  if (st == A3D_OK) st = launch_gn_readback(s, icp->d_partials, (int)icp->blocks, icp->d_readback);
  if (st == A3D_OK && hipMemcpyAsync(sums, icp->d_readback, sizeof(sums), hipMemcpyDeviceToHost, s) != hipSuccess)
    st = A3D_HIP_ERROR;
  if (hipStreamSynchronize(s) != hipSuccess && st == A3D_OK) st = A3D_HIP_ERROR;
  hipFree(d_pose);
  if (st == A3D_HIP_ERROR) set_error("a3d_pcl_icp_accumulate: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  if (st != A3D_OK) return st;
  gn_states_from_sums(sums, out_state, nullptr);
  return A3D_OK;
}

// Instrumentation: device time of the iteration launches of the most recent a3d_pcl_icp_align
// (source already uploaded), between hipEvents on the context stream.
a3d_status a3d_pcl_icp_last_device_ms(a3d_pcl_icp* icp, float* out_ms) {
  A3D_REQUIRE(icp && out_ms, A3D_INVALID_PARAMETER, "null argument");
  *out_ms = icp->last_device_ms;
  return A3D_OK;
}

a3d_status a3d_pcl_icp_free(a3d_pcl_icp* icp) {
  if (!icp) return A3D_OK;
  hipStreamSynchronize(icp->ctx->stream);
  a3d_kdtree_free(icp->tree);
  ctx_block_release(icp->ctx, icp->d_block, icp->block_bytes);
  if (icp->ev0) hipEventDestroy(icp->ev0);
  if (icp->ev1) hipEventDestroy(icp->ev1);
  delete icp;
  return A3D_OK;
}

}  // extern "C"
