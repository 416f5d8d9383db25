// R3dTree::new (src/kdtree.rs:28-58) on the device: dispatch + the SORTING build.
//
// The product's build is the selection build (kdtree_select.hip, round 5).  This file keeps the round-2..4 build — one
// segmented stable sort per level — in the DIAGNOSTICS build only (A3D_KDTREE_BUILD=sorted), as the cross-check:
// The reference sorts the index list of every node by one coordinate (stable sort, depth % 3), splits it at
// len / 2 and recurses until len <= 16.  The shape of that recursion depends on N only, so level d of the
// tree is a set of disjoint index ranges that is known without looking at the data; the sorting build runs
// one SEGMENTED STABLE sort per level over all ranges of that level at once:
//   keys   = coordinate (d % 3) of the point each index refers to, with -0.0 canonicalised to +0.0 so that
//            the radix order equals `partial_cmp` (which calls the two zeros equal) and stability then keeps
//            the parent's order among equal keys, exactly like `slice::sort_by`;
//   values = the indices.
// Short ranges: LDS bitonic sort per range; long ranges: ONE device-wide stable radix sort on 64-bit keys (range
// number << 32 | order-preserving key bits) (kdtree_sort.hip), or rocPRIM's sorts (A3D_KDTREE_SORT=rocprim).
// Bit-identical to the host build (kdtree.hip, std::stable_sort; A3D_KDTREE_BUILD=host) and to the selection build.
#ifdef A3D_DIAGNOSTICS
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#endif

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "kdtree.hpp"

using namespace a3d;

namespace {

#ifdef A3D_DIAGNOSTICS  // kernels of the sorting build
// Range of node `j` (0-based within its level) at `level`: follow the bits of j from the root.
__device__ __forceinline__ void node_range(uint32_t n, uint32_t level, uint32_t j, uint32_t* start, uint32_t* len,
                                           bool* exists) {
  uint32_t s = 0, l = n;
  bool ok = true;
  for (uint32_t t = 0; t < level; ++t) {
    if (l <= 16) {  // an ancestor already is a leaf: this node does not exist
      ok = false;
      break;
    }
    const uint32_t mid = l >> 1;
    const uint32_t right = (j >> (level - 1 - t)) & 1u;
    if (right) {
      s += mid;
      l -= mid;
    } else {
      l = mid;
    }
  }
  *start = s, *len = l, *exists = ok;
}

// begin/end offsets of the ranges to sort at `level` (empty range for nodes that are leaves or do not exist)
__global__ void level_offsets_kernel(uint32_t n, uint32_t level, uint32_t* __restrict__ begin,
                                     uint32_t* __restrict__ end) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (1u << level)) return;
  uint32_t s, l;
  bool ok;
  node_range(n, level, j, &s, &l, &ok);
  const bool sort = ok && l > 16;
  begin[j] = sort ? s : 0u;
  end[j] = sort ? s + l : 0u;
}

// Descends the shape for position i: returns the range that contains i at `level` (or the leaf above it).
__device__ __forceinline__ void position_range(uint32_t n, uint32_t level, uint32_t i, uint32_t* start, uint32_t* len,
                                               uint32_t* path, uint32_t* depth) {
  uint32_t s = 0, l = n, p = 0, d = 0;
  while (d < level && l > 16) {
    const uint32_t mid = l >> 1;
    if (i - s < mid) {
      l = mid;
      p = 2 * p;
    } else {
      s += mid;
      l -= mid;
      p = 2 * p + 1;
    }
    ++d;
  }
  *start = s, *len = l, *path = p, *depth = d;
}

// keys[i] = canonical coordinate k of points[idx[i]]; flags a NaN among the keys that get compared
__global__ void gather_keys_kernel(const float* __restrict__ points, const uint32_t* __restrict__ idx, uint32_t n,
                                   uint32_t level, int k, float* __restrict__ keys, uint32_t* __restrict__ nan_flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = points[3 * (size_t)idx[i] + k];
  keys[i] = v + 0.0f;  // -0.0 -> +0.0
  if (v != v) {
    uint32_t s, l, p, d;
    position_range(n, level, i, &s, &l, &p, &d);
    if (d == level && l > 16) atomicOr(nan_flag, 1u);  // partial_cmp().unwrap() would panic (kdtree.rs:43)
  }
}

// split[heap node] = coordinate k of the point at start + len / 2 of each sorted range (kdtree.rs:47-49)
__global__ void extract_splits_kernel(const float* __restrict__ points, const uint32_t* __restrict__ idx, uint32_t n,
                                      uint32_t level, int k, float* __restrict__ split) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (1u << level)) return;
  uint32_t s, l;
  bool ok;
  node_range(n, level, j, &s, &l, &ok);
  if (ok && l > 16) split[((1u << level) - 1u) + j] = points[3 * (size_t)idx[s + (l >> 1)] + k];
}

// wide levels: key64 = (range number at `level`) << 32 | monotone u32 image of the canonical coordinate
__global__ void gather_keys64_kernel(const float* __restrict__ points, const uint32_t* __restrict__ idx, uint32_t n,
                                     uint32_t level, int k, uint64_t* __restrict__ keys,
                                     uint32_t* __restrict__ nan_flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = points[3 * (size_t)idx[i] + k];
  if (v != v) atomicOr(nan_flag, 1u);  // every position is inside a sorted range at these levels
  const uint32_t u = __float_as_uint(v + 0.0f);
  const uint32_t ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  uint32_t s, l, p, d;
  position_range(n, level, i, &s, &l, &p, &d);
  keys[i] = ((uint64_t)p << 32) | ord;
}

__global__ void fill_leaves_kernel(float4* __restrict__ leaves, uint64_t slots) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < slots) leaves[i] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f);
}

// leaf slot of position i = (path << (max_depth - depth)) * 16 + (i - start); record = {point, index bits}
__global__ void pack_leaves_kernel(const float* __restrict__ points, const uint32_t* __restrict__ idx, uint32_t n,
                                   uint32_t max_depth, float4* __restrict__ leaves,
                                   uint32_t* __restrict__ slot_of_point) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s, l, p, d;
  position_range(n, max_depth, i, &s, &l, &p, &d);
  const uint32_t slot = (p << (max_depth - d)) * 16u + (i - s);
  const uint32_t pi = idx[i];
  leaves[slot] = make_float4(points[3 * (size_t)pi], points[3 * (size_t)pi + 1], points[3 * (size_t)pi + 2],
                             __uint_as_float(pi));
  slot_of_point[pi] = slot;
}

__global__ void iota_kernel(uint32_t* __restrict__ idx, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = i;
}

#endif  // A3D_DIAGNOSTICS (kernels of the sorting build)

__global__ void scatter_normals_kernel(const float* __restrict__ normals, const uint32_t* __restrict__ slot_of_point,
                                       uint32_t n, float4* __restrict__ leaf_normals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  leaf_normals[slot_of_point[i]] = make_float4(normals[3 * (size_t)i], normals[3 * (size_t)i + 1], normals[3 * (size_t)i + 2], 0.f);
}

inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + 255) / 256)); }

}  // namespace

namespace a3d {

// d_points: [n][3] on the device.  Fills t->d_split, t->d_leaves, t->d_slot_of_point (all device).
// Bytes of temporaries kdtree_build_device needs for n points (behind the staged points, see a3d_kdtree_new).
size_t kdtree_build_scratch_bytes(uint32_t n, uint32_t max_depth, hipStream_t s) {
  auto pad = [](size_t b) { return ((b + 255) / 256) * 256; };
  size_t sorted = 0;
#ifdef A3D_DIAGNOSTICS  // the sorting build's temporaries share the region with the selection build's: the larger of the two
  const size_t max_nodes = max_depth ? (1ull << (max_depth - 1)) : 1;
  size_t sort_bytes = 0, wide_bytes = 0;
  if (max_depth > 0) {
    (void)rocprim::segmented_radix_sort_pairs(nullptr, sort_bytes, (float*)nullptr, (float*)nullptr, (uint32_t*)nullptr,
                                              (uint32_t*)nullptr, n, (unsigned)max_nodes, (uint32_t*)nullptr,
                                              (uint32_t*)nullptr, 0, 32, s);
    (void)rocprim::radix_sort_pairs(nullptr, wide_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                    (uint32_t*)nullptr, n, 0, 64, s);
    sort_bytes = std::max(sort_bytes, wide_bytes);
  }
  sorted = 2 * pad((size_t)n * 4) + 2 * pad((size_t)n * 8) + 2 * pad(max_nodes * 4) + 256 + pad(sort_bytes) +
           pad(kdtree_sort_scratch_bytes(n));
#else
  (void)max_depth, (void)s;
#endif
  return std::max(sorted, pad(kdtree_select_scratch_bytes(n))) + pad((size_t)n * 12);
}

// The tree's three arrays in ONE allocation (hipMalloc synchronises the device): leaves first (16-byte records).
static a3d_status kdtree_alloc_arrays(a3d_kdtree* t) {
  const uint64_t n_slots = (1ull << t->max_depth) * 16;
  auto pad256 = [](size_t b) { return ((b + 255) / 256) * 256; };
  const size_t leaves_b = pad256(n_slots * sizeof(float4)), split_b = pad256(std::max<size_t>(1, (size_t)t->n_split) * 4);
  char* block = nullptr;
  A3D_TRY(ctx_block_alloc(t->ctx, leaves_b + split_b + (size_t)t->n * sizeof(uint32_t), (void**)&block, &t->block_bytes));
  t->d_block = block;
  t->d_leaves = (float4*)block;
  t->d_split = (float*)(block + leaves_b);
  t->d_slot_of_point = (uint32_t*)(block + leaves_b + split_b);
  return A3D_OK;
}

#ifdef A3D_DIAGNOSTICS
static a3d_status kdtree_build_device_sorted(a3d_kdtree* t, const float* d_points);
#endif

// The selection build (kdtree_select.hip).  Diagnostics build: A3D_KDTREE_BUILD=sorted runs the sorting build below
// instead (the cross-check of the selection build, as the host build and rocPRIM are of the sorting build).
a3d_status kdtree_build_device(a3d_kdtree* t, const float* d_points) {
  A3D_TRY(kdtree_alloc_arrays(t));
#ifdef A3D_DIAGNOSTICS
  const char* mode = getenv("A3D_KDTREE_BUILD");
  if (mode && !strcmp(mode, "sorted")) {
    t->built_by = 2;
    return kdtree_build_device_sorted(t, d_points);
  }
#endif
  t->built_by = 1;  // (3 once the context's builds add the wide placement launches: kdtree_build_device_select)
  auto pad = [](size_t b) { return ((b + 255) / 256) * 256; };
  char* scratch = (char*)t->ctx->scratch[2] + pad((size_t)t->n * 12);  // behind the points the caller may have staged there
  A3D_REQUIRE(t->ctx->scratch[2] && t->ctx->scratch_size[2] >= pad((size_t)t->n * 12) + kdtree_select_scratch_bytes(t->n),
              A3D_INVALID_PARAMETER, "internal: kd-tree scratch region too small");
  // instrumentation (a3d_kdtree_build_ms): the build's launches between two events on the stream
  hipEvent_t e0 = t->ctx->kd_ev[0], e1 = t->ctx->kd_ev[1];  // (the context's: an event made and destroyed per build cost ~10 us)
  const bool timed = e0 && e1 && hipEventRecord(e0, t->ctx->stream) == hipSuccess;
  const a3d_status st = kdtree_build_device_select(t, d_points, scratch, timed ? e1 : nullptr);
  if (timed && st == A3D_OK) (void)hipEventElapsedTime(&t->build_ms, e0, e1);
  return st;
}

#ifdef A3D_DIAGNOSTICS
static a3d_status kdtree_build_device_sorted(a3d_kdtree* t, const float* d_points) {
  a3d_context* ctx = t->ctx;
  hipStream_t s = ctx->stream;
  const uint32_t n = t->n, D = t->max_depth;
  const uint64_t n_slots = (1ull << D) * 16;
  A3D_HIP_TRY(hipMemsetAsync(t->d_split, 0, std::max<size_t>(1, (size_t)t->n_split) * sizeof(float), s));
  hipLaunchKernelGGL(fill_leaves_kernel, grid_for(n_slots), dim3(256), 0, s, t->d_leaves, n_slots);

  // Levels whose ranges are longer than this use the device-wide sort (no leaf can exist there: len > 16).
  uint32_t wide_len = 4096u;
#ifdef A3D_DIAGNOSTICS
  const char* sort_env = getenv("A3D_KDTREE_SORT");
  const bool use_rocprim = sort_env && !strcmp(sort_env, "rocprim");  // cross-check path; default: kdtree_sort.hip
  if (getenv("A3D_KDTREE_WIDE_LEN")) wide_len = (uint32_t)atoi(getenv("A3D_KDTREE_WIDE_LEN"));
  if (!use_rocprim) wide_len = std::min(wide_len, 4096u);  // the LDS sort holds at most 4096 points per range
#else
  constexpr bool use_rocprim = false;
#endif
  auto max_len = [&](uint32_t level) { return (uint32_t)(((uint64_t)n + (1ull << level) - 1) >> level); };
  auto level_is_wide = [&](uint32_t level) { return max_len(level) > std::max(wide_len, 64u); };

  // scratch: two index buffers, two key buffers, offsets for the widest level, NaN flag, rocPRIM storage
  const size_t max_nodes = D ? (1ull << (D - 1)) : 1;
  size_t sort_bytes = 0;
#ifdef A3D_DIAGNOSTICS
  size_t wide_bytes = 0;
  if (D > 0) {
    // size query with the widest level's segment count
    A3D_HIP_TRY(rocprim::segmented_radix_sort_pairs(nullptr, sort_bytes, (float*)nullptr, (float*)nullptr,
                                                    (uint32_t*)nullptr, (uint32_t*)nullptr, n, (unsigned)max_nodes,
                                                    (uint32_t*)nullptr, (uint32_t*)nullptr, 0, 32, s));
    A3D_HIP_TRY(rocprim::radix_sort_pairs(nullptr, wide_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                          (uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 64, s));
    sort_bytes = std::max(sort_bytes, wide_bytes);
  }
#endif
  auto pad = [](size_t b) { return ((b + 255) / 256) * 256; };
  const size_t hist_bytes = pad(kdtree_sort_scratch_bytes(n));
  const size_t total = 2 * pad((size_t)n * 4) + 2 * pad((size_t)n * 8) + 2 * pad(max_nodes * 4) + 256 +
                       pad(sort_bytes) + hist_bytes;
  // temporaries live in the context's grow-only kd-tree scratch region, BEHIND the points the caller staged there
  char* base = (char*)ctx->scratch[2] + pad((size_t)n * 12);
  A3D_REQUIRE(ctx->scratch[2] && ctx->scratch_size[2] >= pad((size_t)n * 12) + total, A3D_INVALID_PARAMETER,
              "internal: kd-tree scratch region too small");
  uint32_t* idx_a = (uint32_t*)base;
  uint32_t* idx_b = (uint32_t*)(base + pad((size_t)n * 4));
  char* keys_a = base + 2 * pad((size_t)n * 4);
  char* keys_b = keys_a + pad((size_t)n * 8);
  uint32_t* begin = (uint32_t*)(keys_b + pad((size_t)n * 8));
  uint32_t* end = (uint32_t*)((char*)begin + pad(max_nodes * 4));
  uint32_t* nan_flag = (uint32_t*)((char*)end + pad(max_nodes * 4));
  void* sort_tmp = (char*)nan_flag + 256;
  uint32_t* hist = (uint32_t*)((char*)sort_tmp + pad(sort_bytes));

  A3D_HIP_TRY(hipMemsetAsync(nan_flag, 0, 4, s));
  hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(256), 0, s, idx_a, n);
  uint32_t *cur = idx_a, *nxt = idx_b;
  for (uint32_t level = 0; level < D; ++level) {
    const int k = (int)(level % 3);
    const uint32_t nodes = 1u << level;
    size_t bytes = sort_bytes;
    if (!use_rocprim) {
      bool sorted_in_nxt = true;
      if (level_is_wide(level)) {
        hipLaunchKernelGGL(gather_keys64_kernel, grid_for(n), dim3(256), 0, s, d_points, cur, n, level, k,
                           (uint64_t*)keys_a, nan_flag);
        A3D_TRY(kdtree_radix_sort_pairs(s, (uint64_t*)keys_a, (uint64_t*)keys_b, cur, nxt, n, 32 + (int)level, hist,
                                        &sorted_in_nxt));
      } else {
        uint32_t cap_log2 = 5;
        while ((1u << cap_log2) < max_len(level)) ++cap_log2;
        // positions outside this level's ranges (already leaves) keep their order
        A3D_HIP_TRY(hipMemcpyAsync(nxt, cur, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        A3D_TRY(kdtree_sort_ranges(s, d_points, cur, nxt, n, level, k, cap_log2, nan_flag));
      }
      if (sorted_in_nxt) std::swap(cur, nxt);
      hipLaunchKernelGGL(extract_splits_kernel, grid_for(nodes), dim3(256), 0, s, d_points, cur, n, level, k,
                         t->d_split);
      continue;
    }
#ifdef A3D_DIAGNOSTICS  // the same build on rocPRIM's stable sorts
    if (level_is_wide(level)) {
      hipLaunchKernelGGL(gather_keys64_kernel, grid_for(n), dim3(256), 0, s, d_points, cur, n, level, k,
                         (uint64_t*)keys_a, nan_flag);
      A3D_HIP_TRY(rocprim::radix_sort_pairs(sort_tmp, bytes, (uint64_t*)keys_a, (uint64_t*)keys_b, cur, nxt, n, 0,
                                            32 + level, s));
    } else {
      hipLaunchKernelGGL(level_offsets_kernel, grid_for(nodes), dim3(256), 0, s, n, level, begin, end);
      hipLaunchKernelGGL(gather_keys_kernel, grid_for(n), dim3(256), 0, s, d_points, cur, n, level, k, (float*)keys_a,
                         nan_flag);
      // positions outside this level's ranges (already leaves) keep their order
      A3D_HIP_TRY(hipMemcpyAsync(nxt, cur, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
      A3D_HIP_TRY(rocprim::segmented_radix_sort_pairs(sort_tmp, bytes, (float*)keys_a, (float*)keys_b, cur, nxt, n,
                                                      nodes, begin, end, 0, 32, s));
    }
    hipLaunchKernelGGL(extract_splits_kernel, grid_for(nodes), dim3(256), 0, s, d_points, nxt, n, level, k, t->d_split);
    std::swap(cur, nxt);
#endif
  }
  hipLaunchKernelGGL(pack_leaves_kernel, grid_for(n), dim3(256), 0, s, d_points, cur, n, D, t->d_leaves,
                     t->d_slot_of_point);
  uint32_t h_nan = 0;
  A3D_HIP_TRY(hipMemcpyAsync(&h_nan, nan_flag, 4, hipMemcpyDeviceToHost, s));
  A3D_HIP_TRY(hipStreamSynchronize(s));
  A3D_HIP_TRY(hipGetLastError());
  A3D_REQUIRE(!h_nan, A3D_NAN_IN_INPUT,
              "NaN coordinate in kd-tree input (the reference panics in partial_cmp().unwrap())");
  return A3D_OK;
}

#endif  // A3D_DIAGNOSTICS (the sorting build)

a3d_status kdtree_scatter_normals_device(a3d_kdtree* t, const float* d_normals) {
  hipStream_t s = t->ctx->stream;
  A3D_TRY(ctx_block_alloc(t->ctx, t->n_leaf_slots * sizeof(float4), &t->d_normals_block, &t->normals_block_bytes));
  t->d_leaf_normals = (float4*)t->d_normals_block;
  A3D_HIP_TRY(hipMemsetAsync(t->d_leaf_normals, 0, t->n_leaf_slots * sizeof(float4), s));
  hipLaunchKernelGGL(scatter_normals_kernel, grid_for(t->n), dim3(256), 0, s, d_normals, t->d_slot_of_point, t->n,
                     t->d_leaf_normals);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

}  // namespace a3d
