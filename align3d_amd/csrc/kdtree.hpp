// R3dTree (src/kdtree.rs) in an implicit, GPU-friendly layout.
//
// The reference's tree shape is a pure function of N (leaf iff len <= 16, mid = len / 2), so the
// tree needs no child pointers: a node is the pair (heap index, [start, len)) and the only stored
// data per internal node is its split value.  Leaves are padded to 16 slots of float4
// {x, y, z, original index bits}; slot s of the leaf reached by descent path `path` at depth d lives at
// ((path << (max_depth - d)) * 16 + s).  Padding slots hold +inf and can never win the strict `<` scan.
#pragma once
#include "common.hpp"

struct a3d_kdtree {
  a3d_context* ctx = nullptr;
  uint32_t n = 0;
  uint32_t max_depth = 0;      // depth of the deepest leaf
  uint32_t n_split = 0;        // heap entries: 2^max_depth - 1
  uint64_t n_leaf_slots = 0;   // 2^max_depth leaves * 16
  float* d_split = nullptr;    // [n_split] split values, heap order (root = 0, children 2i+1, 2i+2)
  float4* d_leaves = nullptr;  // [n_leaf_slots]
  float4* d_leaf_normals = nullptr;  // same slots: {nx, ny, nz, 0} (only for Icp targets)
  uint32_t* d_slot_of_point = nullptr;  // [n] leaf slot of each original point index (device build)
  // host copies (host build only, A3D_KDTREE_BUILD=host), used for attaching normals
  std::vector<float> h_split;
  std::vector<uint32_t> h_slot_of_point;  // [n] leaf slot of each original point index
  uint64_t n_leaves = 0, n_internal = 0;
};

namespace a3d {

// Host build (stable sort per level, exactly R3dTree::new) -> split table + leaf slots.
// Returns A3D_NAN_IN_INPUT if a coordinate that gets compared is NaN.
a3d_status kdtree_build_host(const float* points, uint32_t n, std::vector<float>* split,
                             std::vector<float4>* leaves, std::vector<uint32_t>* slot_of_point,
                             uint32_t* max_depth, uint64_t* n_leaves, uint64_t* n_internal);

// Shape of the tree for n points (data independent): depth of the deepest leaf, leaf and internal node counts.
void kdtree_shape(uint32_t n, uint32_t* max_depth, uint64_t* n_leaves, uint64_t* n_internal);

// Device build (kdtree_build.hip): one segmented stable radix sort per level; bit-identical to the host build.
// Expects t->n, max_depth, n_split, n_leaf_slots set; fills d_split, d_leaves, d_slot_of_point.
a3d_status kdtree_build_device(a3d_kdtree* t, const float* d_points);
// leaf_normals[slot_of_point[i]] = normals[i] (device build)
a3d_status kdtree_scatter_normals_device(a3d_kdtree* t, const float* d_normals);

// The hand-written stable sorts of the device build (kdtree_sort.hip).
size_t kdtree_sort_scratch_bytes(uint32_t n);
a3d_status kdtree_radix_sort_pairs(hipStream_t s, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a,
                                   uint32_t* vals_b, uint32_t n, int end_bit, uint32_t* hist, bool* in_b);
a3d_status kdtree_sort_ranges(hipStream_t s, const float* points, const uint32_t* idx_in, uint32_t* idx_out, uint32_t n,
                              uint32_t level, int k, uint32_t cap_log2, uint32_t* nan_flag);

// Descent of R3dTree::nearest (src/kdtree.rs:69-105): returns the first slot of the leaf the query falls in.
// `split_top` (nullable) is an LDS copy of the first `top_entries` heap entries of the split table.
__device__ __forceinline__ uint32_t kdtree_descend(const float* __restrict__ split, const float* split_top,
                                                   uint32_t top_entries, uint32_t n, uint32_t max_depth, V3 q) {
  uint32_t len = n, node = 0, path = 0, depth = 0;
  int dim = 0;
  while (len > 16) {
    const float sv = node < top_entries ? split_top[node] : split[node];
    const float qd = dim == 0 ? q.x : (dim == 1 ? q.y : q.z);
    const uint32_t right = (qd < sv) ? 0u : 1u;  // `point[dim] < mid` goes left; NaN goes right
    const uint32_t mid = len >> 1;
    len = right ? len - mid : mid;
    node = 2 * node + 1 + right;
    path = 2 * path + right;
    ++depth;
    dim = dim == 2 ? 0 : dim + 1;
  }
  return (path << (max_depth - depth)) * 16u;
}

// The 16-slot scan of one lane's own leaf (used where a lane works alone: the Icp kernel).
__device__ __forceinline__ uint32_t kdtree_scan_leaf(const float4* __restrict__ leaves, uint32_t base, V3 q,
                                                     float* out_dist, float4* out_point) {
  float4 pts[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) pts[s] = leaves[base + s];
  float min_dist = 3.402823466e+38f;  // f32::MAX
  uint32_t min_slot = 0;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    // (point - leaf_point).norm_squared() = (dx^2 + dy^2) + dz^2, no contraction
    const float dist = norm_squared(q - V3{pts[s].x, pts[s].y, pts[s].z});
    if (dist < min_dist) {  // strict <: the first minimum wins (kdtree.rs:96)
      min_dist = dist;
      min_slot = s;
    }
  }
  *out_dist = min_dist;
  *out_point = leaves[base + min_slot];  // L1-resident re-read instead of a 16-way register select
  return base + min_slot;
}

__device__ __forceinline__ uint32_t kdtree_nearest_slot(const float* __restrict__ split,
                                                        const float4* __restrict__ leaves, uint32_t n,
                                                        uint32_t max_depth, V3 q, float* out_dist,
                                                        float4* out_point) {
  return kdtree_scan_leaf(leaves, kdtree_descend(split, nullptr, 0, n, max_depth, q), q, out_dist, out_point);
}

// Cooperative leaf scan: the 64 queries of a wave are served 4 at a time, 16 lanes per query, one leaf
// slot per lane, so a wave-wide load touches 4 leaves = 8 cache lines instead of 64 lanes x 1 line each
// (the per-lane scan makes ~1000 L1 tag lookups per wave; this makes 128).  Within the 16 lanes the
// minimum of (distance, slot) in lexicographic order is the reference's "first strict minimum".
// In: base/q of THIS lane's query.  Out: this lane's winner (slot index, distance, leaf record).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

__device__ __forceinline__ void kdtree_scan_leaves_coop(const float4* __restrict__ leaves, uint32_t base, V3 q,
                                                        uint32_t* out_slot, float* out_dist, float4* out_point) {
  const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const unsigned sub = lane & 15u, grp16 = lane & 48u;
  float my_d = 3.402823466e+38f;
  unsigned my_slot = 0;
  float4 my_pt = make_float4(0.f, 0.f, 0.f, 0.f);
  // eight rounds at a time: all eight loads are in flight before the first reduction
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float4 p[8];
    float qx[8], qy[8], qz[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int j = half * 8 + r;
      const int src = (int)(grp16 | (unsigned)j);  // the lane of this 16-group whose query is served in round j
      const uint32_t b = (uint32_t)__shfl((int)base, src, 64);
      qx[r] = __shfl(q.x, src, 64), qy[r] = __shfl(q.y, src, 64), qz[r] = __shfl(q.z, src, 64);
      p[r] = leaves[b + sub];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int j = half * 8 + r;
      float d = norm_squared(V3{qx[r], qy[r], qz[r]} - V3{p[r].x, p[r].y, p[r].z});
      // `dist < min_dist` with min_dist starting at f32::MAX: NaN and +inf can never win; rank them last
      if (!(d < 3.402823466e+38f)) d = __builtin_inff();
      unsigned s = sub;
      float4 w = p[r];
#define A3D_ARGMIN_STEP(CTRL)                                                             \
  {                                                                                       \
    const float od = dpp_f<CTRL>(d);                                                      \
    const unsigned os = dpp_u<CTRL>(s);                                                   \
    const float ox = dpp_f<CTRL>(w.x), oy = dpp_f<CTRL>(w.y), oz = dpp_f<CTRL>(w.z), ow = dpp_f<CTRL>(w.w); \
    const bool take = od < d || (od == d && os < s);                                      \
    d = take ? od : d, s = take ? os : s;                                                 \
    w.x = take ? ox : w.x, w.y = take ? oy : w.y, w.z = take ? oz : w.z, w.w = take ? ow : w.w; \
  }
      A3D_ARGMIN_STEP(0x128)  // lane ^ 8 (row_ror:8)
      A3D_ARGMIN_STEP(0x141)  // 7 - lane within each 8 (row_half_mirror)
      A3D_ARGMIN_STEP(0x4E)   // lane ^ 2 (quad_perm [2,3,0,1])
      A3D_ARGMIN_STEP(0xB1)   // lane ^ 1 (quad_perm [1,0,3,2])
#undef A3D_ARGMIN_STEP
      if (sub == (unsigned)j) {
        // all 16 candidates lost (every distance NaN / inf): the reference returns slot 0 and f32::MAX
        const bool none = !(d < 3.402823466e+38f);
        my_d = none ? 3.402823466e+38f : d;
        my_slot = none ? 0u : s;
        my_pt = w;
      }
    }
  }
  // `none` case: w is whichever record the argmin kept; the reference returns leaf entry 0 -> re-read it
  if (my_slot == 0u && !(my_d < 3.402823466e+38f)) my_pt = leaves[base];
  *out_slot = base + my_slot;
  *out_dist = my_d;
  *out_point = my_pt;
}

}  // namespace a3d
