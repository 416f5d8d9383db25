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
  // host copies of the build, kept for tests and for attaching normals
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

// Descent of R3dTree::nearest (src/kdtree.rs:69-105) + 16-slot leaf scan.  Returns the winning
// slot (absolute index into leaves) and writes the squared distance.
__device__ __forceinline__ uint32_t kdtree_nearest_slot(const float* __restrict__ split,
                                                        const float4* __restrict__ leaves, uint32_t n,
                                                        uint32_t max_depth, V3 q, float* out_dist,
                                                        float4* out_point) {
  uint32_t len = n, node = 0, path = 0, depth = 0;
  int dim = 0;
  while (len > 16) {
    const float sv = split[node];
    const float qd = dim == 0 ? q.x : (dim == 1 ? q.y : q.z);
    const uint32_t right = (qd < sv) ? 0u : 1u;  // `point[dim] < mid` goes left; NaN goes right
    const uint32_t mid = len >> 1;
    len = right ? len - mid : mid;
    node = 2 * node + 1 + right;
    path = 2 * path + right;
    ++depth;
    dim = dim == 2 ? 0 : dim + 1;
  }
  const uint32_t base = (path << (max_depth - depth)) * 16u;
  float4 pts[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) pts[s] = leaves[base + s];
  float min_dist = 3.402823466e+38f;  // f32::MAX
  uint32_t min_slot = 0;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    // (point - leaf_point).norm_squared() = (dx^2 + dy^2) + dz^2, no contraction
    const V3 dlt = q - V3{pts[s].x, pts[s].y, pts[s].z};
    const float dist = norm_squared(dlt);
    if (dist < min_dist) {
      min_dist = dist;
      min_slot = s;
    }
  }
  *out_dist = min_dist;
  *out_point = leaves[base + min_slot];  // L1-resident re-read instead of a 16-way register select
  return base + min_slot;
}

}  // namespace a3d
