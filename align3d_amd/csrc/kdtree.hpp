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
  void* d_block = nullptr;     // device build: ONE block (ctx_block_alloc) behind d_leaves, d_split, d_slot_of_point
  size_t block_bytes = 0;
  void* d_normals_block = nullptr;  // d_leaf_normals of a device-built tree (ctx_block_alloc)
  size_t normals_block_bytes = 0;
  float* d_split = nullptr;    // [n_split] split values, heap order (root = 0, children 2i+1, 2i+2)
  float4* d_leaves = nullptr;  // [n_leaf_slots]
  float4* d_leaf_normals = nullptr;  // same slots: {nx, ny, nz, 0} (only for Icp targets)
  uint32_t* d_slot_of_point = nullptr;  // [n] leaf slot of each original point index (device build)
  // host copies (host build only, A3D_KDTREE_BUILD=host), used for attaching normals
  std::vector<float> h_split;
  std::vector<uint32_t> h_slot_of_point;  // [n] leaf slot of each original point index
  uint64_t n_leaves = 0, n_internal = 0;
  float build_ms = 0.f;  // instrumentation: device time of the build's launches (selection build)
  int built_by = 0;  // instrumentation: 0 host build, 1 selection build (kdtree_select.hip), 2 sorting build (diagnostics)
};

namespace a3d {

// Host build (stable sort per level, exactly R3dTree::new) -> split table + leaf slots.
// Returns A3D_NAN_IN_INPUT if a coordinate that gets compared is NaN.
a3d_status kdtree_build_host(const float* points, uint32_t n, std::vector<float>* split,
                             std::vector<float4>* leaves, std::vector<uint32_t>* slot_of_point,
                             uint32_t* max_depth, uint64_t* n_leaves, uint64_t* n_internal);

// Shape of the tree for n points (data independent): depth of the deepest leaf, leaf and internal node counts.
void kdtree_shape(uint32_t n, uint32_t* max_depth, uint64_t* n_leaves, uint64_t* n_internal);

// Device build: the selection build (kdtree_select.hip); diagnostics build: also the sorting build (kdtree_build.hip: one
// segmented stable sort per level, A3D_KDTREE_BUILD=sorted); both bit-identical to the host build.
// Expects t->n, max_depth, n_split, n_leaf_slots set; allocates and fills d_split, d_leaves, d_slot_of_point.
a3d_status kdtree_build_device(a3d_kdtree* t, const float* d_points);
// The selection build: `scratch` holds kdtree_select_scratch_bytes(n) bytes.
size_t kdtree_select_scratch_bytes(uint32_t n);
a3d_status kdtree_build_device_select(a3d_kdtree* t, const float* d_points, void* scratch, hipEvent_t done);
// Size of the context scratch region [2] a device build of n points needs: the staged points + the temporaries.
size_t kdtree_build_scratch_bytes(uint32_t n, uint32_t max_depth, hipStream_t s);
// leaf_normals[slot_of_point[i]] = normals[i] (device build)
a3d_status kdtree_scatter_normals_device(a3d_kdtree* t, const float* d_normals);

// The hand-written stable sorts of the device build (kdtree_sort.hip).
size_t kdtree_sort_scratch_bytes(uint32_t n);
a3d_status kdtree_radix_sort_pairs(hipStream_t s, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a,
                                   uint32_t* vals_b, uint32_t n, int end_bit, uint32_t* hist, bool* in_b);
a3d_status kdtree_sort_ranges(hipStream_t s, const float* points, const uint32_t* idx_in, uint32_t* idx_out, uint32_t n,
                              uint32_t level, int k, uint32_t cap_log2, uint32_t* nan_flag);

// ---- device side of R3dTree::nearest (src/kdtree.rs:69-105) -----------------------------------------------------
//
// Shape facts the kernels rely on (checked on the host for every n by kdtree_shape_check, tests/test_abi_cpu.py):
//  * the node reached by descent path `path` at depth d holds  len = (n + bitrev_d(path)) >> d  points
//    (left child = floor(len / 2), right child = ceil(len / 2));
//  * every leaf sits at depth max_depth - 1 or max_depth.
// So a descent is max_depth - 1 unconditional steps plus one conditional step, the coordinate tested at depth d is
// d % 3, and neither `len` nor `path` has to be carried along: node = heap index, path = node + 1 - 2^depth.

// One descent step: `point[dim] < mid` goes left, everything else (NaN included) goes right (kdtree.rs:80-88).
__device__ __forceinline__ uint32_t kd_step(uint32_t node, float qd, float sv) {
  return 2u * node + ((qd < sv) ? 1u : 2u);
}

// Split table: heap levels [0, lds_levels) come from the block's LDS copy, deeper ones from global memory.
struct KdSplits {
  const float* __restrict__ global;
  const float* lds;
  uint32_t lds_levels;
};

// Copies the first min(2^levels - 1, n_split) heap entries of the split table into LDS (all threads of the block;
// `lds` 16-byte aligned).  16-byte loads, four in flight per thread before the first LDS write: the copy costs one or
// two memory round trips instead of one per 4 bytes and thread.
__device__ __forceinline__ void kd_stage_splits(const float* __restrict__ split, uint32_t n_split, uint32_t levels,
                                                float* lds) {
  const uint32_t want = (1u << levels) - 1u, top = n_split < want ? n_split : want;
  const float4* s4 = (const float4*)split;
  float4* l4 = (float4*)lds;
  const uint32_t n4 = top >> 2, B = blockDim.x;
  uint32_t k = threadIdx.x;
  for (; k + 3 * B < n4; k += 4 * B) {
    const float4 a = s4[k], b = s4[k + B], c = s4[k + 2 * B], d = s4[k + 3 * B];
    l4[k] = a, l4[k + B] = b, l4[k + 2 * B] = c, l4[k + 3 * B] = d;
  }
  for (; k < n4; k += B) l4[k] = s4[k];
  for (k = 4 * n4 + threadIdx.x; k < top; k += B) lds[k] = split[k];
  __syncthreads();
}

// `count` unconditional descent steps through the LDS table, the first on coordinate a, then b, c, a, ...
__device__ __forceinline__ uint32_t kd_walk(const float* tab, uint32_t node, uint32_t count, float a, float b, float c) {
  uint32_t k = 0;
  for (; k + 3 <= count; k += 3) {  // dim = depth % 3 without a select
    node = kd_step(node, a, tab[node]);
    node = kd_step(node, b, tab[node]);
    node = kd_step(node, c, tab[node]);
  }
  if (k < count) {
    node = kd_step(node, a, tab[node]);
    if (k + 1 < count) node = kd_step(node, b, tab[node]);
  }
  return node;
}

// (x, y, z) rotated left by r (wave-uniform r in 0..2): the coordinate tested at depth d is rot(q, d % 3).x
__device__ __forceinline__ V3 kd_rot(V3 q, uint32_t r) {
  return r == 0 ? q : (r == 1 ? V3{q.y, q.z, q.x} : V3{q.z, q.x, q.y});
}

typedef float kd_f32x2 __attribute__((ext_vector_type(2)));
typedef float kd_f32x4 __attribute__((ext_vector_type(4)));
typedef kd_f32x2 __attribute__((aligned(4))) kd_f32x2_u;
typedef kd_f32x4 __attribute__((aligned(4))) kd_f32x4_u;

// Three heap levels below `node` fetched in ONE memory round trip: the node's split value, its two children's
// (adjacent heap entries 2n+1, 2n+2) and its four grandchildren's (4n+3 .. 4n+6); the two decisions then pick the
// values that a level-by-level descent would have read.
struct KdTriple {
  float s0;
  kd_f32x2 s1;
  kd_f32x4 s2;
};
__device__ __forceinline__ KdTriple kd_fetch3(const float* __restrict__ tab, uint32_t node) {
  KdTriple t;
  t.s0 = tab[node];
  t.s1 = *(const kd_f32x2_u*)(tab + 2u * (size_t)node + 1u);
  t.s2 = *(const kd_f32x4_u*)(tab + 4u * (size_t)node + 3u);
  return t;
}

// Returns the first slot of the leaf the query falls in.
__device__ __forceinline__ uint32_t kdtree_descend(const KdSplits& sp, uint32_t n, uint32_t max_depth, V3 q) {
  if (max_depth == 0) return 0u;
  const uint32_t last = max_depth - 1;  // depth of the conditional step
  // the heap entry of a node at depth `last` always exists (the table has 2^max_depth - 1 entries; a leaf's entry
  // holds 0 and is ignored), so the last split value is read unconditionally, from whichever table holds that level
  uint32_t node;  // the node reached at depth `last`
  float sv;       // its split value
  if (last < sp.lds_levels) {  // wave-uniform: the whole descent runs out of LDS
    node = kd_walk(sp.lds, 0u, last, q.x, q.y, q.z);
    sv = sp.lds[node];
  } else {
    node = kd_walk(sp.lds, 0u, sp.lds_levels, q.x, q.y, q.z);
    uint32_t level = sp.lds_levels;
    const float* __restrict__ tab = sp.global;
    for (; level + 3 <= last; level += 3) {  // three unconditional levels per round trip
      const V3 c = kd_rot(q, level % 3u);
      const KdTriple t = kd_fetch3(tab, node);
      const uint32_t r0 = (c.x < t.s0) ? 0u : 1u;
      const uint32_t r1 = (c.y < (r0 ? t.s1.y : t.s1.x)) ? 0u : 1u;
      const uint32_t g = 2u * r0 + r1;
      const float s2 = g == 0 ? t.s2.x : (g == 1 ? t.s2.y : (g == 2 ? t.s2.z : t.s2.w));
      node = 8u * node + 7u + 4u * r0 + 2u * r1 + ((c.z < s2) ? 0u : 1u);
    }
    const V3 c = kd_rot(q, level % 3u);
    const uint32_t rest = last - level;  // unconditional levels left before the one at depth `last`: 0, 1 or 2
    if (rest == 0) {
      sv = tab[node];
    } else if (rest == 1) {
      const float s0 = tab[node];
      const kd_f32x2 s1 = *(const kd_f32x2_u*)(tab + 2u * (size_t)node + 1u);
      const uint32_t r0 = (c.x < s0) ? 0u : 1u;
      node = 2u * node + 1u + r0;
      sv = r0 ? s1.y : s1.x;
    } else {
      const KdTriple t = kd_fetch3(tab, node);
      const uint32_t r0 = (c.x < t.s0) ? 0u : 1u;
      const uint32_t r1 = (c.y < (r0 ? t.s1.y : t.s1.x)) ? 0u : 1u;
      const uint32_t g = 2u * r0 + r1;
      node = 4u * node + 3u + g;
      sv = g == 0 ? t.s2.x : (g == 1 ? t.s2.y : (g == 2 ? t.s2.z : t.s2.w));
    }
  }
  // depth `last`: leaf iff len <= 16
  const uint32_t path = node + 1u - (1u << last);
  const uint32_t rev = last ? (__brev(path) >> (32u - last)) : 0u;
  const uint32_t len = (n + rev) >> last;
  const float qd = kd_rot(q, last % 3u).x;
  const bool inner = len > 16u;
  const uint32_t child = kd_step(node, qd, sv) + 1u - (2u << last);  // path at depth max_depth
  return (inner ? child : (path << 1)) * 16u;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// Lane J of every row of 16 lanes, broadcast to the whole row (DPP row_newbcast, gfx90a+): one VALU move, no LDS.
template <int J>
__device__ __forceinline__ float row_bcast_f(float v) { return dpp_f<0x150 + J>(v); }
template <int J>
__device__ __forceinline__ unsigned row_bcast_u(unsigned v) { return dpp_u<0x150 + J>(v); }

constexpr float KD_F32_MAX = 3.402823466e+38f;  // f32::MAX, the scan's initial min_dist (kdtree.rs:91)

// One round of the cooperative leaf scan: the row's 16 lanes hold the 16 slots of the leaf of the query owned by
// lane J of the row; `d` is this lane's candidate distance as its bit pattern (distances are sums of squares, i.e.
// never negative and never -0.0, so unsigned order = float order; NaN / +inf / f32::MAX already mapped to +inf).
// Returns in every lane of the row the row minimum and the FIRST slot that attains it: the reference's
// `dist < min_dist` scan in stored order keeps exactly that one (kdtree.rs:93-100).
__device__ __forceinline__ void kd_row_argmin(unsigned d, unsigned row_shift, unsigned* out_min, unsigned* out_slot) {
  unsigned m = d;
  m = min(m, dpp_u<0x128>(m));  // lane ^ 8 (row_ror:8)
  m = min(m, dpp_u<0x141>(m));  // 7 - lane within each 8 (row_half_mirror)
  m = min(m, dpp_u<0x4E>(m));   // lane ^ 2
  m = min(m, dpp_u<0xB1>(m));   // lane ^ 1
  const unsigned long long eq = __ballot(d == m);  // never empty within a row
  *out_min = m;
  *out_slot = (unsigned)__builtin_ctz((unsigned)(eq >> row_shift));  // lowest lane of MY row (rows above only add higher bits)
}

// Cooperative leaf scan: the 64 queries of a wave are served 4 at a time, 16 lanes per query, one leaf slot per
// lane, so a wave-wide load touches 4 leaves = 8 cache lines instead of 64 lanes x 1 line each.
// Split in two so that a kernel can put other work (the descent of its NEXT query) between the loads and their use:
//   KdScan sc(leaves, base);  sc.issue_first();  ...other work...  sc.finish(q, &slot, &dist);
// base / q belong to THIS lane's query (leaf byte offsets fit 32 bits: max_depth <= 23 is checked on the host); the
// caller re-reads the winning record by slot where it needs it.  Every lane of the wave must be active.
struct KdScan {
  const char* lbase;
  unsigned sub, row_shift, base, base_off, sub_off;
  float4 p[8];

  __device__ __forceinline__ KdScan(const float4* __restrict__ leaves, uint32_t base_) {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    lbase = (const char*)leaves;
    sub = lane & 15u, row_shift = lane & 48u;
    base = base_, base_off = base_ * 16u, sub_off = sub * 16u;  // bytes
  }
  template <int J>
  __device__ __forceinline__ float4 load() const {
    return *(const float4*)(lbase + (size_t)(row_bcast_u<J>(base_off) + sub_off));
  }
  // rounds 0..7: all eight loads in flight
  __device__ __forceinline__ void issue_first() {
    p[0] = load<0>(), p[1] = load<1>(), p[2] = load<2>(), p[3] = load<3>();
    p[4] = load<4>(), p[5] = load<5>(), p[6] = load<6>(), p[7] = load<7>();
  }
  template <int J>
  __device__ __forceinline__ void round(const float4& P, V3 q, unsigned& my_d, unsigned& my_slot) const {
    const V3 c{P.x, P.y, P.z};
    const float df = norm_squared(V3{row_bcast_f<J>(q.x), row_bcast_f<J>(q.y), row_bcast_f<J>(q.z)} - c);
    // `dist < min_dist` from f32::MAX: NaN, +inf and f32::MAX itself can never win; rank them last
    const unsigned d = (df < KD_F32_MAX) ? __float_as_uint(df) : 0x7F800000u;
    unsigned m, s;
    kd_row_argmin(d, row_shift, &m, &s);
    const bool mine = sub == (unsigned)J;
    // all 16 candidates lost: the reference returns slot 0 and f32::MAX (ctz gave 0 already)
    my_d = mine ? min(m, __float_as_uint(KD_F32_MAX)) : my_d;
    my_slot = mine ? s : my_slot;
  }
  __device__ __forceinline__ void finish(V3 q, uint32_t* out_slot, float* out_dist) {
    unsigned my_d = __float_as_uint(KD_F32_MAX), my_slot = 0;
    round<0>(p[0], q, my_d, my_slot), round<1>(p[1], q, my_d, my_slot), round<2>(p[2], q, my_d, my_slot);
    round<3>(p[3], q, my_d, my_slot), round<4>(p[4], q, my_d, my_slot), round<5>(p[5], q, my_d, my_slot);
    round<6>(p[6], q, my_d, my_slot), round<7>(p[7], q, my_d, my_slot);
    p[0] = load<8>(), p[1] = load<9>(), p[2] = load<10>(), p[3] = load<11>();
    p[4] = load<12>(), p[5] = load<13>(), p[6] = load<14>(), p[7] = load<15>();
    round<8>(p[0], q, my_d, my_slot), round<9>(p[1], q, my_d, my_slot), round<10>(p[2], q, my_d, my_slot);
    round<11>(p[3], q, my_d, my_slot), round<12>(p[4], q, my_d, my_slot), round<13>(p[5], q, my_d, my_slot);
    round<14>(p[6], q, my_d, my_slot), round<15>(p[7], q, my_d, my_slot);
    *out_slot = base + my_slot;
    *out_dist = __uint_as_float(my_d);
  }
};

// The scan in one piece.
__device__ __forceinline__ void kdtree_scan_leaves_coop(const float4* __restrict__ leaves, uint32_t base, V3 q,
                                                        uint32_t* out_slot, float* out_dist) {
  KdScan sc(leaves, base);
  sc.issue_first();
  sc.finish(q, out_slot, out_dist);
}

}  // namespace a3d
