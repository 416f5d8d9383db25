// R3dTree::new (src/kdtree.rs:28-58) on the device by SELECTION instead of by sorting (round 5).
//
// The reference sorts every node's index list by one coordinate (stable, axis = depth % 3), splits it at len / 2 and
// recurses until len <= 16.  Two facts make a much cheaper device build give the same tree, bit for bit:
//   (1) The shape (every node's [start, len)) depends on n only.
//   (2) A stable sort keeps the previous order among equal keys, and the previous order is the parent's sorted order,
//       so by induction the order of a node at depth d after its sort is the LEXICOGRAPHIC order
//           L_d = (x[d % 3], x[(d - 1) % 3], x[(d - 2) % 3], original index)
//       (keys of negative depths dropped; -0.0 == +0.0 as partial_cmp has it; three distinct axes exhaust the
//       coordinates, so older keys can no longer decide anything and the original index breaks what is left).  L_d does
//       not depend on how the node's points were arranged before: WHICH points go left (the len / 2 smallest under
//       L_d), the split value (coordinate of the point of rank len / 2) and the order inside a leaf (L of its parent's
//       depth) are all functions of the point set alone.
// So a level needs no sort: it needs, per node, the point of rank len / 2 under L_d and a partition around it, in any
// order — and only the last few levels, where the leaves' order is fixed, need real sorts.
//
//   wide levels (ranges longer than 2048 points), ONE launch per level, nothing sorted:
//     sel_split_kernel    every point's key falls into one of <= 2048 buckets, linear over the node's bounding box
//                         along the axis (a monotone map: equal keys share a bucket).  The bucket that holds rank
//                         len / 2 is known from the node's histogram (the `plan`); points below it go to the left end
//                         of the range, points above it to the right end (block-wise reservations, any order), the
//                         few in it to a side buffer.  The points routed left / right are at once counted into their
//                         CHILD's histogram along the next axis (the child's box along that axis is the parent's).
//     the resolve step    (resolve_node) one block per node: among the side buffer's points (a bucket: len / buckets
//                         points) the one of rank len / 2 under L_d is found exactly — rounds of finer buckets over the
//                         set's actual [min, max], component of L_d by component, then brute-force counting among the
//                         last <= 128 — and all of them placed; the block writes the split value, the children's boxes,
//                         adds its points to the children's histograms and derives the children's plans.  Round 6: it runs
//                         in the LAST block of the node's split launch to finish (a ticket per node; what it reads from
//                         the other blocks went out as agent-scope stores and memory-side atomics, so no L2 write-back is
//                         needed) instead of in sel_resolve_kernel, a launch of its own — that one remains for levels with
//                         a placement launch (below) and as the diagnostics build's cross-check (A3D_KDTREE_FUSE=0).
//     sel_place_kernel    (a launch between the two on the levels whose ranges can hold more than 4096 points in one
//                         bucket, only once a cloud of the context had such a bucket: a wall facing the camera is tens of
//                         thousands of equal z.)  One resolve block narrows such a set at one CU's rate (30-60 GB/s:
//                         133 us for 63 k points); instead the split kernel histograms the bucket's points one step finer
//                         — by the key inside the bucket's bounds and, for a set of EQUAL keys, by the next component of
//                         L_d — and this launch places all but one finer bucket of them with the whole chip (12 us then).
//   ranges of <= 2048 points: sel_narrow_kernel, ONE launch, one block per range: a bitonic network on 64-bit words
//     `key bits << 32 | position in the range` (the position carries L_{d-1}) per level — at the level a range enters at
//     followed by ranking the (rare) runs of equal keys under L_d, or by the network on the 128-bit words of L_d itself when
//     the range holds long runs —; split values, leaf slots (+inf padding included) and slot_of_point are written from there.
//     The product keeps the network's words in REGISTERS (sel_narrow_kernel<2048, true>, round 6: four consecutive words per
//     thread, partner distances 4 .. 128 over the VALU's lane paths — DPP, v_permlane16/32_swap —, only distances >= 256
//     through LDS; compare-exchange with the direction as data: profiles/round6_kdtree_narrow_stamps.txt); the network with
//     its words in LDS (<2048, false>, A3D_KDTREE_SORTNET=lds: round 5's product) and a selection inside the block (=select)
//     are cross-checks of the diagnostics build.
//
// 500 k points: 11 launches (pack, root histogram + plan, 8 levels, in-block levels) instead of the sorting build's ~110
// (DESIGN.md §5).  Exact for every input: a degenerate cloud (one coordinate constant over a node, thousands of equal points)
// only costs the resolve step more narrowing rounds.  The sorting build (kdtree_build.hip, diagnostics build) is the
// cross-check: same tree, bit for bit (tests/test_gpu_kdtree.py).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kdtree.hpp"

using namespace a3d;

namespace {

constexpr uint32_t NARROW = 2048;      // ranges up to this many points are finished inside one block
constexpr uint32_t MIDDLE_CAP = 4096;  // candidates the resolve block holds in LDS (64 KiB); larger sets are narrowed from global memory first
constexpr uint32_t NB_MAX = 2048;      // most buckets per node and level the kernels' LDS tables hold
constexpr uint32_t NB_DEFAULT = 2048;  // buckets per node and level (fewer on deep levels: ~32 points per bucket)
constexpr uint32_t NSUB = 1024;        // finer buckets inside the median bucket
constexpr uint32_t K1_THREADS = 512, K1_ROUNDS = 4, K1_TILE = K1_THREADS * K1_ROUNDS;
constexpr uint32_t K2_THREADS = 1024;
constexpr uint32_t PACK_THREADS = 256, PACK_ROUNDS = 4, PACK_TILE = PACK_THREADS * PACK_ROUNDS;
#ifndef A3D_CURSOR_STRIDE
#define A3D_CURSOR_STRIDE 32
#endif
constexpr uint32_t CURSOR_STRIDE = A3D_CURSOR_STRIDE;  // words between two of a level's place counters (node, class): each on its own 128-byte line

enum : uint32_t { FLAG_NAN = 0, FLAG_OVERSIZED = 1 };  // words of the flag table the host reads back after the build

struct SelPlan {   // of one node: which bucket holds rank len / 2
  uint32_t bucket, below, count, pad;
};
struct SelBox {    // bounds of the node's points (canonical values), per axis
  float lo[3], hi[3], pad[2];
};
// A node whose median bucket holds more than `wide_cap` points (a degenerate cloud: thousands of equal keys): the split
// kernel also histograms those points one step finer, so that a chip-wide launch (sel_place_kernel) can place all but
// one finer bucket of them before the node's single resolve block takes over.
struct SelWide {
  uint32_t inv_mn0, mx0;  // ~min and max of the set's key bits (zero-initialised tables: both kept by atomicMax)
  uint32_t cursor[3];     // places handed out by sel_place_kernel: left / kept / right
  uint32_t pad[3];
};

__device__ __forceinline__ float canon(float v) { return v + 0.0f; }  // -0.0 -> +0.0 (partial_cmp: equal)
__device__ __forceinline__ uint32_t ord_bits(float v) {                // monotone f32 -> u32 of the canonical value
  const uint32_t u = __float_as_uint(v + 0.0f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float comp(const float4& r, uint32_t a) { return a == 0 ? r.x : (a == 1 ? r.y : r.z); }
// (selects, not indexing: a dynamically indexed array of a struct held in registers goes to scratch memory)
__device__ __forceinline__ float pick3(const float* v, uint32_t a) { return a == 0 ? v[0] : (a == 1 ? v[1] : v[2]); }

// Monotone in v for fixed (lo, hi, nb): rounded subtraction, multiplication by a non-negative constant, clamp and
// truncation all keep order, so v1 <= v2 => bucket(v1) <= bucket(v2) and equal keys share a bucket.  A degenerate or
// non-finite box gives scale 0 (everything in bucket 0); NaN falls into bucket 0 (and is flagged by the caller).
__device__ __forceinline__ float bucket_pos(float v, float lo, float hi, uint32_t nb) {
  const float scale = (hi > lo) ? (float)nb / (hi - lo) : 0.0f;
  const float g = (v - lo) * scale;
  return fminf(fmaxf(g, 0.0f), (float)(nb - 1));
}
__device__ __forceinline__ uint32_t bucket_of(float v, float lo, float hi, uint32_t nb) {
  return (uint32_t)bucket_pos(v, lo, hi, nb);
}
// NSUB finer buckets inside bucket `b`, monotone in v.  The position is NOT clamped to nb - 1 first: the last bucket also
// takes everything up to the box's upper bound (a wall of a room is exactly there), which would all land in sub-bucket 0.
__device__ __forceinline__ uint32_t subbucket_of(float v, float lo, float hi, uint32_t nb, uint32_t b) {
  const float scale = (hi > lo) ? (float)nb / (hi - lo) : 0.0f;
  const float u = ((v - lo) * scale - (float)b) * (float)NSUB;
  return (uint32_t)fminf(fmaxf(u, 0.0f), (float)(NSUB - 1));
}

// [start, len) of node `j` of `level` (kdtree.rs:46-52: mid = len / 2).  `exists` is false below a leaf.
__device__ __forceinline__ void sel_node_range(uint32_t n, uint32_t level, uint32_t j, uint32_t* start, uint32_t* len,
                                               bool* exists) {
  uint32_t s = 0, l = n;
  bool ok = true;
  for (uint32_t t = 0; t < level; ++t) {
    if (l <= 16) {
      ok = false;
      break;
    }
    const uint32_t mid = l >> 1;
    if ((j >> (level - 1 - t)) & 1u) {
      s += mid;
      l -= mid;
    } else {
      l = mid;
    }
  }
  *start = s, *len = l, *exists = ok;
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Exclusive scan of one value per thread over a block of THREADS threads (THREADS / 64 <= 16 waves); `tmp` >= 16 words.
template <uint32_t THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* tmp) {
  const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= (uint32_t)off) incl += up;
  }
  __syncthreads();  // tmp may still be read from a previous call
  if (lane == 63) tmp[w] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t ww = 0; ww < w; ++ww) before += tmp[ww];
  return before + incl - v;
}

// Which of the `nb` <= NB_MAX buckets of `h` (LDS) holds rank `rank`: bucket b with  below(b) <= rank < below(b) + h[b].
// Every thread of the block calls it (NB_MAX / THREADS consecutive buckets each); the one that owns the bucket writes *out.
template <uint32_t THREADS>
__device__ __forceinline__ void plan_from_hist(const uint32_t* h, uint32_t nb, uint32_t rank, SelPlan* out, uint32_t* tmp) {
  constexpr uint32_t PER = 2048 / THREADS;
  static_assert(PER * THREADS == 2048 && PER >= 1, "NB_MAX buckets over the block");
  const uint32_t b0 = PER * threadIdx.x;
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; ++k) v[k] = b0 + k < nb ? h[b0 + k] : 0u, sum += v[k];
  uint32_t ex = block_exclusive_scan<THREADS>(sum, tmp);
#pragma unroll
  for (uint32_t k = 0; k < PER; ++k) {
    if (ex <= rank && rank < ex + v[k]) *out = SelPlan{b0 + k, ex, v[k], 0u};
    ex += v[k];
  }
}

// The same for two tables at once (a node's two children): the two running sums travel as the halves of one 64-bit word
// (each stays below 2^32: at most n points), one block scan instead of two.  `tmp64` >= 16 words of 64 bits.
template <uint32_t THREADS>
__device__ __forceinline__ void plans_from_two_hists(const uint32_t* h0, const uint32_t* h1, uint32_t nb, uint32_t rank0, uint32_t rank1,
                                                     SelPlan* out0, SelPlan* out1, unsigned long long* tmp64) {
  constexpr uint32_t PER = 2048 / THREADS;
  const uint32_t b0 = PER * threadIdx.x, lane = lane_id(), w = threadIdx.x >> 6;
  uint32_t v0[PER], v1[PER];
  unsigned long long sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; ++k) {
    v0[k] = b0 + k < nb ? h0[b0 + k] : 0u, v1[k] = b0 + k < nb ? h1[b0 + k] : 0u;
    sum += ((unsigned long long)v1[k] << 32) | v0[k];
  }
  unsigned long long incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = (unsigned long long)__shfl_up((long long)incl, off, 64);
    if (lane >= (uint32_t)off) incl += up;
  }
  __syncthreads();  // tmp64 may still be read from a previous call
  if (lane == 63) tmp64[w] = incl;
  __syncthreads();
  unsigned long long before = 0;
  for (uint32_t ww = 0; ww < w; ++ww) before += tmp64[ww];
  const unsigned long long ex = before + incl - sum;
  uint32_t e0 = (uint32_t)ex, e1 = (uint32_t)(ex >> 32);
#pragma unroll
  for (uint32_t k = 0; k < PER; ++k) {
    if (e0 <= rank0 && rank0 < e0 + v0[k]) *out0 = SelPlan{b0 + k, e0, v0[k], 0u};
    if (e1 <= rank1 && rank1 < e1 + v1[k]) *out1 = SelPlan{b0 + k, e1, v1[k], 0u};
    e0 += v0[k], e1 += v1[k];
  }
}

// ---- records {x, y, z, index bits} + per-block bounds -------------------------------------------------------------
typedef float sel_f32x3 __attribute__((ext_vector_type(3)));
typedef sel_f32x3 __attribute__((aligned(4))) sel_f32x3_u;

// (round 6: it also zeroes the build's flags, place counters and histogram tables — `zero`, in 16-byte words — which a memset
// launch of 4.7 us did before it: nothing reads them before the next launch)
__global__ void __launch_bounds__(PACK_THREADS)
    sel_pack_kernel(const float* __restrict__ points, uint32_t n, float4* __restrict__ recs, float* __restrict__ partials,
                    uint4* __restrict__ zero, uint32_t zero_words) {
  __shared__ float red[PACK_THREADS / 64][6];
  for (uint32_t i = blockIdx.x * PACK_THREADS + threadIdx.x; i < zero_words; i += gridDim.x * PACK_THREADS) zero[i] = make_uint4(0u, 0u, 0u, 0u);
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  // (all of the thread's loads first, from clamped indices: a load behind `if (i < n)` is followed by s_waitcnt vmcnt(0) at the
  // branch's join, and four of them are four memory round trips one after the other — round 6 found every kernel of this file
  // loading that way: 3-4 us per launch)
  sel_f32x3 pk[PACK_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < PACK_ROUNDS; ++k) {
    const uint32_t i = blockIdx.x * PACK_TILE + k * PACK_THREADS + threadIdx.x;
    pk[k] = *(const sel_f32x3_u*)(points + 3 * (size_t)(i < n ? i : n - 1u));
  }
#pragma unroll
  for (uint32_t k = 0; k < PACK_ROUNDS; ++k) asm volatile("" : "+v"(pk[k].x), "+v"(pk[k].y), "+v"(pk[k].z));  // (keeps the loads up here)
#pragma unroll
  for (uint32_t k = 0; k < PACK_ROUNDS; ++k) {
    const uint32_t i = blockIdx.x * PACK_TILE + k * PACK_THREADS + threadIdx.x;
    if (i < n) {
      const sel_f32x3 p = pk[k];
      recs[i] = make_float4(p.x, p.y, p.z, __uint_as_float(i));
      const float c[3] = {canon(p.x), canon(p.y), canon(p.z)};
#pragma unroll
      for (int a = 0; a < 3; ++a) lo[a] = fminf(lo[a], c[a]), hi[a] = fmaxf(hi[a], c[a]);  // (NaN is skipped)
    }
  }
#pragma unroll
  for (int off = 32; off; off >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
  const uint32_t w = threadIdx.x >> 6;
  if (lane_id() == 0)
    for (int a = 0; a < 3; ++a) red[w][a] = lo[a], red[w][3 + a] = hi[a];
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (uint32_t ww = 1; ww < PACK_THREADS / 64; ++ww)
      v = threadIdx.x < 3 ? fminf(v, red[ww][threadIdx.x]) : fmaxf(v, red[ww][threadIdx.x]);
    partials[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

// The root's box (every block reduces the pack kernel's partials itself) and the root's histogram along x.
// FUSED (round 6): the last block to finish derives the root's plan from the finished histogram (memory-side atomics, read
// back with agent-scope loads) instead of sel_plan0_kernel in a launch of its own; `ticket` is a zeroed word.
template <bool FUSED>
__global__ void __launch_bounds__(K1_THREADS)
    sel_hist0_kernel(const float4* __restrict__ recs, uint32_t n, const float* __restrict__ partials, uint32_t n_partials,
                     uint32_t nb, SelBox* __restrict__ boxes, uint32_t* __restrict__ hist, uint32_t* __restrict__ ticket,
                     SelPlan* __restrict__ plans) {
  extern __shared__ uint32_t h[];
  __shared__ float red[K1_THREADS / 64][6];
  __shared__ float box[6];
  for (uint32_t q = threadIdx.x; q < nb; q += K1_THREADS) h[q] = 0u;
  // (the thread's keys are requested first: their round trip runs under the reduction of the partial bounds)
  float xk[K1_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < K1_ROUNDS; ++k) {
    const uint32_t i = blockIdx.x * K1_TILE + k * K1_THREADS + threadIdx.x;
    xk[k] = recs[i < n ? i : n - 1u].x;  // (unconditional: see sel_pack_kernel)
  }
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (uint32_t b = threadIdx.x; b < n_partials; b += K1_THREADS)
    for (int a = 0; a < 3; ++a) lo[a] = fminf(lo[a], partials[(size_t)b * 6 + a]), hi[a] = fmaxf(hi[a], partials[(size_t)b * 6 + 3 + a]);
#pragma unroll
  for (int off = 32; off; off >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
  if (lane_id() == 0)
    for (int a = 0; a < 3; ++a) red[threadIdx.x >> 6][a] = lo[a], red[threadIdx.x >> 6][3 + a] = hi[a];
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (uint32_t ww = 1; ww < K1_THREADS / 64; ++ww)
      v = threadIdx.x < 3 ? fminf(v, red[ww][threadIdx.x]) : fmaxf(v, red[ww][threadIdx.x]);
    box[threadIdx.x] = v;
    if (blockIdx.x == 0) (threadIdx.x < 3 ? boxes[0].lo[threadIdx.x] : boxes[0].hi[threadIdx.x - 3]) = v;
  }
  __syncthreads();
  const float blo = box[0], bhi = box[3];
#pragma unroll
  for (uint32_t k = 0; k < K1_ROUNDS; ++k) {
    const uint32_t i = blockIdx.x * K1_TILE + k * K1_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[bucket_of(canon(xk[k]), blo, bhi, nb)], 1u);
  }
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < nb; q += K1_THREADS) {
    const uint32_t v = h[q];
    if (v) atomicAdd(&hist[q], v);
  }
  if (FUSED) {
    __shared__ uint32_t is_last;
    __shared__ uint32_t tmp[16];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): this thread's atomics have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (!is_last) return;
    constexpr uint32_t PER = NB_MAX / K1_THREADS;
    uint32_t v[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      const uint32_t q = threadIdx.x + k * K1_THREADS;
      v[k] = q < nb ? __hip_atomic_load(hist + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      const uint32_t q = threadIdx.x + k * K1_THREADS;
      if (q < nb) h[q] = v[k], hist[q] = 0u;  // (zeroed for level 2)
    }
    __syncthreads();
    plan_from_hist<K1_THREADS>(h, nb, n >> 1, &plans[0], tmp);
  }
}

__global__ void __launch_bounds__(K2_THREADS)
    sel_plan0_kernel(uint32_t n, uint32_t nb, uint32_t* __restrict__ hist, SelPlan* __restrict__ plans) {
  __shared__ uint32_t h[NB_MAX];
  __shared__ uint32_t tmp[16];
  for (uint32_t q = threadIdx.x; q < nb; q += K2_THREADS) h[q] = hist[q], hist[q] = 0u;
  __syncthreads();
  plan_from_hist<K2_THREADS>(h, nb, n >> 1, &plans[0], tmp);
}

// One count into an LDS table per active lane; the lanes that share the first active lane's bucket add as one (a
// quantised coordinate puts whole waves into one bucket: an LDS atomic per lane on one address is 64 serial operations).
__device__ __forceinline__ void hist_add_wave(uint32_t* table, uint32_t b, bool valid) {
  const unsigned long long act = __builtin_amdgcn_ballot_w64(valid);
  if (!act) return;
  const uint32_t lead = (uint32_t)__builtin_ctzll(act);
  const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)lead);
  const unsigned long long same = __builtin_amdgcn_ballot_w64(valid && b == bl);
  if (lane_id() == lead) atomicAdd(&table[bl], (uint32_t)__builtin_popcountll(same));
  else if (valid && b != bl) atomicAdd(&table[b], 1u);
}

// The two finer histograms of an oversized median bucket (NSUB buckets each), both monotone under L_d on the sets they
// are used for:  [0] the key itself inside the bucket's bounds;  [1] what decides among EQUAL keys: the previous axis'
// coordinate over the node's box (level >= 1), the original index (level 0: L_0 = (x, index)).
__device__ __forceinline__ uint32_t wide_bucket0(const float4& r, uint32_t a, float lo_a, float hi_a, uint32_t nb, uint32_t bucket) {
  return subbucket_of(canon(comp(r, a)), lo_a, hi_a, nb, bucket);
}
__device__ __forceinline__ uint32_t wide_bucket1(const float4& r, uint32_t level, uint32_t n, float lo_p, float hi_p) {
  if (level == 0) return (uint32_t)(((unsigned long long)__float_as_uint(r.w) * NSUB) / n);
  return bucket_of(canon(comp(r, (level + 2) % 3)), lo_p, hi_p, NSUB);
}
// Which of the two tables decides (1 when every point of the set has the same key) and the finer bucket that holds the
// set's rank `t`; every thread of the block calls it, `hs` (LDS, NSUB words) receives the table.
template <uint32_t THREADS>
__device__ __forceinline__ uint32_t wide_sub_plan(const SelWide* wide, const uint32_t* whist, uint32_t node, uint32_t t,
                                                  uint32_t* hs, SelPlan* out, uint32_t* tmp) {
  const uint32_t which = (~wide[node].inv_mn0 == wide[node].mx0) ? 1u : 0u;
  const uint32_t* g = whist + ((size_t)node * 2 + which) * NSUB;
  for (uint32_t q = threadIdx.x; q < NSUB; q += THREADS) hs[q] = g[q];
  __syncthreads();
  plan_from_hist<THREADS>(hs, NSUB, t, out, tmp);
  __syncthreads();
  return which;
}

// ---- wide levels: route every point of a node below / into / above the bucket of its median -----------------------
template <uint32_t THREADS, bool FOREIGN>
__device__ __forceinline__ void resolve_node(unsigned char* smem, uint32_t node, float4* __restrict__ midbuf, float4* __restrict__ spare,
                                             float4* __restrict__ rout, uint32_t n, uint32_t level, uint32_t nb, uint32_t nb_next,
                                             const SelPlan plan, const SelBox box,
                                             SelPlan* __restrict__ plans_next, SelBox* __restrict__ boxes_next, uint32_t* __restrict__ hist_next,
                                             float* __restrict__ split, uint32_t* __restrict__ flags, uint32_t wide_cap, SelWide* __restrict__ wide,
                                             uint32_t* __restrict__ whist);
__device__ __forceinline__ void store_rec_agent(float4* p, const float4& r);

// FUSED (round 6): the node's resolve step runs in the LAST of the node's blocks to finish instead of in a launch of its own —
// a launch per level less (8 of the 21 of a 500 k build, ~4 us each of launch and first-touch latency for 3 us of work).  The
// hand-over needs no L2 write-back: what the resolve step reads from other blocks — the side buffer's points, the children's
// histograms — is written with agent-scope stores / memory-side atomics, every thread waits for its own to be acknowledged
// (s_waitcnt), and only then the block takes its ticket (the fourth, otherwise unused, place counter of the node).
#ifdef A3D_TAIL_STAMPS
extern __device__ unsigned long long g_sel_stamps[64];
extern __device__ uint32_t g_sel_stamp_level, g_sel_stamp_node;
#endif
template <bool FUSED>
__global__ void __launch_bounds__(K1_THREADS)
    sel_split_kernel(const float4* rin, float4* __restrict__ rout, float4* __restrict__ midbuf, uint32_t n,
                     uint32_t level, uint32_t blocks_per_node, uint32_t nb, uint32_t nb_next,
                     const SelPlan* __restrict__ plans, const SelBox* __restrict__ boxes, uint32_t* __restrict__ cursors,
                     uint32_t* __restrict__ hist_next, uint32_t* __restrict__ flags, uint32_t wide_cap,
                     SelWide* __restrict__ wide, uint32_t* __restrict__ whist, SelPlan* __restrict__ plans_next,
                     SelBox* __restrict__ boxes_next, float* __restrict__ split) {
  extern __shared__ __attribute__((aligned(16))) uint32_t h[];  // [2][nb_next]: the two children's histograms along the next axis; oversized: + [2][NSUB]; FUSED: the resolve step's buffers afterwards
  __shared__ uint32_t w_mm[2];
  __shared__ uint32_t wcnt[3][K1_ROUNDS * (K1_THREADS / 64)];  // per class: (round, wave) counts, then exclusive prefixes
  __shared__ uint32_t base[3];
  const uint32_t node = blockIdx.x / blocks_per_node, part = blockIdx.x % blocks_per_node;
  uint32_t s, l;
  bool exists;
  sel_node_range(n, level, node, &s, &l, &exists);
  const uint32_t tile_lo = part * K1_TILE;
  if (tile_lo >= l) return;
#ifdef A3D_TAIL_STAMPS  // (scripts/sel_stamps.py: when the node's first block started, when its last block took the ticket)
  if (FUSED && threadIdx.x == 0 && part == 0 && level == g_sel_stamp_level && node == g_sel_stamp_node) g_sel_stamps[60] = __builtin_amdgcn_s_memrealtime();
#endif
  const SelPlan plan = plans[node];
  const SelBox box = boxes[node];
  const uint32_t a = level % 3, a2 = (level + 1) % 3;
  const float lo_a = pick3(box.lo, a), hi_a = pick3(box.hi, a), lo_a2 = pick3(box.lo, a2), hi_a2 = pick3(box.hi, a2);
  const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
  const unsigned long long lower = (1ull << lane) - 1ull;
  const bool oversized = plan.count > wide_cap;  // (block-uniform; the launch reserved the LDS for the finer tables)
  if (oversized && part == 0 && threadIdx.x == 0) atomicOr(&flags[FLAG_OVERSIZED], 1u);  // (the host keeps the placement launches on)
  uint32_t* hw = h + 2 * nb_next;
  for (uint32_t q = threadIdx.x; q < 2 * nb_next + (oversized ? 2 * NSUB : 0u); q += K1_THREADS) h[q] = 0u;
  if (threadIdx.x < 2) w_mm[threadIdx.x] = 0u;
  float4 r[K1_ROUNDS];
  uint32_t cls[K1_ROUNDS], rank[K1_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < K1_ROUNDS; ++k) {
    const uint32_t i = tile_lo + k * K1_THREADS + threadIdx.x;
    r[k] = rin[s + (i < l ? i : l - 1u)];  // (unconditional, all four in flight: see sel_pack_kernel)
  }
#pragma unroll
  for (uint32_t k = 0; k < K1_ROUNDS; ++k) {
    const bool valid = tile_lo + k * K1_THREADS + threadIdx.x < l;
    const float key = canon(comp(r[k], a));
    if (valid && key != key) atomicOr(&flags[FLAG_NAN], 1u);  // partial_cmp().unwrap() would panic (kdtree.rs:43)
    const uint32_t b = bucket_of(key, lo_a, hi_a, nb);
    cls[k] = !valid ? 3u : (b < plan.bucket ? 0u : (b > plan.bucket ? 2u : 1u));
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(cls[k] == 0u), m1 = __builtin_amdgcn_ballot_w64(cls[k] == 1u),
                             m2 = __builtin_amdgcn_ballot_w64(cls[k] == 2u);
    const unsigned long long mine = cls[k] == 0u ? m0 : (cls[k] == 1u ? m1 : m2);
    rank[k] = (uint32_t)__builtin_popcountll(mine & lower);
    if (lane == 0) {
      const uint32_t e = k * (K1_THREADS / 64) + w;
      wcnt[0][e] = (uint32_t)__builtin_popcountll(m0), wcnt[1][e] = (uint32_t)__builtin_popcountll(m1),
      wcnt[2][e] = (uint32_t)__builtin_popcountll(m2);
    }
  }
  __syncthreads();
  // waves 0..2: exclusive prefix of class w over the 32 (round, wave) entries; the class total reserves the block's
  // places in the node's left end / side buffer / right end (one atomic with return per class: its round trip to the
  // memory side runs under the histogram work below)
  constexpr uint32_t ENTRIES = K1_ROUNDS * (K1_THREADS / 64);
  static_assert(ENTRIES <= 64, "one wave scans the (round, wave) counts");
  uint32_t reserved = 0;
  if (w < 3) {
    const uint32_t v = lane < ENTRIES ? wcnt[w][lane] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
      if (lane >= (uint32_t)off) incl += up;
    }
    if (lane < ENTRIES) wcnt[w][lane] = incl - v;
    if (lane == 63 && incl) reserved = atomicAdd(&cursors[(node * 4 + w) * CURSOR_STRIDE], incl);
  }
  if (oversized) {  // the median bucket's points, one step finer (see SelWide)
    const uint32_t ap = (level + 2) % 3;
    const float lo_p = pick3(box.lo, ap), hi_p = pick3(box.hi, ap);
    uint32_t inv_mn = 0u, mx = 0u;
#pragma unroll
    for (uint32_t k = 0; k < K1_ROUNDS; ++k) {
      const bool in = cls[k] == 1u;
      hist_add_wave(hw, wide_bucket0(r[k], a, lo_a, hi_a, nb, plan.bucket), in);
      hist_add_wave(hw + NSUB, wide_bucket1(r[k], level, n, lo_p, hi_p), in);
      if (in) {
        const uint32_t kb = ord_bits(comp(r[k], a));
        inv_mn = max(inv_mn, ~kb), mx = max(mx, kb);
      }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
      inv_mn = max(inv_mn, (uint32_t)__shfl_xor((int)inv_mn, off, 64));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
    }
    if (lane == 0 && (inv_mn | mx)) atomicMax(&w_mm[0], inv_mn), atomicMax(&w_mm[1], mx);
  }
  if (nb_next) {  // the points routed left / right, counted into their child's histogram along the next axis
#pragma unroll
    for (uint32_t k = 0; k < K1_ROUNDS; ++k)
      if (cls[k] == 0u || cls[k] == 2u)
        atomicAdd(&h[(cls[k] >> 1) * nb_next + bucket_of(canon(comp(r[k], a2)), lo_a2, hi_a2, nb_next)], 1u);
    __syncthreads();
    uint32_t* g = hist_next + (size_t)(2 * node) * nb_next;  // children 2 node, 2 node + 1: adjacent tables
    for (uint32_t q = threadIdx.x; q < 2 * nb_next; q += K1_THREADS) {
      const uint32_t v = h[q];
      if (v) atomicAdd(&g[q], v);  // (executes at the memory side: contiguous lanes, contiguous words)
    }
  }
  if (w < 3 && lane == 63) base[w] = reserved;
  __syncthreads();
  if (oversized) {
    uint32_t* g = whist + (size_t)node * 2 * NSUB;
    for (uint32_t q = threadIdx.x; q < 2 * NSUB; q += K1_THREADS) {
      const uint32_t v = hw[q];
      if (v) atomicAdd(&g[q], v);
    }
    if (threadIdx.x == 0 && (w_mm[0] | w_mm[1])) atomicMax(&wide[node].inv_mn0, w_mm[0]), atomicMax(&wide[node].mx0, w_mm[1]);
  }
#pragma unroll
  for (uint32_t k = 0; k < K1_ROUNDS; ++k) {
    if (cls[k] == 3u) continue;
    const uint32_t off = base[cls[k]] + wcnt[cls[k]][k * (K1_THREADS / 64) + w] + rank[k];
    if (cls[k] == 0u) rout[s + off] = r[k];
    else if (cls[k] == 2u) rout[s + l - 1u - off] = r[k];
    else if (FUSED) store_rec_agent(&midbuf[s + off], r[k]);
    else midbuf[s + off] = r[k];
  }
  if (FUSED) {
    __shared__ uint32_t is_last;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): this thread's stores and atomics have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t blocks_of_node = (l + K1_TILE - 1) / K1_TILE;  // (the launch's other blocks of the node left at once)
      is_last = __hip_atomic_fetch_add(&cursors[(node * 4 + 3) * CURSOR_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == blocks_of_node - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return;
#ifdef A3D_TAIL_STAMPS
    if (threadIdx.x == 0 && level == g_sel_stamp_level && node == g_sel_stamp_node) g_sel_stamps[61] = __builtin_amdgcn_s_memrealtime();
#endif
    resolve_node<K1_THREADS, true>((unsigned char*)h, node, midbuf, const_cast<float4*>(rin), rout, n, level, nb, nb_next, plan, box, plans_next,
                                   boxes_next, hist_next, split, flags, wide_cap, wide, whist);
  }
}

// Oversized median buckets only (every other block leaves at once): all of the chip places the set's points that fall
// below / above the finer bucket of rank t at their final side of the node's range and counts them into the children's
// histograms, exactly as the split kernel does one step coarser; the points of that finer bucket go to `spare` (this
// node's range of the level's input, dead since the split kernel read it), where the resolve block finds them.
__global__ void __launch_bounds__(K1_THREADS)
    sel_place_kernel(const float4* __restrict__ midbuf, float4* __restrict__ spare, float4* __restrict__ rout, uint32_t n,
                     uint32_t level, uint32_t blocks_per_node, uint32_t nb, uint32_t nb_next,
                     const SelPlan* __restrict__ plans, const SelBox* __restrict__ boxes, uint32_t* __restrict__ hist_next,
                     uint32_t wide_cap, SelWide* __restrict__ wide, const uint32_t* __restrict__ whist) {
  extern __shared__ uint32_t h[];  // [2][nb_next]
  __shared__ uint32_t hs[NSUB];
  __shared__ uint32_t tmp[16];
  __shared__ SelPlan sub_plan;
  __shared__ uint32_t wcnt[3][K1_ROUNDS * (K1_THREADS / 64)];
  __shared__ uint32_t base[3];
  const uint32_t node = blockIdx.x / blocks_per_node, part = blockIdx.x % blocks_per_node;
  const SelPlan plan = plans[node];
  const uint32_t c = plan.count, tile_lo = part * K1_TILE;
  if (c <= wide_cap || tile_lo >= c) return;
  uint32_t s, l;
  bool exists;
  sel_node_range(n, level, node, &s, &l, &exists);
  const uint32_t mid = l >> 1;
  const SelBox box = boxes[node];
  const uint32_t a = level % 3, a2 = (level + 1) % 3, ap = (level + 2) % 3;
  const float lo_a = pick3(box.lo, a), hi_a = pick3(box.hi, a), lo_a2 = pick3(box.lo, a2), hi_a2 = pick3(box.hi, a2),
              lo_p = pick3(box.lo, ap), hi_p = pick3(box.hi, ap);
  const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
  const unsigned long long lower = (1ull << lane) - 1ull;
  for (uint32_t q = threadIdx.x; q < 2 * nb_next; q += K1_THREADS) h[q] = 0u;
  float4 r[K1_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < K1_ROUNDS; ++k) {
    const uint32_t i = tile_lo + k * K1_THREADS + threadIdx.x;
    r[k] = midbuf[s + (i < c ? i : c - 1u)];  // (unconditional, all four in flight: see sel_pack_kernel)
  }
  const uint32_t which = wide_sub_plan<K1_THREADS>(wide, whist, node, mid - plan.below, hs, &sub_plan, tmp);
  const uint32_t star = sub_plan.bucket;
  uint32_t cls[K1_ROUNDS], rank[K1_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < K1_ROUNDS; ++k) {
    const bool valid = tile_lo + k * K1_THREADS + threadIdx.x < c;
    const uint32_t b = which ? wide_bucket1(r[k], level, n, lo_p, hi_p) : wide_bucket0(r[k], a, lo_a, hi_a, nb, plan.bucket);
    cls[k] = !valid ? 3u : (b < star ? 0u : (b > star ? 2u : 1u));  // left / kept / right
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(cls[k] == 0u), m1 = __builtin_amdgcn_ballot_w64(cls[k] == 1u),
                             m2 = __builtin_amdgcn_ballot_w64(cls[k] == 2u);
    const unsigned long long mine = cls[k] == 0u ? m0 : (cls[k] == 1u ? m1 : m2);
    rank[k] = (uint32_t)__builtin_popcountll(mine & lower);
    if (lane == 0) {
      const uint32_t e = k * (K1_THREADS / 64) + w;
      wcnt[0][e] = (uint32_t)__builtin_popcountll(m0), wcnt[1][e] = (uint32_t)__builtin_popcountll(m1),
      wcnt[2][e] = (uint32_t)__builtin_popcountll(m2);
    }
  }
  __syncthreads();
  constexpr uint32_t ENTRIES = K1_ROUNDS * (K1_THREADS / 64);
  uint32_t reserved = 0;
  if (w < 3) {
    const uint32_t v = lane < ENTRIES ? wcnt[w][lane] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
      if (lane >= (uint32_t)off) incl += up;
    }
    if (lane < ENTRIES) wcnt[w][lane] = incl - v;
    if (lane == 63 && incl) reserved = atomicAdd(&wide[node].cursor[w], incl);
  }
  if (nb_next) {
#pragma unroll
    for (uint32_t k = 0; k < K1_ROUNDS; ++k)
      if (cls[k] == 0u || cls[k] == 2u)
        atomicAdd(&h[(cls[k] >> 1) * nb_next + bucket_of(canon(comp(r[k], a2)), lo_a2, hi_a2, nb_next)], 1u);
    __syncthreads();
    uint32_t* g = hist_next + (size_t)(2 * node) * nb_next;
    for (uint32_t q = threadIdx.x; q < 2 * nb_next; q += K1_THREADS) {
      const uint32_t v = h[q];
      if (v) atomicAdd(&g[q], v);
    }
  }
  if (w < 3 && lane == 63) base[w] = reserved;
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < K1_ROUNDS; ++k) {
    if (cls[k] == 3u) continue;
    const uint32_t off = base[cls[k]] + wcnt[cls[k]][k * (K1_THREADS / 64) + w] + rank[k];
    if (cls[k] == 0u) rout[s + plan.below + off] = r[k];  // behind the points the split kernel put at the left end
    else if (cls[k] == 2u) rout[s + mid + off] = r[k];     // from the median's place upwards
    else spare[s + off] = r[k];
  }
}

// L_d as a tuple of u32: (key, previous axis' key, the one before, original index)
struct LKey {
  uint32_t k0, k1, k2, idx;
};
__device__ __forceinline__ LKey lkey_of(const float4& r, uint32_t level) {
  const uint32_t a = level % 3;
  return LKey{ord_bits(comp(r, a)), level >= 1 ? ord_bits(comp(r, (a + 2) % 3)) : 0u,
              level >= 2 ? ord_bits(comp(r, (a + 1) % 3)) : 0u, __float_as_uint(r.w)};
}
__device__ __forceinline__ bool lkey_less(const LKey& p, const LKey& q) {  // (branch-free: see lt128)
  const unsigned long long ph = ((unsigned long long)p.k0 << 32) | p.k1, pl = ((unsigned long long)p.k2 << 32) | p.idx;
  const unsigned long long qh = ((unsigned long long)q.k0 << 32) | q.k1, ql = ((unsigned long long)q.k2 << 32) | q.idx;
  return (ph < qh) | ((ph == qh) & (pl < ql));
}

// (p, q as 128-bit numbers, x the most significant word; branch-free: a chain of `?:` compiles to a ladder of divergent branches)
__device__ __forceinline__ bool lt128(const uint4& p, const uint4& q) {
  const unsigned long long ph = ((unsigned long long)p.x << 32) | p.y, pl = ((unsigned long long)p.z << 32) | p.w;
  const unsigned long long qh = ((unsigned long long)q.x << 32) | q.y, ql = ((unsigned long long)q.z << 32) | q.w;
  return (ph < qh) | ((ph == qh) & (pl < ql));
}
constexpr uint32_t RANK_SMALL = 128;  // a set this small is ranked by brute force (count the smaller ones)
constexpr size_t K2_LDS_BYTES = MIDDLE_CAP * sizeof(float4) + 2 * MIDDLE_CAP * sizeof(uint16_t) + (NSUB + 2 * NB_MAX) * sizeof(uint32_t);

__device__ __forceinline__ uint32_t lkey_comp(const float4& r, uint32_t level, uint32_t c) {
  const LKey k = lkey_of(r, level);
  return c == 0 ? k.k0 : (c == 1 ? k.k1 : (c == 2 ? k.k2 : k.idx));
}
// Monotone map of [mn, mx] onto < NSUB buckets by a shift; mn and mx land in different buckets whenever mn < mx.
__device__ __forceinline__ uint32_t shift_for(uint32_t mn, uint32_t mx) {
  const uint32_t span = mx - mn, bits = span ? 32u - (uint32_t)__builtin_clz(span) : 0u;
  return bits > 10u ? bits - 10u : 0u;  // (NSUB = 2^10)
}
static_assert(NSUB == 1024, "shift_for assumes 2^10 finer buckets");

#ifdef A3D_TAIL_STAMPS
__device__ unsigned long long g_sel_stamps[64];  // (time, tag << 32 | set size) pairs of one resolve block (scripts/sel_stamps.py)
__device__ uint32_t g_sel_stamp_level = 2, g_sel_stamp_node = 0;
#define A3D_SEL_STAMP(tag, cnt)                                                                                   \
  do {                                                                                                            \
    if (threadIdx.x == 0 && level == g_sel_stamp_level && node == g_sel_stamp_node && n_stamp < 32) {             \
      g_sel_stamps[2 * n_stamp] = __builtin_amdgcn_s_memrealtime();                                               \
      g_sel_stamps[2 * n_stamp + 1] = ((unsigned long long)(tag) << 32) | (cnt);                                  \
      ++n_stamp;                                                                                                  \
    }                                                                                                             \
  } while (0)
#else
#define A3D_SEL_STAMP(tag, cnt) do { } while (0)
#endif
#if defined(A3D_TAIL_STAMPS) && defined(A3D_NARROW_PHASE_STAMPS)  // scripts/narrow_stamps.py phases: block 0's network, phase by phase
__device__ uint32_t g_nw_phase_n;
#define A3D_PHASE_STAMP(tag, val)                                                                        \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0 && cap == SLOTS && g_nw_phase_n < 32) {                      \
      g_sel_stamps[2 * g_nw_phase_n] = __builtin_amdgcn_s_memtime();                                     \
      g_sel_stamps[2 * g_nw_phase_n + 1] = ((unsigned long long)(tag) << 32) | (val);                    \
      ++g_nw_phase_n;                                                                                    \
    }                                                                                                    \
  } while (0)
#else
#define A3D_PHASE_STAMP(tag, val) do { } while (0)
#endif

// One block per node: finds the point of rank `t` under L_d among the node's median-bucket points and places them all.
// The set is narrowed round by round — buckets over the actual [min, max] of one component of L_d at a time (a set
// whose points all agree in a component moves on to the next one; the original index, unique, ends it) — first from
// global memory while it is larger than the block's LDS (a degenerate cloud: thousands of equal or nearly equal
// coordinates), then in LDS; the last <= 64 candidates are ranked by brute force.  Exact for every input.
// FOREIGN: the caller is the LAST block of the node's split launch (sel_split_kernel<true>) — the side buffer and the children's
// histograms were written by other blocks of the same launch, possibly through another XCD's L2: they are read with agent-scope
// loads (the writers used agent-scope stores and memory-side atomics, and handed over through the node's ticket).
__device__ __forceinline__ float4 load_rec_agent(const float4* p) {
  const float* f = (const float*)p;
  return make_float4(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                     __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(f + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void store_rec_agent(float4* p, const float4& r) {
  float* f = (float*)p;
  __hip_atomic_store(f, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_store(f + 1, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(f + 2, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_store(f + 3, r.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <uint32_t THREADS, bool FOREIGN>
__device__ __forceinline__ void resolve_node(unsigned char* smem, uint32_t node, float4* __restrict__ midbuf, float4* __restrict__ spare,
                                             float4* __restrict__ rout, uint32_t n, uint32_t level, uint32_t nb, uint32_t nb_next,
                                             const SelPlan plan, const SelBox box,
                                             SelPlan* __restrict__ plans_next, SelBox* __restrict__ boxes_next, uint32_t* __restrict__ hist_next,
                                             float* __restrict__ split, uint32_t* __restrict__ flags, uint32_t wide_cap, SelWide* __restrict__ wide,
                                             uint32_t* __restrict__ whist) {
  float4* rec = (float4*)smem;                         // [MIDDLE_CAP] the candidates once they fit
  uint16_t* list_a = (uint16_t*)(rec + MIDDLE_CAP);    // [MIDDLE_CAP] the current candidate set (indices into rec) ...
  uint16_t* list_b = list_a + MIDDLE_CAP;              // ... and the next
  uint32_t* hsub = (uint32_t*)(list_b + MIDDLE_CAP);   // [NSUB] histogram of a round
  uint32_t* hchild = hsub + NSUB;                      // [2][nb_next] the children's histograms along the next axis
  __shared__ uint32_t tmp[16];
  __shared__ unsigned long long tmp64[16];
  __shared__ SelPlan sub_plan;
  __shared__ uint32_t n_left, n_right, n_keep, s_mn, s_mx;
  __shared__ float split_raw;
#ifdef A3D_TAIL_STAMPS
  uint32_t n_stamp = 0;
#endif
  uint32_t s, l;
  bool exists;
  sel_node_range(n, level, node, &s, &l, &exists);
  const uint32_t mid = l >> 1;
  A3D_SEL_STAMP(0, plan.count);
  const uint32_t a = level % 3, a2 = (level + 1) % 3;
  const float lo_a2 = pick3(box.lo, a2), hi_a2 = pick3(box.hi, a2);
  uint32_t* g = hist_next + (size_t)(2 * node) * nb_next;
  // FOREIGN: the side buffer's first points (all of them, on ordinary data: a bucket holds a few hundred) are fetched next to
  // the histograms instead of a memory round trip later
  constexpr uint32_t PRE = 2;
  float4 pre[PRE];
  if (FOREIGN) {
#pragma unroll
    for (uint32_t k = 0; k < PRE; ++k) {
      const uint32_t i = threadIdx.x + k * THREADS;
      pre[k] = load_rec_agent(midbuf + s + (i < plan.count ? i : (plan.count ? plan.count - 1u : 0u)));  // (unconditional; used only if the set was not narrowed from global memory)
    }
  }
  {  // (all of the thread's loads in flight at once: as a loop of load -> store pairs the agent-scope loads went one memory round trip at a time)
    constexpr uint32_t PER = 2 * NB_MAX / THREADS;
    uint32_t v[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      const uint32_t q = threadIdx.x + k * THREADS;
      const uint32_t qc = q < 2 * nb_next ? q : 0u;  // (unconditional loads: see sel_pack_kernel)
      v[k] = FOREIGN ? __hip_atomic_load(g + qc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g[qc];
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      const uint32_t q = threadIdx.x + k * THREADS;
      if (q < 2 * nb_next) hchild[q] = v[k], g[q] = 0u;  // (zeroed for level + 2)
    }
  }
  if (threadIdx.x == 0) n_left = 0u, n_right = 0u;
  __syncthreads();
  float4* left_out = rout + s + plan.below;  // behind the points the split kernel put at the left end
  float4* right_out = rout + s + mid;        // in front of the points it put at the right end
  auto place = [&](const float4& r, bool right, uint32_t at) {  // a point at its place, counted into its child's histogram
    (right ? right_out : left_out)[at] = r;
    if (nb_next)
      atomicAdd(&hchild[(right ? nb_next : 0u) + bucket_of(canon(comp(r, a2)), lo_a2, hi_a2, nb_next)], 1u);
  };
  auto emit = [&](const float4& r, bool right) { place(r, right, atomicAdd(right ? &n_right : &n_left, 1u)); };
  // min / max of component `cmp` over the set: one LDS atomic per wave
  auto min_max = [&](auto fetch, uint32_t cnt, uint32_t cmp) {
    if (threadIdx.x == 0) s_mn = ~0u, s_mx = 0u;
    __syncthreads();
    uint32_t mn = ~0u, mx = 0u;
    for (uint32_t i = threadIdx.x; i < cnt; i += THREADS) {
      const uint32_t k = lkey_comp(fetch(i), level, cmp);
      mn = min(mn, k), mx = max(mx, k);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
      mn = min(mn, (uint32_t)__shfl_xor((int)mn, off, 64));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
    }
    if (lane_id() == 0) atomicMin(&s_mn, mn), atomicMax(&s_mx, mx);
    __syncthreads();
  };
  // one narrowing round: histogram of `bucket` (a monotone map of the set under L_d onto < NSUB buckets), the bucket of
  // rank t, everything else placed; returns through sub_plan / n_keep.  `keep(r, i, at)` stores a candidate of the next round.
  auto by_component = [&](uint32_t cmp) {  // buckets over the [min, max] min_max has just found for component `cmp`
    const uint32_t mn = s_mn, sh = shift_for(s_mn, s_mx);
    return [=](const float4& r) { return (lkey_comp(r, level, cmp) - mn) >> sh; };
  };
  auto round = [&](auto fetch, auto keep, uint32_t cnt, auto bucket, uint32_t t) {
    for (uint32_t q = threadIdx.x; q < NSUB; q += THREADS) hsub[q] = 0u;
    if (threadIdx.x == 0) n_keep = 0u;
    __syncthreads();
    // (a set with few distinct keys — a quantised coordinate — puts whole waves into one bucket: an LDS atomic per lane
    // on one address is 64 serial operations; the lanes that share the first lane's bucket add as one)
    for (uint32_t i0 = 0; i0 < cnt; i0 += THREADS) {
      const uint32_t i = i0 + threadIdx.x;
      const bool valid = i < cnt;
      const uint32_t b = valid ? bucket(fetch(i)) : 0u;
      const unsigned long long act = __builtin_amdgcn_ballot_w64(valid);
      if (!act) continue;
      const uint32_t lead = (uint32_t)__builtin_ctzll(act);
      const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)lead);
      const unsigned long long same = __builtin_amdgcn_ballot_w64(valid && b == bl);
      if (lane_id() == lead) atomicAdd(&hsub[bl], (uint32_t)__builtin_popcountll(same));
      else if (valid && b != bl) atomicAdd(&hsub[b], 1u);
    }
    __syncthreads();
    plan_from_hist<THREADS>(hsub, NSUB, t, &sub_plan, tmp);
    __syncthreads();
    const uint32_t star = sub_plan.bucket;
    // places are handed out per WAVE (three ballots, one LDS atomic per class and wave-iteration, the lane's rank from
    // the ballot) instead of one atomic per point on three shared counters
    const uint32_t lane = lane_id();
    const unsigned long long lower = (1ull << lane) - 1ull;
    for (uint32_t i0 = 0; i0 < cnt; i0 += THREADS) {  // (block-uniform trip count)
      const uint32_t i = i0 + threadIdx.x;
      const bool valid = i < cnt;
      const float4 r = valid ? fetch(i) : make_float4(0.f, 0.f, 0.f, 0.f);
      const uint32_t b = bucket(r);
      const uint32_t cls = !valid ? 3u : (b == star ? 0u : (b > star ? 2u : 1u));  // keep / left / right
      const unsigned long long m0 = __builtin_amdgcn_ballot_w64(cls == 0u), m1 = __builtin_amdgcn_ballot_w64(cls == 1u),
                               m2 = __builtin_amdgcn_ballot_w64(cls == 2u);
      uint32_t b0 = 0, b1 = 0, b2 = 0;
      if (lane == 0) {
        if (m0) b0 = atomicAdd(&n_keep, (uint32_t)__builtin_popcountll(m0));
        if (m1) b1 = atomicAdd(&n_left, (uint32_t)__builtin_popcountll(m1));
        if (m2) b2 = atomicAdd(&n_right, (uint32_t)__builtin_popcountll(m2));
      }
      b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0), b1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b1),
      b2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b2);
      if (cls == 0u) keep(r, i, b0 + (uint32_t)__builtin_popcountll(m0 & lower));
      else if (cls == 1u) place(r, false, b1 + (uint32_t)__builtin_popcountll(m1 & lower));
      else if (cls == 2u) place(r, true, b2 + (uint32_t)__builtin_popcountll(m2 & lower));
    }
    __syncthreads();
  };
  uint32_t c = plan.count, t = mid - plan.below, cmp = 0;  // the set's rank-t point is the node's median
  // ---- while the set does not fit: rounds from global memory, the survivors ping-pong between the side buffer and
  // this node's range of the previous level's input (dead since the split kernel read it)
  float4* src = midbuf + s;
  float4* dst = spare + s;
  bool fresh = true;  // nothing has narrowed the set yet
  bool untouched = true;  // `src` still is the side buffer as the split launch's blocks wrote it
  if (c > wide_cap) {
    fresh = false;  // sel_place_kernel has placed all but one finer bucket of the set: that bucket is in `spare`
    const uint32_t which = wide_sub_plan<THREADS>(wide, whist, node, t, hsub, &sub_plan, tmp);
    if (threadIdx.x == 0) n_left = sub_plan.below, n_right = c - sub_plan.below - sub_plan.count;
    t -= sub_plan.below, c = sub_plan.count, cmp = which;
    src = spare + s, dst = midbuf + s;
    uint32_t* g2 = whist + (size_t)node * 2 * NSUB;  // the tables serve the node of this index on the next level too
    for (uint32_t q = threadIdx.x; q < 2 * NSUB; q += THREADS) g2[q] = 0u;
    __syncthreads();
    A3D_SEL_STAMP(9, c);
  }
  if (c > MIDDLE_CAP && threadIdx.x == 0) atomicOr(&flags[FLAG_OVERSIZED], 1u);  // (the host's cue for sel_place_kernel)
  while (c > MIDDLE_CAP) {
    const bool foreign = FOREIGN && untouched;  // (the survivors of a round are this block's own stores)
    auto fetch = [&](uint32_t i) { return foreign ? load_rec_agent(src + i) : src[i]; };
    min_max(fetch, c, cmp);
    A3D_SEL_STAMP(1, c);
    if (s_mn == s_mx) {  // all agree in this component of L_d: the next one decides
      ++cmp;
      continue;
    }
    fresh = false;
    round(fetch, [&](const float4& r, uint32_t, uint32_t at) { dst[at] = r; }, c, by_component(cmp), t);
    t -= sub_plan.below, c = sub_plan.count;
    A3D_SEL_STAMP(2, c);
    float4* sw = src;
    src = dst, dst = sw;
    untouched = false;
  }
  // ---- in LDS
  if (FOREIGN && untouched) {  // (nothing narrowed from global memory: c = plan.count <= MIDDLE_CAP, the points as the split launch wrote them)
#pragma unroll
    for (uint32_t k = 0; k < PRE; ++k) {
      const uint32_t i = threadIdx.x + k * THREADS;
      if (i < c) rec[i] = pre[k], list_a[i] = (uint16_t)i;
    }
    for (uint32_t i = threadIdx.x + PRE * THREADS; i < c; i += THREADS) rec[i] = load_rec_agent(src + i), list_a[i] = (uint16_t)i;
  } else {  // (all of the thread's loads in flight together: see sel_pack_kernel)
    constexpr uint32_t PER_FILL = MIDDLE_CAP / THREADS;
    float4 got[PER_FILL];
    const uint32_t last = c ? c - 1u : 0u;
#pragma unroll
    for (uint32_t k = 0; k < PER_FILL; ++k) {
      const uint32_t i = threadIdx.x + k * THREADS;
      got[k] = src[i < c ? i : last];
    }
#pragma unroll
    for (uint32_t k = 0; k < PER_FILL; ++k) asm volatile("" : "+v"(got[k].x), "+v"(got[k].y), "+v"(got[k].z), "+v"(got[k].w));
#pragma unroll
    for (uint32_t k = 0; k < PER_FILL; ++k) {
      const uint32_t i = threadIdx.x + k * THREADS;
      if (i < c) rec[i] = got[k], list_a[i] = (uint16_t)i;
    }
  }
  __syncthreads();
  A3D_SEL_STAMP(3, c);
  uint16_t *cur = list_a, *nxt = list_b;
  while (c > RANK_SMALL) {
    auto fetch = [&](uint32_t i) { return rec[cur[i]]; };
    if (fresh) {
      // the set as the split kernel left it — one bucket of the node's box along the axis: its first round takes its
      // buckets from the bucket's bounds (NSUB finer ones) instead of a pass for the set's own [min, max]; a set of
      // equal keys stays whole and goes on below
      fresh = false;
      const float lo_a = pick3(box.lo, a), hi_a = pick3(box.hi, a);
      const uint32_t bk = plan.bucket;
      round(fetch, [&](const float4&, uint32_t i, uint32_t at) { nxt[at] = cur[i]; }, c,
            [=](const float4& r) { return wide_bucket0(r, a, lo_a, hi_a, nb, bk); }, t);
      t -= sub_plan.below, c = sub_plan.count;
      A3D_SEL_STAMP(5, c);
      uint16_t* sw = cur;
      cur = nxt, nxt = sw;
      continue;
    }
    min_max(fetch, c, cmp);
    A3D_SEL_STAMP(4, c);
    if (s_mn == s_mx) {
      ++cmp;
      continue;
    }
    round(fetch, [&](const float4&, uint32_t i, uint32_t at) { nxt[at] = cur[i]; }, c, by_component(cmp), t);
    t -= sub_plan.below, c = sub_plan.count;
    A3D_SEL_STAMP(5, c);
    uint16_t* sw = cur;
    cur = nxt, nxt = sw;
  }
  A3D_SEL_STAMP(6, c);
  // exact rank under L_d among the last <= RANK_SMALL candidates, by the whole block: PARTS adjacent lanes share a candidate,
  // each counts the smaller ones among every PARTS-th candidate (keys from LDS, the same address for the lanes of a part:
  // broadcast reads), the partial counts meet by lane exchange.  (Rounds 5: one wave, a candidate per lane, the others' keys
  // by v_readlane — 63 candidates took it 4.6 us; a set of 65 .. 128 went through a narrowing round of 2.6 us first.)
  {
    constexpr uint32_t PARTS = THREADS / RANK_SMALL;
    static_assert(PARTS * RANK_SMALL == THREADS && PARTS <= 64 && (PARTS & (PARTS - 1)) == 0, "lanes of a candidate sit in one wave");
    uint4* keys = (uint4*)hsub;  // (the rounds' table is free now: NSUB words = 256 keys)
    static_assert(RANK_SMALL * sizeof(uint4) <= NSUB * sizeof(uint32_t), "keys fit the round's table");
    if (threadIdx.x < c) {
      const LKey lk = lkey_of(rec[cur[threadIdx.x]], level);
      keys[threadIdx.x] = make_uint4(lk.k0, lk.k1, lk.k2, lk.idx);
    }
    __syncthreads();
    const uint32_t e = threadIdx.x / PARTS, part = threadIdx.x % PARTS;
    uint32_t rnk = 0;
    if (e < c) {
      const uint4 me = keys[e];
      for (uint32_t f = part; f < c; f += PARTS) rnk += lt128(keys[f], me) ? 1u : 0u;
    }
#pragma unroll
    for (uint32_t off = 1; off < PARTS; off <<= 1) rnk += (uint32_t)__shfl_xor((int)rnk, (int)off, 64);
    // (places by one LDS atomic per candidate; per-wave ballots and an unrolled counting loop were measured slower here)
    if (e < c && part == 0) {
      const float4 r = rec[cur[e]];
      emit(r, rnk >= t);
      if (rnk == t) split_raw = comp(r, a);  // the point of rank len / 2: `points[mid][k]` (kdtree.rs:47-49)
    }
  }
  __syncthreads();
  A3D_SEL_STAMP(7, c);
  if (threadIdx.x == 0) split[((1u << level) - 1u) + node] = split_raw;
  if (!nb_next) return;  // the children are finished by the narrow kernel: no boxes or plans needed
  if (threadIdx.x < 2) {  // the children's boxes: the parent's, cut at the split value along the split axis
    SelBox cb = box;
    const float m = canon(split_raw);
#pragma unroll
    for (uint32_t ax = 0; ax < 3; ++ax) {
      if (ax == a && threadIdx.x == 0) cb.hi[ax] = m;
      if (ax == a && threadIdx.x == 1) cb.lo[ax] = m;
    }
    boxes_next[2 * node + threadIdx.x] = cb;
  }
  plans_from_two_hists<THREADS>(hchild, hchild + nb_next, nb_next, mid >> 1, (l - mid) >> 1, &plans_next[2 * node], &plans_next[2 * node + 1], tmp64);
  A3D_SEL_STAMP(8, c);
}

__global__ void __launch_bounds__(K2_THREADS)
    sel_resolve_kernel(float4* __restrict__ midbuf, float4* __restrict__ spare, float4* __restrict__ rout, uint32_t n,
                       uint32_t level, uint32_t nb, uint32_t nb_next, const SelPlan* __restrict__ plans, const SelBox* __restrict__ boxes,
                       SelPlan* __restrict__ plans_next, SelBox* __restrict__ boxes_next, uint32_t* __restrict__ hist_next,
                       float* __restrict__ split, uint32_t* __restrict__ flags, uint32_t wide_cap, SelWide* __restrict__ wide,
                       uint32_t* __restrict__ whist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  resolve_node<K2_THREADS, false>(smem, blockIdx.x, midbuf, spare, rout, n, level, nb, nb_next, plans[blockIdx.x], boxes[blockIdx.x], plans_next, boxes_next, hist_next,
                                  split, flags, wide_cap, wide, whist);
}

// ---- ranges of <= NARROW points: all remaining levels in one block ---------------------------------------------------
// A sort word: `key bits << 32 | position` — the position makes every word unique and carries the previous order, so
// the (unstable) network yields exactly the stable order.  Padding sorts last.
typedef unsigned long long Word;
constexpr Word WORD_PADDING = ~0ull;

// Compare-exchange of two 64-bit words WITHOUT a mask passing through the scalar unit.  `if ((hi < lo) == up) swap` compiles to
// v_cmp_lt_u64 -> s_xor_b64 with the lanes' direction mask -> s_nop -> v_cndmask x 4, and the VALU -> SGPR -> SALU -> SGPR -> VALU
// round trip costs a wave ~26 cycles each time (scripts/valu_rate.hip cmp: 39 cycles per compare + select through s_xor against 13
// with the mask forwarded inside the VALU) — that, not the LDS, bound the network (round 6: the same 24 400 cycles for a 2048-word
// sort with its words in LDS and with them in registers).  Here the direction is DATA: flip = 0 (ascending: lo keeps the smaller
// word) or ~0 per lane, the compare's mask is consumed at once by one v_cndmask that turns it into the swap mask m, and the
// words move by v_bfi_b32.
// (the direction word goes through an empty asm first: knowing that it is 0 or ~0, hipcc folds all of this back into mask
// arithmetic on the scalar unit)
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ uint32_t swap_mask(unsigned long long a, unsigned long long b, uint32_t flip, uint32_t nflip) { return a < b ? nflip : flip; }
__device__ __forceinline__ unsigned long long bfi64(uint32_t m, unsigned long long take, unsigned long long keep) {
  const uint32_t lo = ((uint32_t)take & m) | ((uint32_t)keep & ~m), hi = ((uint32_t)(take >> 32) & m) | ((uint32_t)(keep >> 32) & ~m);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void compare_exchange(unsigned long long& lo, unsigned long long& hi, uint32_t flip, uint32_t nflip) {
  const uint32_t m = swap_mask(hi, lo, flip, nflip);
  const unsigned long long nl = bfi64(m, hi, lo), nh = bfi64(m, lo, hi);
  lo = nl, hi = nh;
}

// The word of lane ^ M of the same wave, over the VALU's lane paths (checked lane by lane on the card: scripts/lane_xor_probe.hip).
template <int M>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v) {
  if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false);       // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
  else if constexpr (M == 4) {  // row_shl:4 into banks 0 and 2 of every row, row_shr:4 into banks 1 and 3
    const int r = __builtin_amdgcn_mov_dpp((int)v, 0x104, 0xF, 0x5, false);  // (the other banks: the next line)
    return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)v, 0x114, 0xF, 0xA, false);
  } else if constexpr (M == 8) {  // row_shl:8 into banks 0 and 1, row_shr:8 into banks 2 and 3
    const int r = __builtin_amdgcn_mov_dpp((int)v, 0x108, 0xF, 0x3, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)v, 0x118, 0xF, 0xC, false);
  } else if constexpr (M == 16) {  // odd rows of the first copy <-> even rows of the second (gfx950)
    const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16u) ? p[0] : p[1];
  } else {  // upper half of the first copy <-> lower half of the second (gfx950)
    static_assert(M == 32, "lane distance");
    const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32u) ? p[0] : p[1];
  }
}
template <int M>
__device__ __forceinline__ void lane_step(unsigned long long (&x)[4], uint32_t flip_raw) {  // flip = 0: this lane keeps the smaller word
  const uint32_t flip = opaque(flip_raw), nflip = opaque(~flip_raw);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned long long y = ((unsigned long long)lane_xor32<M>((uint32_t)(x[e] >> 32)) << 32) | lane_xor32<M>((uint32_t)x[e]);
    x[e] = bfi64(swap_mask(y, x[e], flip, nflip), y, x[e]);  // (equal words — padding — : either way the same word stays)
  }
}

// The end of one merge phase of the network for a thread that holds the words i0 .. i0 + 3: distances J0, J0 / 2, .. 4 by
// lane exchange, 2 and 1 inside the thread.
template <int J0>
__device__ __forceinline__ void merge_tail(unsigned long long (&x)[4], uint32_t i0, uint32_t fu) {  // fu = 0: the cell sorts upwards, ~0: downwards
  // (the lower partner of an upward cell keeps the smaller word: flip = -(bit log2 J of i0) ^ fu, two VALU instructions)
  if constexpr (J0 >= 128) lane_step<32>(x, (uint32_t)(((int)(i0 << 24)) >> 31) ^ fu);
  if constexpr (J0 >= 64) lane_step<16>(x, (uint32_t)(((int)(i0 << 25)) >> 31) ^ fu);
  if constexpr (J0 >= 32) lane_step<8>(x, (uint32_t)(((int)(i0 << 26)) >> 31) ^ fu);
  if constexpr (J0 >= 16) lane_step<4>(x, (uint32_t)(((int)(i0 << 27)) >> 31) ^ fu);
  if constexpr (J0 >= 8) lane_step<2>(x, (uint32_t)(((int)(i0 << 28)) >> 31) ^ fu);
  lane_step<1>(x, (uint32_t)(((int)(i0 << 29)) >> 31) ^ fu);
  const uint32_t f = opaque(fu), nf = opaque(~fu);
  compare_exchange(x[0], x[2], f, nf), compare_exchange(x[1], x[3], f, nf);
  compare_exchange(x[0], x[1], f, nf), compare_exchange(x[2], x[3], f, nf);
}

// Bitonic network over SLOTS words in cells of `cap` (a power of two, 32 <= cap <= SLOTS), four CONSECUTIVE words per
// thread (word i = 4 * threadIdx.x + e) in registers on entry and exit.
//   REGS: partner distances 1 and 2 stay inside the thread, 4 .. 128 are lane exchanges inside the wave (no barrier),
//         256 and more go through LDS (two barriers each).
//   else: the words live in LDS; two network stages (distances j and j / 2) share one LDS round trip: a thread owns the
//         four words base + {0, j/2, j, 3j/2}.
template <uint32_t SLOTS, bool REGS>
__device__ __forceinline__ void bitonic_sort(Word (&x)[4], uint32_t cap, Word* lds) {
  constexpr uint32_t THREADS = SLOTS / 4;
  const uint32_t i0 = 4u * threadIdx.x;
#ifdef A3D_NW_SKIP  // timing experiment (wrong trees): what the kernel costs without its networks
  return;
#endif
  auto inside = [&](Word& lo, Word& hi, bool up) { compare_exchange(lo, hi, opaque(up ? 0u : ~0u), opaque(up ? ~0u : 0u)); };  // of two words one thread holds
  // Word i lives at phys(i) = i ^ 5 * (bits 4-5 of i) ^ (bit 6 of i) << 4, a GF(2)-linear shuffle inside every 32 words: a
  // stage pair of distances (j, j / 2) has quad q read the words base(q) | {0, j / 2, j, 3 j / 2}, i.e. the 32 lanes of a
  // ds_read_b64 group put q's low five bits at the address bits below log2(j / 2) and from log2(j) + 1 up — for j = 4 .. 64
  // two or more of them land on bits 5 and 6, outside the 32 8-byte slots of a bank row (the plain layout, and the one
  // padded by a word per 32 of rounds 5-6: two- to four-way conflicts on the middle distances).  With this shuffle the 32
  // lanes of every stage pair's reads and the 16 lanes of every ds_write_b64 group hit distinct slots
  // (scripts/lds_shuffle_search.py enumerates the linear maps that do).
  auto phys = [](uint32_t i) { return i ^ (((i >> 4) & 3u) * 5u) ^ ((i >> 2) & 16u); };
  if (REGS) {
    // The thread keeps its four consecutive words in registers through the whole network.  Distances 1 and 2 are inside
    // the thread; 4 .. 128 are the lanes l ^ 1 .. l ^ 32 of the same wave: the partner's word comes over the VALU's own
    // lane paths (DPP quad_perm / row shifts, v_permlane16_swap, v_permlane32_swap: lane_xor32) — no LDS traffic, no
    // waiting, both partners compare and each keeps its side.  Only distances >= 256 (other waves' words) go through LDS,
    // two distances per round trip.  The network with its words in LDS (below) was bound by the LDS store path (ds_write_b64:
    // ~85 B/clk per CU, MI355X_MICROARCH.md) and by one LDS round trip per stage pair: 24 400 cycles for the 2048-word sort.
    const uint32_t cell_i0 = i0 & (cap - 1u);
    inside(x[0], x[1], (cell_i0 & 2u) == 0u), inside(x[2], x[3], ((cell_i0 + 2u) & 2u) == 0u);
    {
      const bool up = (cell_i0 & 4u) == 0u;
      inside(x[0], x[2], up), inside(x[1], x[3], up), inside(x[0], x[1], up), inside(x[2], x[3], up);
    }
    const uint32_t p_own = phys(i0);  // (i0 = 4 t: the words i0 + e stand at p_own ^ e)
    A3D_PHASE_STAMP(40, 4);
    for (uint32_t kk = 8; kk <= cap; kk <<= 1) {
      const uint32_t fu = (cell_i0 & kk) ? ~0u : 0u;  // (kk >= 8: one direction for the thread's four words; 0 = upwards)
      uint32_t j = kk >> 1;
      if (j >= 256u) {
        __syncthreads();  // (the buffer's previous use may have been anyone's)
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) lds[p_own ^ e] = x[e];
        while (j >= 256u) {
          const uint32_t hh = j >> 1, lh = 31u - (uint32_t)__builtin_clz(hh);
          const uint32_t q = threadIdx.x;
          const uint32_t base = ((q >> lh) << (lh + 2)) | (q & (hh - 1u));
          const bool upq = ((base & (cap - 1u)) & kk) == 0u;
          const uint32_t p0 = phys(base), dh = phys(hh), dj = phys(j);
          __syncthreads();
          Word a0 = lds[p0], a1 = lds[p0 ^ dh], a2 = lds[p0 ^ dj], a3 = lds[p0 ^ dj ^ dh];
          inside(a0, a2, upq), inside(a1, a3, upq);
          inside(a0, a1, upq), inside(a2, a3, upq);
          lds[p0] = a0, lds[p0 ^ dh] = a1, lds[p0 ^ dj] = a2, lds[p0 ^ dj ^ dh] = a3;
          j >>= 2;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) x[e] = lds[p_own ^ e];
        A3D_PHASE_STAMP(41, kk);
      }
      // (block-uniform) j <= 128 by now: the partners of the remaining distances j .. 4 are the lanes ^ (j / 4) .. ^ 1, then the
      // thread's own words.  One straight-line tail per starting distance (a loop over j with a switch on the lane distance
      // inside made hipcc keep three copies of every word alive across the cases).
      switch (j) {
        case 128u: merge_tail<128>(x, i0, fu); break;
        case 64u: merge_tail<64>(x, i0, fu); break;
        case 32u: merge_tail<32>(x, i0, fu); break;
        case 16u: merge_tail<16>(x, i0, fu); break;
        case 8u: merge_tail<8>(x, i0, fu); break;
        default: merge_tail<4>(x, i0, fu); break;
      }
      A3D_PHASE_STAMP(42, kk);
    }
    __syncthreads();
    return;
  }
  // the first phases (kk = 2, 4: distances 1 | 2, 1) stay inside the thread, then everything through LDS (layout: phys).
  // Wave w's threads own the quads of words [256 w, 256 w + 256) in every stage pair of distances <= 128 (and the words
  // 4 t .. 4 t + 3 on entry, exit and in the distance-1 stage): such a stage reads only what its own wave wrote, the LDS
  // serves a wave's instructions in order, so between two of them no block barrier is needed — only the stage pairs of
  // distances >= 256 (5 of the 33 of a 2048-word sort, none below 512 words) exchange words between waves.  With two
  // waves per SIMD and a barrier per stage pair the network was bound by LDS round trips the barrier kept from overlapping.
  auto at = [&](uint32_t i) -> Word& { return lds[phys(i)]; };
  bool prev_cross = true;  // (the buffer's previous use may have been anyone's)
  auto stage_sync = [&](bool cross) {
    if (cross || prev_cross) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"), __builtin_amdgcn_wave_barrier();
    prev_cross = cross;
  };
  inside(x[0], x[1], ((i0 & (cap - 1u)) & 2u) == 0u), inside(x[2], x[3], (((i0 + 2u) & (cap - 1u)) & 2u) == 0u);
  {
    const bool up = ((i0 & (cap - 1u)) & 4u) == 0u;
    inside(x[0], x[2], up), inside(x[1], x[3], up), inside(x[0], x[1], up), inside(x[2], x[3], up);
  }
  stage_sync(false);
#pragma unroll
  for (int e = 0; e < 4; ++e) at(i0 + e) = x[e];
  for (uint32_t kk = 8; kk <= cap; kk <<= 1) {
    uint32_t j = kk >> 1;
    while (j >= 2) {
      const uint32_t hh = j >> 1, lh = 31u - (uint32_t)__builtin_clz(hh);
      const uint32_t q = threadIdx.x;  // SLOTS / 4 quads, one per thread
      const uint32_t base = ((q >> lh) << (lh + 2)) | (q & (hh - 1u));
      const bool up = ((base & (cap - 1u)) & kk) == 0u;
      stage_sync(j > 128u);
      // (phys is linear over xor and the bits of hh and j are clear in base: one shuffle per stage pair, the rest uniform)
      const uint32_t p0 = phys(base), dh = phys(hh), dj = phys(j);
      Word a0 = lds[p0], a1 = lds[p0 ^ dh], a2 = lds[p0 ^ dj], a3 = lds[p0 ^ dj ^ dh];
      inside(a0, a2, up), inside(a1, a3, up);
      inside(a0, a1, up), inside(a2, a3, up);
      lds[p0] = a0, lds[p0 ^ dh] = a1, lds[p0 ^ dj] = a2, lds[p0 ^ dj ^ dh] = a3;
      j >>= 2;
    }
    if (j == 1) {  // the thread's own four consecutive words
      const bool up = ((i0 & (cap - 1u)) & kk) == 0u;  // (kk >= 8: one direction for all four)
      stage_sync(false);
      const uint32_t p0 = phys(i0);  // (i0 = 4 t: the words i0 + e stand at p0 ^ e)
      Word a0 = lds[p0], a1 = lds[p0 ^ 1u], a2 = lds[p0 ^ 2u], a3 = lds[p0 ^ 3u];
      inside(a0, a1, up), inside(a2, a3, up);
      lds[p0] = a0, lds[p0 ^ 1u] = a1, lds[p0 ^ 2u] = a2, lds[p0 ^ 3u] = a3;
    }
  }
  stage_sync(false);
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] = at(i0 + e);
  __syncthreads();
}

// The same network on 128-bit words (L_d itself: key, previous axis' key, the one before, original index) with the
// point's position as payload, words and payload in LDS: the entry level's order when the range holds long runs of equal
// keys (a cloud from a depth image: thousands of points share a quantised z), where ranking every run by counting would
// cost its length squared.  Branch-free comparison: a chain of `?:` compiles to a ladder of divergent branches.
template <uint32_t SLOTS>
__device__ __forceinline__ void bitonic_lds128(uint4* w, uint32_t* pay, uint32_t cap) {
  constexpr uint32_t THREADS = SLOTS / 4;
  // (compare-exchange as in bitonic_sort: the direction as data, the 128-bit comparison as the borrow of hi - lo through four
  // VALU subtractions, one v_cndmask into the swap mask, the ten words moved by v_bfi — as `if (lt128(hi, lo) == up) swap` it
  // was three 64-bit compares joined on the scalar unit and ten VOP2 selects in a row: profiles/round6_valu_rate.txt)
  auto inside = [&](uint4& lo, uint4& hi, uint32_t& plo, uint32_t& phi, bool up) {
    const uint32_t flip = opaque(up ? 0u : ~0u), nflip = opaque(up ? ~0u : 0u);
    uint32_t m, t;
    asm volatile(
        "v_sub_co_u32 %1, vcc, %2, %6\n v_subb_co_u32 %1, vcc, %3, %7, vcc\n v_subb_co_u32 %1, vcc, %4, %8, vcc\n"
        "v_subb_co_u32 %1, vcc, %5, %9, vcc\n v_cndmask_b32 %0, %10, %11, vcc"
        : "=&v"(m), "=&v"(t)
        : "v"(hi.w), "v"(hi.z), "v"(hi.y), "v"(hi.x), "v"(lo.w), "v"(lo.z), "v"(lo.y), "v"(lo.x), "v"(flip), "v"(nflip)
        : "vcc");  // m = hi < lo (x the most significant word) ? nflip : flip
    auto bfi = [&](uint32_t take, uint32_t keep) { return (take & m) | (keep & ~m); };
    const uint4 nl = make_uint4(bfi(hi.x, lo.x), bfi(hi.y, lo.y), bfi(hi.z, lo.z), bfi(hi.w, lo.w));
    const uint4 nh = make_uint4(bfi(lo.x, hi.x), bfi(lo.y, hi.y), bfi(lo.z, hi.z), bfi(lo.w, hi.w));
    const uint32_t npl = bfi(phi, plo), nph = bfi(plo, phi);
    lo = nl, hi = nh, plo = npl, phi = nph;
  };
  // (block barriers only around the stage pairs that exchange words between waves: see bitonic_sort)
  bool prev_cross = true;
  auto stage_sync = [&](bool cross) {
    if (cross || prev_cross) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"), __builtin_amdgcn_wave_barrier();
    prev_cross = cross;
  };
  const uint32_t i0 = 4u * threadIdx.x;
  for (uint32_t kk = 2; kk <= cap; kk <<= 1) {
    uint32_t j = kk >> 1;
    while (j >= 2) {
      const uint32_t hh = j >> 1, lh = 31u - (uint32_t)__builtin_clz(hh);
      const uint32_t q = threadIdx.x;
      const uint32_t base = ((q >> lh) << (lh + 2)) | (q & (hh - 1u));
      const bool up = ((base & (cap - 1u)) & kk) == 0u;
      stage_sync(j > 128u);
      uint4 a0 = w[base], a1 = w[base + hh], a2 = w[base + j], a3 = w[base + j + hh];
      uint32_t p0 = pay[base], p1 = pay[base + hh], p2 = pay[base + j], p3 = pay[base + j + hh];
      inside(a0, a2, p0, p2, up), inside(a1, a3, p1, p3, up);
      inside(a0, a1, p0, p1, up), inside(a2, a3, p2, p3, up);
      w[base] = a0, w[base + hh] = a1, w[base + j] = a2, w[base + j + hh] = a3;
      pay[base] = p0, pay[base + hh] = p1, pay[base + j] = p2, pay[base + j + hh] = p3;
      j >>= 2;
    }
    if (j == 1) {  // the thread's own four consecutive words: two pairs, each with its own direction (kk = 2: they differ)
      stage_sync(false);
#pragma unroll
      for (uint32_t h2 = 0; h2 < 2; ++h2) {
        const uint32_t i = i0 + 2 * h2;
        uint4 a0 = w[i], a1 = w[i + 1];
        uint32_t p0 = pay[i], p1 = pay[i + 1];
        inside(a0, a1, p0, p1, ((i & (cap - 1u)) & kk) == 0u);
        w[i] = a0, w[i + 1] = a1, pay[i] = p0, pay[i + 1] = p1;
      }
    }
  }
  __syncthreads();
}

template <uint32_t SLOTS>
constexpr size_t nw_lds_bytes() { return SLOTS * (sizeof(float4) + sizeof(uint4) + 2 * sizeof(uint32_t)); }  // (>= the padded 64-bit words)

template <uint32_t SLOTS, bool REGS>
__global__ void __launch_bounds__(SLOTS / 4)
    sel_narrow_kernel(const float4* __restrict__ recs, uint32_t n, uint32_t d0, uint32_t D, float* __restrict__ split,
                      float4* __restrict__ leaves, uint32_t* __restrict__ slot_of_point, uint32_t* __restrict__ flags) {
  constexpr uint32_t THREADS = SLOTS / 4;  // four words per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* rec = (float4*)smem;           // [SLOTS] the range's points in their current arrangement
  Word* xw = (Word*)(rec + SLOTS);       // [SLOTS] the network's words (or its exchange buffer)
  // A coordinate of a point in LDS by ADDRESS (axis = dword offset in the record), never by a select on the axis:
  // hipcc (ROCm 7.2) turns `axis == 0 ? r.x : axis == 1 ? r.y : r.z` with a wave-uniform axis inside a divergent
  // while-loop into a ladder of scalar branches around a strength-reduced pointer, and the axis-0 arm of that pointer is
  // left one record behind when the loop exits (observed: runs of equal x keys ranked over [lo - 1, hi - 1)).
  __shared__ uint32_t long_runs, any_tie;
  const float* recf = (const float*)rec;
  auto coord = [&](uint32_t pos, uint32_t axis) { return recf[4u * pos + axis]; };
  auto lkey_at = [&](uint32_t pos, uint32_t level) {
    const uint32_t ax = level % 3;
    return LKey{ord_bits(coord(pos, ax)), level >= 1 ? ord_bits(coord(pos, (ax + 2) % 3)) : 0u,
                level >= 2 ? ord_bits(coord(pos, (ax + 1) % 3)) : 0u, __float_as_uint(coord(pos, 3))};
  };
  const uint32_t node0 = blockIdx.x;
#if defined(A3D_TAIL_STAMPS) && defined(A3D_NARROW_PHASE_STAMPS)
  if (threadIdx.x == 0 && blockIdx.x == 0) g_nw_phase_n = 0;
#endif
#if defined(A3D_TAIL_STAMPS) && defined(A3D_NARROW_NET_STAMPS) && !defined(A3D_NARROW_PHASE_STAMPS)  // scripts/narrow_stamps.py: block 0's levels (s_memtime)
  uint32_t n_stamp = 0;
#define A3D_NW_STAMP(tag, val)                                                                 \
  do {                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x == 0 && n_stamp < 32) {                                 \
      g_sel_stamps[2 * n_stamp] = __builtin_amdgcn_s_memtime();                                \
      g_sel_stamps[2 * n_stamp + 1] = ((unsigned long long)(tag) << 32) | (val);               \
      ++n_stamp;                                                                               \
    }                                                                                          \
  } while (0)
#else
#define A3D_NW_STAMP(tag, val) do { } while (0)
#endif
  A3D_NW_STAMP(30, 0);
  uint32_t s0, l0;
  bool exists0;
  sel_node_range(n, d0, node0, &s0, &l0, &exists0);
  uint32_t cap0 = 32;
  while (cap0 < l0) cap0 <<= 1;
  // pm[i]: which record stands at position i.  The entry level moves the records themselves (its run handling reads them by
  // position) and leaves pm the identity; every deeper level only permutes pm — 2 bytes per point and level instead of a
  // 16-byte record gathered and written back (round 6: those moves were 37 000 of the kernel's 198 000 cycles,
  // profiles/round6_kdtree_narrow_stamps.txt) — and the leaves are gathered through it once at the end.
  uint16_t* pm = (uint16_t*)(xw + (SLOTS + SLOTS / 32u + 64u));  // behind the network's padded words
  {  // (SLOTS / THREADS = 4 loads per thread, unconditional and all in flight: see sel_pack_kernel)
    float4 got[SLOTS / THREADS];
#pragma unroll
    for (uint32_t k = 0; k < SLOTS / THREADS; ++k) {
      const uint32_t p = threadIdx.x + k * THREADS;
      got[k] = recs[s0 + (p < l0 ? p : l0 - 1u)];
    }
    // (an empty asm that reads them: otherwise hipcc sinks every load into the `if` that stores its value, one round trip each)
#pragma unroll
    for (uint32_t k = 0; k < SLOTS / THREADS; ++k) asm volatile("" : "+v"(got[k].x), "+v"(got[k].y), "+v"(got[k].z), "+v"(got[k].w));
#pragma unroll
    for (uint32_t k = 0; k < SLOTS / THREADS; ++k) {
      const uint32_t p = threadIdx.x + k * THREADS;
      if (p < l0) rec[p] = got[k];
    }
  }
  for (uint32_t p = threadIdx.x; p < SLOTS; p += THREADS) pm[p] = (uint16_t)p;
  __syncthreads();
  A3D_NW_STAMP(31, l0);
  const uint32_t i0 = 4u * threadIdx.x;
  // The level a range enters at finds its points in arbitrary order and has to establish L_d itself: the network on
  // `key << 32 | current position` orders the keys, then every run of EQUAL keys (none, on ordinary data) is put into the
  // order of the rest of L_d by counting.  (Level d0 - 1 first if the range already is a leaf — possible with a lowered
  // wide / narrow border only: its points must stand in the order its parent's sort left them in.)  Every deeper level
  // finds L_{d-1} in the positions and needs the network only.
  const uint32_t first = (l0 <= 16 && d0 >= 1) ? d0 - 1 : d0;
  // the thread's cell of the level before, carried down (kdtree.rs:46-52: mid = len / 2; a range of <= 16 points is a leaf
  // and has no children): one step per level instead of sel_node_range's walk from the root (round 6)
  uint32_t c_rs = 0, c_lc = l0;
  bool c_ok = true;
  for (uint32_t d = first; d < D || d == first; ++d) {
    const bool entry = d <= d0;
    const uint32_t a = d % 3, rel = entry ? 0u : d - d0;
    const uint32_t cap = cap0 >> rel;
    if (cap < 32) break;  // everything below is a leaf
    const uint32_t cell = i0 / cap, pos0 = i0 & (cap - 1u);
    uint32_t rs = 0, lc = 0;
    bool sorted = false;
    if (cell < (1u << rel)) {
      if (entry) {
        rs = 0, lc = l0;
        sorted = d < d0 ? true : l0 > 16;  // (d < d0: the leaf's own order, no split below)
      } else {
        if (c_ok && c_lc > 16) {
          const uint32_t mid = c_lc >> 1;
          if (cell & 1u) c_rs += mid, c_lc -= mid;
          else c_lc = mid;
        } else {
          c_ok = false;
        }
        rs = c_rs, lc = c_lc;
        sorted = c_ok && lc > 16;
      }
    }
    constexpr uint32_t SHORT_RUN = 8;  // a run no longer than this on either side of a point is ranked by counting
    // Entry level: does the range hold long runs of equal keys?  Every key is counted into a 4096-entry hash table (the
    // network's buffer, free here); equal keys share an entry, so a run of more than SHORT_RUN keys always shows (a few
    // unequal keys that collide can show too: that only picks the slower of two exact paths).  A range with long runs
    // skips the network on 64-bit words — its result would be thrown away — and goes through the one on L_d itself at once.
    bool heavy = false;
    if (d == d0 && l0 > 16) {
      uint32_t* tbl = (uint32_t*)xw;
      if (threadIdx.x == 0) long_runs = 0u;
      for (uint32_t q = threadIdx.x; q < 4096u; q += THREADS) tbl[q] = 0u;
      __syncthreads();
      bool seen = false;
      for (uint32_t p = threadIdx.x; p < l0; p += THREADS)
        seen |= atomicAdd(&tbl[(ord_bits(coord(p, a)) * 0x9E3779B1u) >> 20], 1u) >= SHORT_RUN;
      if (seen) long_runs = 1u;  // (benign race: every writer stores 1)
      __syncthreads();
      heavy = long_runs != 0u;
      __syncthreads();
    }
    A3D_NW_STAMP(32, d);  // level set up (entry level: the long-run check done)
    bool mine[4];
    Word x[4];
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
      mine[e] = sorted && pos0 + e < lc;
      x[e] = WORD_PADDING;
      if (mine[e]) {
        // (at the entry level pm is the identity by definition — and its LDS words are the hash table's / the 128-bit network's
        // until the level's end: read the position itself)
        const float v = coord(entry ? rs + pos0 + e : (uint32_t)pm[rs + pos0 + e], a);
        if (d >= d0 && v != v) atomicOr(&flags[FLAG_NAN], 1u);  // partial_cmp().unwrap() would panic (kdtree.rs:43)
        x[e] = ((Word)ord_bits(v) << 32) | (pos0 + e);
      }
    }
    if (!heavy) {
      bitonic_sort<SLOTS, REGS>(x, cap, xw);
      A3D_NW_STAMP(33, cap);  // network done
      if (entry) {  // (block-uniform) the records themselves; pm stays the identity
        // (four named values, loaded unconditionally: as an array, or behind `if (mine[e])`, hipcc keeps them in scratch memory
        // across the barrier — 5 900 cycles of the kernel for this move alone)
        const float4 m0 = rec[mine[0] ? rs + (uint32_t)x[0] : 0u], m1 = rec[mine[1] ? rs + (uint32_t)x[1] : 0u],
                     m2 = rec[mine[2] ? rs + (uint32_t)x[2] : 0u], m3 = rec[mine[3] ? rs + (uint32_t)x[3] : 0u];
        __syncthreads();
        if (mine[0]) rec[rs + pos0] = m0;
        if (mine[1]) rec[rs + pos0 + 1] = m1;
        if (mine[2]) rec[rs + pos0 + 2] = m2;
        if (mine[3]) rec[rs + pos0 + 3] = m3;
      } else {
        uint16_t np[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) np[e] = mine[e] ? pm[rs + (uint32_t)x[e]] : (uint16_t)0;
        __syncthreads();
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e)
          if (mine[e]) pm[rs + pos0 + e] = np[e];
      }
      __syncthreads();
    }
    A3D_NW_STAMP(34, d);  // records moved
    if (entry) {  // runs of equal keys into the order of (previous axis' key, the one before, original index)
      uint32_t* dest = (uint32_t*)xw;    // (the network's buffer is free between sorts)
      if (threadIdx.x == 0) long_runs = heavy ? 1u : 0u, any_tie = heavy ? 1u : 0u;
      __syncthreads();
      // ordinary data has no two equal keys in a range: one look at both neighbours of the thread's four points (all loads in
      // flight at once) settles that, and nothing of the run handling below runs — not even `dest` is written (round 6: the loop
      // below, with its dependent LDS reads per point, was 11 600 of the kernel's cycles on a cloud without a single tie)
      uint32_t tied = 0u;  // bit e: the thread's e-th point has a neighbour with its key
      if (!heavy) {
        uint32_t kc[4], kl[4], kr[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
          const uint32_t p = threadIdx.x + e * THREADS, pc = p < l0 ? p : 0u;
          kc[e] = ord_bits(coord(pc, a)), kl[e] = ord_bits(coord(pc > 0 ? pc - 1 : pc, a)), kr[e] = ord_bits(coord(pc + 1 < l0 ? pc + 1 : pc, a));
        }
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
          const uint32_t p = threadIdx.x + e * THREADS;
          tied |= (p < l0 && ((p > 0 && kl[e] == kc[e]) || (p + 1 < l0 && kr[e] == kc[e])) ? 1u : 0u) << e;
        }
        if (tied) any_tie = 1u;  // (benign race: every writer stores 1)
      }
      __syncthreads();
      A3D_NW_STAMP(38, any_tie);  // neighbours compared
      const bool ties = any_tie != 0u;  // (block-uniform)
      if (!heavy && ties) {
#pragma unroll 1
        for (uint32_t e = 0; e < 4; ++e) {
          const uint32_t p = threadIdx.x + e * THREADS;
          if (p >= l0) continue;
          if (!((tied >> e) & 1u)) {  // (nearly every point: it stays where it is)
            dest[p] = p;
            continue;
          }
          const uint32_t k = ord_bits(coord(p, a));
          uint32_t lo = p, hi = p + 1;
          while (lo > 0 && p - lo < SHORT_RUN && ord_bits(coord(lo - 1, a)) == k) --lo;
          while (hi < l0 && hi - p <= SHORT_RUN && ord_bits(coord(hi, a)) == k) ++hi;
          if (p - lo >= SHORT_RUN || hi - p > SHORT_RUN) long_runs = 1u;  // (benign race: every writer stores 1)
          const LKey me = lkey_at(p, d);
          uint32_t rnk = 0;
          for (uint32_t q = lo; q < hi; ++q) rnk += lkey_less(lkey_at(q, d), me) ? 1u : 0u;
          dest[p] = lo + rnk;
        }
      }
      __syncthreads();
      A3D_NW_STAMP(39, long_runs);  // short runs ranked
      // (no two equal keys in the range — ordinary data —: every point already stands where L_d puts it, nothing to move)
      const bool reorder = ties;
      if (long_runs) {
        // long runs of equal keys (a cloud from a depth image: thousands of points share a quantised z): the whole range
        // once more through the network, on the 128-bit words of L_d, instead of run-length-squared counting
        uint4* w128 = (uint4*)xw;
        uint32_t* pay = (uint32_t*)(w128 + SLOTS);
        for (uint32_t q = threadIdx.x; q < SLOTS; q += THREADS) {
          uint4 word = make_uint4(~0u, ~0u, ~0u, ~0u);
          if (q < l0) {
            const LKey lk = lkey_at(q, d);
            word = make_uint4(lk.k0, lk.k1, lk.k2, lk.idx);
          }
          w128[q] = word, pay[q] = q;
        }
        __syncthreads();
        bitonic_lds128<SLOTS>(w128, pay, cap0);
        dest = (uint32_t*)xw + SLOTS * 5;  // behind the 128-bit words and their payload
        for (uint32_t q = threadIdx.x; q < l0; q += THREADS) dest[pay[q]] = q;  // the point that stood at pay[q] goes to q
        __syncthreads();
      }
      if (reorder) {  // (block-uniform)
        const uint32_t q0 = threadIdx.x, q1 = q0 + THREADS, q2 = q1 + THREADS, q3 = q2 + THREADS;
        const float4 v0 = rec[q0 < l0 ? q0 : 0u], v1 = rec[q1 < l0 ? q1 : 0u], v2 = rec[q2 < l0 ? q2 : 0u], v3 = rec[q3 < l0 ? q3 : 0u];
        __syncthreads();
        if (q0 < l0) rec[dest[q0]] = v0;
        if (q1 < l0) rec[dest[q1]] = v1;
        if (q2 < l0) rec[dest[q2]] = v2;
        if (q3 < l0) rec[dest[q3]] = v3;
        __syncthreads();
      }
      A3D_NW_STAMP(43, reorder);  // tied points in place
    }
    if (entry) {  // (the long-run path's 128-bit words lie over pm: the identity again before the first permuting level)
      for (uint32_t p = threadIdx.x; p < SLOTS; p += THREADS) pm[p] = (uint16_t)p;
      __syncthreads();
      A3D_NW_STAMP(35, d);  // entry level: runs of equal keys ordered
    }
    if (d >= d0 && sorted) {
      const uint32_t mid = lc >> 1;
      if (mid >= pos0 && mid < pos0 + 4) split[((1u << d) - 1u) + (node0 << rel) + cell] = coord(pm[rs + mid], a);  // points[mid][k] (kdtree.rs:47-49)
    }
  }
  A3D_NW_STAMP(36, 0);  // all levels done
  // leaves: slot r of the leaf reached by `path` at depth `depth` lives at (path << (D - depth)) * 16 + r (kdtree.hpp)
  for (uint32_t p = threadIdx.x; p < l0; p += THREADS) {
    uint32_t rs = 0, rl = l0, path = node0, depth = d0;
    while (depth < D && rl > 16) {
      const uint32_t mid = rl >> 1;
      if (p - rs < mid) rl = mid, path = 2 * path;
      else rs += mid, rl -= mid, path = 2 * path + 1;
      ++depth;
    }
    const uint32_t slot = (path << (D - depth)) * 16u + (p - rs);
    const float4 r = rec[pm[p]];
    leaves[slot] = r;
    slot_of_point[__float_as_uint(r.w)] = slot;
  }
  // +inf in the slots no point took, 0 in the split entries of this subtree's nodes that are leaves at depth D - 1
  // (how many slots of each of the subtree's 2^sub leaf positions are taken: once per leaf position into the network's
  // buffer, free by now, instead of a walk from the root for each of its sixteen slots)
  const uint32_t sub = D - d0, first_leaf = node0 << sub;
  uint32_t* taken = (uint32_t*)xw;
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < (1u << sub); q += THREADS) {
    const uint32_t P = first_leaf + q;
    uint32_t cnt;
    if (D == 0) {
      cnt = n < 16u ? n : 16u;
    } else {
      uint32_t ps, pl;
      bool pe;
      sel_node_range(n, D - 1, P >> 1, &ps, &pl, &pe);  // (every node of depth D - 1 exists: leaves sit at D - 1 or D)
      if (pl <= 16) cnt = (P & 1u) == 0u ? pl : 0u;       // a leaf at depth D - 1 fills the even child's slots
      else cnt = (P & 1u) ? pl - (pl >> 1) : (pl >> 1);
    }
    taken[q] = cnt;
  }
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < (16u << sub); q += THREADS)
    if ((q & 15u) >= taken[q >> 4])
      leaves[(size_t)first_leaf * 16u + q] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f);
  A3D_NW_STAMP(37, 0);  // leaves, slot_of_point and padding written
  if (D >= 1 && D - 1 >= d0) {
    const uint32_t subm = D - 1 - d0;
    for (uint32_t q = threadIdx.x; q < (1u << subm); q += THREADS) {
      const uint32_t pth = (node0 << subm) + q;
      uint32_t ps, pl;
      bool pe;
      sel_node_range(n, D - 1, pth, &ps, &pl, &pe);
      if (!(pe && pl > 16)) split[((1u << (D - 1)) - 1u) + pth] = 0.0f;
    }
  }
}

#ifdef A3D_DIAGNOSTICS
// ---- ranges of <= NARROW points by SELECTION inside the block: diagnostics build only (A3D_KDTREE_SORTNET=select) ------
// Measured SLOWER than the network kernel above (145 against 91 us at 500 k points; 0.48 against 0.36 ms for Icp::new on
// the sample1 cloud): a narrowing round is ten block-wide phases of an LDS round trip + barrier each (4.4 us of a round
// are fixed, 4.5 us the points' work), a level takes one to two rounds, and the cells of <= 33 points count 33 keys per
// point.  Kept as a third, independent construction of the in-block levels for the identity tests.
// A level needs what the wide levels need: per cell the point of rank len / 2 under L_d and a partition around it, in
// any order — only the leaves have an order, L of their parent's depth, and they hold <= 16 points.  Per level, for all
// cells of the range at once:
//   rounds  every cell narrows its candidates (at first all its points): [min, max] of one component of L_d over them
//           (a component all candidates agree in passes the turn to the next one), up to `cap` buckets over it by a
//           shift of the key bits, histogram, block-wide scan, the bucket of the sought rank; candidates below / above
//           it take places at the cell's left / right end, the bucket's own stay candidates — until no cell has more
//           than NS_RANK of them (one round on ordinary data; an outlier or a run of equal keys costs its cell more);
//   rank    the last candidates of a cell are ranked among themselves under L_d by counting and fill the gap in that
//           order: the one that lands on len / 2 is the median (split value, children's places follow).
// A cell of <= 33 points, which has a leaf among its children, is sorted outright (every point counts the points before it, key
// bits first, the rest of L_d on equal ones): the leaves then stand as the reference's sort of their parent leaves them.
constexpr uint32_t NS_CELLS = 256;  // most cells of a level (2^(levels inside the block))
constexpr uint32_t NS_RANK = 8;     // candidates a selecting cell ranks by counting
constexpr uint32_t NS_SORTS = 33;   // a cell of up to this many points has a leaf among its children (33 -> 16 + 17): it is sorted

// counters[idx]++ for every valid lane, the old value returned; the lanes that share the first valid lane's idx add as one
__device__ __forceinline__ uint32_t take_wave(uint32_t* counters, uint32_t idx, bool valid) {
  const unsigned long long act = __builtin_amdgcn_ballot_w64(valid);
  if (!act) return 0u;
  const uint32_t lane = lane_id(), lead = (uint32_t)__builtin_ctzll(act);
  const uint32_t il = (uint32_t)__builtin_amdgcn_readlane((int)idx, (int)lead);
  const bool with_lead = valid && idx == il;
  const unsigned long long same = __builtin_amdgcn_ballot_w64(with_lead);
  uint32_t base = 0u;
  if (lane == lead) base = atomicAdd(&counters[il], (uint32_t)__builtin_popcountll(same));
  base = (uint32_t)__shfl((int)base, (int)lead, 64);
  if (with_lead) return base + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
  return valid ? atomicAdd(&counters[idx], 1u) : 0u;
}
// mn[idx] = min(mn[idx], k), mx[idx] = max(...) for every valid lane; the first valid lane's group reduces in the wave first
__device__ __forceinline__ void minmax_wave(uint32_t* mn, uint32_t* mx, uint32_t idx, uint32_t k, bool valid) {
  const unsigned long long act = __builtin_amdgcn_ballot_w64(valid);
  if (!act) return;
  const uint32_t lane = lane_id(), lead = (uint32_t)__builtin_ctzll(act);
  const uint32_t il = (uint32_t)__builtin_amdgcn_readlane((int)idx, (int)lead);
  const bool with_lead = valid && idx == il;
  uint32_t lo = with_lead ? k : ~0u, hi = with_lead ? k : 0u;
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
  }
  if (lane == lead) atomicMin(&mn[il], lo), atomicMax(&mx[il], hi);
  else if (valid && !with_lead) atomicMin(&mn[idx], k), atomicMax(&mx[idx], k);
}

template <uint32_t SLOTS>
constexpr size_t ns_lds_bytes() { return SLOTS * (2 * sizeof(float4) + 3 * sizeof(uint32_t)); }

template <uint32_t SLOTS>
__global__ void __launch_bounds__(SLOTS / 4)
    sel_narrow_select_kernel(const float4* __restrict__ recs, uint32_t n, uint32_t d0, uint32_t D, float* __restrict__ split,
                             float4* __restrict__ leaves, uint32_t* __restrict__ slot_of_point, uint32_t* __restrict__ flags) {
  constexpr uint32_t THREADS = SLOTS / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* rec = (float4*)smem;               // [SLOTS] the range's points in their current arrangement
  float4* rec_next = rec + SLOTS;            // [SLOTS] ... and in the next one
  uint32_t* hist = (uint32_t*)(rec_next + SLOTS);  // [SLOTS] bins: cell c owns [c * cap, (c + 1) * cap)
  uint32_t* pre = hist + SLOTS;              // [SLOTS] exclusive prefix of the bins
  uint32_t* cand = pre + SLOTS;              // [SLOTS] positions of a cell's last candidates, in its bins' range
  // per cell of the current level
  __shared__ uint32_t c_ncand[NS_CELLS], c_rank[NS_CELLS], c_left[NS_CELLS], c_right[NS_CELLS];  // candidates, the rank sought among them, points placed at either end
  __shared__ uint32_t c_cmp[NS_CELLS], c_mn[NS_CELLS], c_mx[NS_CELLS], c_shift[NS_CELLS];         // this round: component of L_d, its [min, max], the bucket shift (~0: no round)
  __shared__ uint32_t c_star[NS_CELLS], c_below[NS_CELLS], c_count[NS_CELLS];                     // this round's bucket of the sought rank
  __shared__ uint32_t cur_l[NS_CELLS], cur_r[NS_CELLS], cur_c[NS_CELLS];
  __shared__ uint32_t tmp[16];
  __shared__ uint32_t more;
  // (a coordinate by ADDRESS, never by a select on the axis: see sel_narrow_kernel)
  auto coord = [](const float4* buf, uint32_t pos, uint32_t axis) { return ((const float*)buf)[4u * pos + axis]; };
  auto lkey_at = [&](const float4* buf, uint32_t pos, uint32_t level) {
    const uint32_t ax = level % 3;
    return LKey{ord_bits(coord(buf, pos, ax)), level >= 1 ? ord_bits(coord(buf, pos, (ax + 2) % 3)) : 0u,
                level >= 2 ? ord_bits(coord(buf, pos, (ax + 1) % 3)) : 0u, __float_as_uint(coord(buf, pos, 3))};
  };
  auto lkey_comp_at = [&](const float4* buf, uint32_t pos, uint32_t level, uint32_t cmp) {
    const uint32_t ax = level % 3;
    if (cmp == 0) return ord_bits(coord(buf, pos, ax));
    if (cmp == 1) return level >= 1 ? ord_bits(coord(buf, pos, (ax + 2) % 3)) : 0u;
    if (cmp == 2) return level >= 2 ? ord_bits(coord(buf, pos, (ax + 1) % 3)) : 0u;
    return __float_as_uint(coord(buf, pos, 3));
  };
  const uint32_t node0 = blockIdx.x;
#ifdef A3D_TAIL_STAMPS
  uint32_t n_stamp = 0;
#define A3D_NS_STAMP(tag, val)                                                                      \
  do {                                                                                              \
    if (threadIdx.x == 0 && blockIdx.x == 0 && n_stamp < 32) { \
      g_sel_stamps[2 * n_stamp] = __builtin_amdgcn_s_memrealtime();                                 \
      g_sel_stamps[2 * n_stamp + 1] = ((unsigned long long)(tag) << 32) | (val);                    \
      ++n_stamp;                                                                                    \
    }                                                                                               \
  } while (0)
#else
#define A3D_NS_STAMP(tag, val) do { } while (0)
#endif
  uint32_t s0, l0;
  bool exists0;
  sel_node_range(n, d0, node0, &s0, &l0, &exists0);
  uint32_t cap0 = 32;
  while (cap0 < l0) cap0 <<= 1;
  A3D_NS_STAMP(20, l0);
  for (uint32_t p = threadIdx.x; p < l0; p += THREADS) rec[p] = recs[s0 + p];
  __syncthreads();
  A3D_NS_STAMP(21, l0);
  // the cell of position p at level d0 + rel: id (its path below the range), start, length; false: no cell that splits
  // (a leaf, there or above: its id is shifted up so that it stays unique)
  auto locate = [&](uint32_t p, uint32_t rel, uint32_t* c, uint32_t* s, uint32_t* l) {
    uint32_t cc = 0, ss = 0, ll = l0;
    for (uint32_t t = 0; t < rel; ++t) {
      if (ll <= 16) {
        *c = cc << (rel - t), *s = ss, *l = ll;
        return false;
      }
      const uint32_t mid = ll >> 1;
      if (p - ss < mid) ll = mid, cc = 2 * cc;
      else ss += mid, ll -= mid, cc = 2 * cc + 1;
    }
    *c = cc, *s = ss, *l = ll;
    return ll > 16;
  };
  const uint32_t levels = D > d0 ? D - d0 : 0u;  // levels d0 .. D - 1 can split
  for (uint32_t rel = 0; rel < levels; ++rel) {
    const uint32_t d = d0 + rel, a = d % 3, cap = cap0 >> rel;
    if (cap < 32) break;  // everything below is a leaf
    const uint32_t cap_log = 31u - (uint32_t)__builtin_clz(cap), cells = 1u << rel;
    // ---- this level's cells.  A cell of <= 33 points (a child of <= 16 is a leaf) is SORTED under L_d — every point counts
    // the cell's points before it — so that the leaves stand in the order the reference's sort leaves them in; a larger
    // one only finds its median and partitions (rounds below).
    if (threadIdx.x == 0) more = 0u;
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < cells; c += THREADS) {
      uint32_t s = 0, l = l0;
      bool ok = true;
      for (uint32_t t = 0; t < rel; ++t) {
        if (l <= 16) {
          ok = false;
          break;
        }
        const uint32_t mid = l >> 1;
        if ((c >> (rel - 1 - t)) & 1u) s += mid, l -= mid;
        else l = mid;
      }
      const bool selects = ok && l > NS_SORTS;
      c_ncand[c] = selects ? l : 0u, c_rank[c] = l >> 1, c_left[c] = 0u, c_right[c] = 0u, c_cmp[c] = 0u;
      if (selects) more = 1u;  // (benign race: every writer stores 1)
    }
    // every point: its cell, its place once it has one (a candidate of a selecting cell: none yet)
    uint32_t e_cell[4], e_start[4], e_len[4], e_dest[4];
    bool e_cand[4], e_sorts[4];
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
      const uint32_t p = threadIdx.x + e * THREADS;
      e_cell[e] = 0u, e_start[e] = 0u, e_len[e] = 0u, e_dest[e] = p, e_cand[e] = false, e_sorts[e] = false;
      if (p < l0) {
        const bool splits = locate(p, rel, &e_cell[e], &e_start[e], &e_len[e]);
        e_cand[e] = splits && e_len[e] > NS_SORTS, e_sorts[e] = splits && e_len[e] <= NS_SORTS;
        const float v = coord(rec, p, a);
        if (splits && v != v) atomicOr(&flags[FLAG_NAN], 1u);  // partial_cmp().unwrap() would panic (kdtree.rs:43)
        cand[p] = ord_bits(v);  // (the sorting cells compare these first)
      }
    }
    __syncthreads();
    // ---- sorting cells
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
      if (!e_sorts[e]) continue;
      const uint32_t p = threadIdx.x + e * THREADS, k = cand[p];
      uint32_t r = 0;
      for (uint32_t q = e_start[e]; q < e_start[e] + e_len[e]; ++q) {
        const uint32_t kq = cand[q];
        bool before = kq < k;
        if (kq == k && q != p) before = lkey_less(lkey_at(rec, q, d), lkey_at(rec, p, d));  // (rare: equal keys)
        r += before ? 1u : 0u;
      }
      e_dest[e] = e_start[e] + r;
      if (r == (e_len[e] >> 1))  // the point of rank len / 2: `points[mid][k]` (kdtree.rs:47-49)
        split[((1u << d) - 1u) + (node0 << rel) + e_cell[e]] = coord(rec, p, a);
    }
    A3D_NS_STAMP(22, rel);
    // ---- selecting cells: narrowing rounds
    while (more) {  // (block-uniform: read after a barrier, written before the next one)
      for (uint32_t c = threadIdx.x; c < cells; c += THREADS) {
        c_mn[c] = ~0u, c_mx[c] = 0u, cur_l[c] = 0u, cur_r[c] = 0u;
        c_shift[c] = ~0u;
      }
      for (uint32_t q = threadIdx.x; q < cap0; q += THREADS) hist[q] = 0u;
      __syncthreads();
      if (threadIdx.x == 0) more = 0u;
      uint32_t e_key[4];
#pragma unroll
      for (uint32_t e = 0; e < 4; ++e) {
        const uint32_t p = threadIdx.x + e * THREADS, c = e_cell[e];
        const bool in = e_cand[e] && c_ncand[c] > NS_RANK;
        e_key[e] = in ? lkey_comp_at(rec, p, d, c_cmp[c]) : 0u;
        minmax_wave(c_mn, c_mx, c, e_key[e], in);
      }
      __syncthreads();
      for (uint32_t c = threadIdx.x; c < cells; c += THREADS) {
        if (c_ncand[c] > NS_RANK) {
          if (c_mn[c] == c_mx[c]) {
            c_cmp[c] += 1u;  // all candidates agree in this component: the next one decides (the index, unique, ends it)
            more = 1u;
          } else {
            const uint32_t span = c_mx[c] - c_mn[c], bits = 32u - (uint32_t)__builtin_clz(span);
            c_shift[c] = bits > cap_log ? bits - cap_log : 0u;  // (mn and mx land in different buckets, all below `cap`)
          }
        }
      }
      __syncthreads();
      uint32_t e_bin[4];
#pragma unroll
      for (uint32_t e = 0; e < 4; ++e) {
        const uint32_t c = e_cell[e];
        const bool in = e_cand[e] && c_shift[c] != ~0u;
        e_bin[e] = in ? (e_key[e] - c_mn[c]) >> c_shift[c] : 0u;
        hist_add_wave(hist, (c << cap_log) + e_bin[e], in);
      }
      __syncthreads();
      {  // exclusive prefix of the bins (four consecutive ones per thread)
        const uint32_t b0 = 4u * threadIdx.x;
        uint32_t v[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) v[e] = b0 + e < cap0 ? hist[b0 + e] : 0u;
        const uint32_t ex = block_exclusive_scan<THREADS>(v[0] + v[1] + v[2] + v[3], tmp);
        uint32_t run = ex;
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
          if (b0 + e < cap0) pre[b0 + e] = run;
          run += v[e];
        }
      }
      __syncthreads();
      {  // the bucket that holds the sought rank, per cell
        const uint32_t b0 = 4u * threadIdx.x;
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
          const uint32_t bin = b0 + e;
          if (bin >= cap0) continue;
          const uint32_t cnt = hist[bin], c = bin >> cap_log;
          if (!cnt || c >= cells || c_shift[c] == ~0u) continue;
          const uint32_t before = pre[bin] - pre[c << cap_log], t = c_rank[c];
          if (before <= t && t < before + cnt) c_star[c] = bin - (c << cap_log), c_below[c] = before, c_count[c] = cnt;
        }
      }
      __syncthreads();
#pragma unroll
      for (uint32_t e = 0; e < 4; ++e) {
        const uint32_t c = e_cell[e];
        const bool in = e_cand[e] && c_shift[c] != ~0u;
        const uint32_t star = in ? c_star[c] : 0u;
        const bool left = in && e_bin[e] < star, right = in && e_bin[e] > star;
        const uint32_t tl = take_wave(cur_l, c, left), tr = take_wave(cur_r, c, right);
        if (left) e_dest[e] = e_start[e] + c_left[c] + tl, e_cand[e] = false;
        if (right) e_dest[e] = e_start[e] + e_len[e] - 1u - c_right[c] - tr, e_cand[e] = false;
      }
      __syncthreads();
      for (uint32_t c = threadIdx.x; c < cells; c += THREADS) {
        if (c_shift[c] != ~0u) {
          c_left[c] += c_below[c], c_right[c] += c_ncand[c] - c_below[c] - c_count[c];
          c_rank[c] -= c_below[c], c_ncand[c] = c_count[c];
          if (c_count[c] > NS_RANK) more = 1u;
        }
      }
      __syncthreads();
      A3D_NS_STAMP(23, rel);
    }
    // ---- the last candidates of every selecting cell: ranked among themselves under L_d, they fill the gap in that order
    for (uint32_t c = threadIdx.x; c < cells; c += THREADS) cur_c[c] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
      const uint32_t c = e_cell[e];
      const uint32_t slot = take_wave(cur_c, c, e_cand[e]);
      if (e_cand[e]) hist[(c << cap_log) + slot] = threadIdx.x + e * THREADS;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
      if (!e_cand[e]) continue;
      const uint32_t p = threadIdx.x + e * THREADS, c = e_cell[e], nc = c_ncand[c];
      const LKey me = lkey_at(rec, p, d);
      uint32_t r = 0;
      for (uint32_t j = 0; j < nc; ++j) r += lkey_less(lkey_at(rec, hist[(c << cap_log) + j], d), me) ? 1u : 0u;
      e_dest[e] = e_start[e] + c_left[c] + r;
      if (c_left[c] + r == (e_len[e] >> 1))
        split[((1u << d) - 1u) + (node0 << rel) + c] = coord(rec, p, a);
    }
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
      const uint32_t p = threadIdx.x + e * THREADS;
      if (p < l0) rec_next[e_dest[e]] = rec[p];
    }
    __syncthreads();
    float4* sw = rec;
    rec = rec_next, rec_next = sw;
    A3D_NS_STAMP(24, rel);
  }
  // ---- a range that is a leaf itself (a lowered wide / narrow border only): the order its parent's sort left it in, L of
  // the parent's depth; the whole tree as one leaf (depth 0) keeps the order it came in
  if (l0 <= 16 && d0 >= 1) {
    const uint32_t p = threadIdx.x;
    float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t dest = p;
    if (p < l0) {
      mine = rec[p];
      const LKey me = lkey_at(rec, p, d0 - 1);
      uint32_t r = 0;
      for (uint32_t q = 0; q < l0; ++q) r += lkey_less(lkey_at(rec, q, d0 - 1), me) ? 1u : 0u;
      dest = r;
    }
    __syncthreads();
    if (p < l0) rec[dest] = mine;
    __syncthreads();
  }
  // leaves: slot r of the leaf reached by `path` at depth `depth` lives at (path << (D - depth)) * 16 + r (kdtree.hpp)
  for (uint32_t p = threadIdx.x; p < l0; p += THREADS) {
    uint32_t rs = 0, rl = l0, path = node0, depth = d0;
    while (depth < D && rl > 16) {
      const uint32_t mid = rl >> 1;
      if (p - rs < mid) rl = mid, path = 2 * path;
      else rs += mid, rl -= mid, path = 2 * path + 1;
      ++depth;
    }
    const uint32_t slot = (path << (D - depth)) * 16u + (p - rs);
    const float4 r = rec[p];
    leaves[slot] = r;
    slot_of_point[__float_as_uint(r.w)] = slot;
  }
  // +inf in the slots no point took, 0 in the split entries of this subtree's nodes that are leaves at depth D - 1
  const uint32_t sub = D - d0, first_leaf = node0 << sub;
  for (uint32_t q = threadIdx.x; q < (16u << sub); q += THREADS) {
    const uint32_t P = first_leaf + (q >> 4), r = q & 15u;
    bool used;
    if (D == 0) {
      used = r < n;
    } else {
      uint32_t ps, pl;
      bool pe;
      sel_node_range(n, D - 1, P >> 1, &ps, &pl, &pe);  // (every node of depth D - 1 exists: leaves sit at D - 1 or D)
      if (pl <= 16) used = (P & 1u) == 0u && r < pl;      // a leaf at depth D - 1 fills the even child's slots
      else used = r < ((P & 1u) ? pl - (pl >> 1) : (pl >> 1));
    }
    if (!used) leaves[(size_t)first_leaf * 16u + q] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f);
  }
  if (D >= 1 && D - 1 >= d0) {
    const uint32_t subm = D - 1 - d0;
    for (uint32_t q = threadIdx.x; q < (1u << subm); q += THREADS) {
      const uint32_t pth = (node0 << subm) + q;
      uint32_t ps, pl;
      bool pe;
      sel_node_range(n, D - 1, pth, &ps, &pl, &pe);
      if (!(pe && pl > 16)) split[((1u << (D - 1)) - 1u) + pth] = 0.0f;
    }
  }
}
#endif  // A3D_DIAGNOSTICS

struct SelLayout {
  uint32_t wide_levels = 0;            // levels 0 .. wide_levels - 1 run the split / resolve kernels
  uint32_t narrow_len = NARROW;
  uint32_t nb[32] = {};                // buckets per node on each wide level
  size_t recs_a = 0, recs_b = 0, mid = 0, hist = 0, hist_words = 0, boxes = 0, plans = 0, cursors = 0, partials = 0,
         flags = 0, wide = 0, whist = 0, zero_begin = 0, zero_bytes = 0, total = 0;
  uint32_t pack_blocks = 0;
  uint32_t wide_cap = MIDDLE_CAP;      // a median bucket of more points than this is an oversized one (SelWide)
  uint32_t place_levels = 0;           // levels 0 .. place_levels - 1 can hold one
};

uint32_t max_len_at(uint32_t n, uint32_t level) { return (uint32_t)(((uint64_t)n + (1ull << level) - 1) >> level); }

uint32_t buckets_setting() {
  uint32_t v = NB_DEFAULT;
  if (const char* env = A3D_DIAG_ENV("A3D_KDTREE_BUCKETS"))  // diagnostics build: tuning / tests
    if (*env) v = std::min(NB_MAX, std::max(64u, (uint32_t)atoi(env)));
  uint32_t p = 64;
  while (p < v) p <<= 1;
  return p;
}

SelLayout sel_layout(uint32_t n, uint32_t narrow_len, uint32_t wide_cap) {
  SelLayout L;
  L.narrow_len = narrow_len;
  L.wide_cap = wide_cap;
  while (max_len_at(n, L.wide_levels) > narrow_len) ++L.wide_levels;
  while (L.place_levels < L.wide_levels && max_len_at(n, L.place_levels) > wide_cap) ++L.place_levels;
  size_t hist_words = 1;
  const uint32_t nb_cap = buckets_setting();
  for (uint32_t d = 0; d < L.wide_levels; ++d) {
    uint32_t nb = 64;
    while (nb < nb_cap && nb * 32u < max_len_at(n, d)) nb <<= 1;
    L.nb[d] = nb;
    hist_words = std::max(hist_words, ((size_t)1 << d) * nb);
  }
  auto pad = [](size_t b) { return ((b + 255) / 256) * 256; };
  const size_t nodes = (size_t)1 << L.wide_levels;  // heap-indexed tables over all wide levels: 2^levels - 1 entries
  L.pack_blocks = (n + PACK_TILE - 1) / PACK_TILE;
  size_t off = 0;
  L.recs_a = off, off += pad((size_t)n * 16);
  L.recs_b = off, off += pad((size_t)n * 16);
  L.mid = off, off += pad((size_t)n * 16);
  L.boxes = off, off += pad(nodes * sizeof(SelBox));
  L.plans = off, off += pad(nodes * sizeof(SelPlan));
  L.partials = off, off += pad((size_t)L.pack_blocks * 6 * sizeof(float));
  L.zero_begin = off;
  L.flags = off, off += 256;
  L.cursors = off, off += pad(nodes * 4 * CURSOR_STRIDE * sizeof(uint32_t));
  L.hist_words = hist_words;
  L.hist = off, off += pad(2 * hist_words * sizeof(uint32_t));
  L.wide = off, off += pad(((size_t)1 << L.place_levels) * sizeof(SelWide));  // heap-indexed over the levels that can hold one
  L.whist = off, off += pad(((size_t)1 << (L.place_levels ? L.place_levels - 1 : 0)) * 2 * NSUB * sizeof(uint32_t));  // per node of a level
  L.zero_bytes = off - L.zero_begin;
  L.total = off;
  return L;
}

uint32_t wide_cap_setting() {
  uint32_t v = MIDDLE_CAP;
  if (const char* env = A3D_DIAG_ENV("A3D_KDTREE_WIDE_CAP"))  // diagnostics build: oversized median buckets at test sizes
    if (*env) v = std::min(MIDDLE_CAP, std::max(16u, (uint32_t)atoi(env)));
  return v;
}

uint32_t narrow_len_setting() {
  uint32_t v = NARROW;
  if (const char* env = A3D_DIAG_ENV("A3D_KDTREE_NARROW_LEN"))  // diagnostics build: wide levels at test sizes
    if (*env) v = std::min(NARROW, std::max(32u, (uint32_t)atoi(env)));
  return v;
}

}  // namespace

namespace a3d {

size_t kdtree_select_scratch_bytes(uint32_t n) { return sel_layout(n, narrow_len_setting(), wide_cap_setting()).total; }

// d_points: [n][3] f32 on the device; `scratch`: kdtree_select_scratch_bytes(n) bytes.  Fills t->d_split, t->d_leaves,
// t->d_slot_of_point (allocated by the caller) and synchronises.
a3d_status kdtree_build_device_select(a3d_kdtree* t, const float* d_points, void* scratch, hipEvent_t done) {
  hipStream_t s = t->ctx->stream;
  const uint32_t n = t->n, D = t->max_depth;
  const SelLayout L = sel_layout(n, narrow_len_setting(), wide_cap_setting());
  // sel_place_kernel's launches (one more, and the resolve step a launch of its own, per level that can hold an oversized
  // median bucket): a context starts with them (round 6: its first depth-image cloud paid a lone resolve block's streaming
  // rounds, 0.70 instead of 0.34 ms), drops them after KD_QUIET_BUILDS builds in a row that met no such bucket (clouds without
  // thousands of equal coordinates pay ~35 us more on a context's first builds only) and takes them up again when one shows.
  // The tree is the same either way.
  constexpr int KD_QUIET_BUILDS = 4;
  const bool place_by_context = t->ctx->kd_wide_place.load();
  bool place = place_by_context;
  if (const char* env = A3D_DIAG_ENV("A3D_KDTREE_WIDE_PLACE"))
    if (*env) place = atoi(env) != 0;
  const uint32_t wide_cap = place ? L.wide_cap : 0xffffffffu;
  t->built_by = place ? 3 : 1;
  SelWide* wide = (SelWide*)((char*)scratch + L.wide);
  uint32_t* whist = (uint32_t*)((char*)scratch + L.whist);
  char* base = (char*)scratch;
  float4* recs[2] = {(float4*)(base + L.recs_a), (float4*)(base + L.recs_b)};
  float4* mid = (float4*)(base + L.mid);
  SelBox* boxes = (SelBox*)(base + L.boxes);
  SelPlan* plans = (SelPlan*)(base + L.plans);
  float* partials = (float*)(base + L.partials);
  uint32_t* flags = (uint32_t*)(base + L.flags);
  uint32_t* cursors = (uint32_t*)(base + L.cursors);
  uint32_t* hist[2] = {(uint32_t*)(base + L.hist), (uint32_t*)(base + L.hist) + L.hist_words};
  // more than 64 KiB of dynamic LDS has to be requested once per kernel AND device (a process may drive several: multi.hip)
  static_assert((2 * NB_MAX + 2 * NSUB) * sizeof(uint32_t) <= 64 * 1024, "the split kernel's tables fit the default dynamic LDS limit");
  static std::atomic<bool> lds_allowed_on[64];
  std::atomic<bool>& lds_allowed = lds_allowed_on[t->ctx->device & 63];
  if (!lds_allowed.load()) {
    A3D_HIP_TRY(hipFuncSetAttribute((const void*)sel_resolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K2_LDS_BYTES));
    A3D_HIP_TRY(hipFuncSetAttribute((const void*)sel_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K2_LDS_BYTES));
    A3D_HIP_TRY(hipFuncSetAttribute((const void*)sel_narrow_kernel<2048, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nw_lds_bytes<2048>()));
    A3D_HIP_TRY(hipFuncSetAttribute((const void*)sel_narrow_kernel<2048, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nw_lds_bytes<2048>()));
#ifdef A3D_DIAGNOSTICS
    A3D_HIP_TRY(hipFuncSetAttribute((const void*)sel_narrow_select_kernel<2048>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ns_lds_bytes<2048>()));
#endif
    lds_allowed.store(true);
  }
  // (flags, place counters, both histogram tables, the oversized buckets' tables: zeroed by the pack kernel)
  static_assert(sizeof(uint4) == 16, "zero region in 16-byte words");
  hipLaunchKernelGGL(sel_pack_kernel, dim3(L.pack_blocks), dim3(PACK_THREADS), 0, s, d_points, n, recs[0], partials,
                     (uint4*)(base + L.zero_begin), (uint32_t)(L.zero_bytes / 16));
  const uint32_t W = L.wide_levels;
  bool fuse = true;  // diagnostics build: A3D_KDTREE_FUSE=0 keeps the resolve step a launch of its own (the cross-check)
  if (const char* env = A3D_DIAG_ENV("A3D_KDTREE_FUSE")) fuse = atoi(env) != 0;
  if (W > 0) {
    uint32_t* ticket0 = flags + 8;  // (a word of the flag table's zeroed line)
    if (fuse) {
      hipLaunchKernelGGL(sel_hist0_kernel<true>, dim3((n + K1_TILE - 1) / K1_TILE), dim3(K1_THREADS), L.nb[0] * sizeof(uint32_t), s,
                         recs[0], n, partials, L.pack_blocks, L.nb[0], boxes, hist[0], ticket0, plans);
    } else {
      hipLaunchKernelGGL(sel_hist0_kernel<false>, dim3((n + K1_TILE - 1) / K1_TILE), dim3(K1_THREADS), L.nb[0] * sizeof(uint32_t), s,
                         recs[0], n, partials, L.pack_blocks, L.nb[0], boxes, hist[0], ticket0, plans);
      hipLaunchKernelGGL(sel_plan0_kernel, dim3(1), dim3(K2_THREADS), 0, s, n, L.nb[0], hist[0], plans);
    }
  }
  for (uint32_t d = 0; d < W; ++d) {
    const uint32_t nodes = 1u << d, off = nodes - 1u;  // heap offset of the level in the per-node tables
    const uint32_t bpn = (max_len_at(n, d) + K1_TILE - 1) / K1_TILE;
    const uint32_t nb_next = d + 1 < W ? L.nb[d + 1] : 0u;
    const bool placing = place && d < L.place_levels;
    const uint32_t cap_d = placing ? wide_cap : 0xffffffffu;
    SelWide* wide_d = wide + (placing ? off : 0u);
    // (a level that may place an oversized bucket with the whole chip keeps its three launches)
    const bool fused = fuse && !placing;
    if (fused) {
      hipLaunchKernelGGL(sel_split_kernel<true>, dim3(nodes * bpn), dim3(K1_THREADS), K2_LDS_BYTES, s, recs[d & 1], recs[(d + 1) & 1], mid, n,
                         d, bpn, L.nb[d], nb_next, plans + off, boxes + off, cursors + 4 * (size_t)off * CURSOR_STRIDE, hist[(d + 1) & 1], flags,
                         cap_d, wide_d, whist, plans + (2 * nodes - 1u), boxes + (2 * nodes - 1u), t->d_split);
      continue;
    }
    hipLaunchKernelGGL(sel_split_kernel<false>, dim3(nodes * bpn), dim3(K1_THREADS),
                       (2 * nb_next + (placing ? 2 * NSUB : 0u)) * sizeof(uint32_t), s, recs[d & 1], recs[(d + 1) & 1], mid, n,
                       d, bpn, L.nb[d], nb_next, plans + off, boxes + off, cursors + 4 * (size_t)off * CURSOR_STRIDE, hist[(d + 1) & 1], flags,
                       cap_d, wide_d, whist, plans + (2 * nodes - 1u), boxes + (2 * nodes - 1u), t->d_split);
    if (placing)
      hipLaunchKernelGGL(sel_place_kernel, dim3(nodes * bpn), dim3(K1_THREADS), 2 * nb_next * sizeof(uint32_t), s, mid,
                         recs[d & 1], recs[(d + 1) & 1], n, d, bpn, L.nb[d], nb_next, plans + off, boxes + off,
                         hist[(d + 1) & 1], cap_d, wide_d, whist);
    hipLaunchKernelGGL(sel_resolve_kernel, dim3(nodes), dim3(K2_THREADS), K2_LDS_BYTES, s, mid, recs[d & 1],
                       recs[(d + 1) & 1], n, d, L.nb[d], nb_next, plans + off, boxes + off, plans + (2 * nodes - 1u),
                       boxes + (2 * nodes - 1u), hist[(d + 1) & 1], t->d_split, flags, cap_d, wide_d, whist);
  }
  // the in-block levels by the sorting network with its words in registers (measured fastest: DESIGN.md §5); diagnostics build:
  // A3D_KDTREE_SORTNET=lds (its words in LDS, the product of round 5) / =select (selection inside the block): the cross-checks
  int net = 2;
  if (const char* env = A3D_DIAG_ENV("A3D_KDTREE_SORTNET")) net = !strcmp(env, "lds") ? 1 : (!strcmp(env, "select") ? 0 : 2);
#define A3D_NARROW_LAUNCH(SLOTS, REGS)                                                                                  \
  hipLaunchKernelGGL((sel_narrow_kernel<SLOTS, REGS>), dim3(1u << W), dim3(SLOTS / 4), nw_lds_bytes<SLOTS>(), s, recs[W & 1], n, \
                     W, D, t->d_split, t->d_leaves, t->d_slot_of_point, flags)
#ifdef A3D_DIAGNOSTICS
#define A3D_NARROW_SELECT(SLOTS)                                                                                        \
  hipLaunchKernelGGL((sel_narrow_select_kernel<SLOTS>), dim3(1u << W), dim3(SLOTS / 4), ns_lds_bytes<SLOTS>(), s, recs[W & 1], n, \
                     W, D, t->d_split, t->d_leaves, t->d_slot_of_point, flags)
#else
#define A3D_NARROW_SELECT(SLOTS) A3D_NARROW_LAUNCH(SLOTS, false)
#endif
  if (L.narrow_len <= 1024) {
    if (net == 2) A3D_NARROW_LAUNCH(1024, true);
    else if (net == 1) A3D_NARROW_LAUNCH(1024, false);
    else A3D_NARROW_SELECT(1024);
  } else {
    if (net == 2) A3D_NARROW_LAUNCH(2048, true);
    else if (net == 1) A3D_NARROW_LAUNCH(2048, false);
    else A3D_NARROW_SELECT(2048);
  }
#undef A3D_NARROW_LAUNCH
#undef A3D_NARROW_SELECT
  A3D_HIP_TRY(hipGetLastError());
  if (done) A3D_HIP_TRY(hipEventRecord(done, s));
  // (page-locked words of the context: a copy into pageable memory is staged by the runtime and blocks for longer)
  volatile uint32_t* h_flags = t->ctx->pinned_words + a3d_context::PINNED_WORDS;
  A3D_HIP_TRY(hipMemcpyAsync((void*)h_flags, flags, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  A3D_HIP_TRY(hipStreamSynchronize(s));
  if (h_flags[FLAG_OVERSIZED]) {
    t->ctx->kd_wide_place.store(true), t->ctx->kd_quiet_builds.store(0);
  } else if (place_by_context && L.place_levels > 0) {  // (a cloud too small to hold such a bucket says nothing)
    if (t->ctx->kd_quiet_builds.fetch_add(1) + 1 >= KD_QUIET_BUILDS) t->ctx->kd_wide_place.store(false), t->ctx->kd_quiet_builds.store(0);
  }
  A3D_REQUIRE(!h_flags[FLAG_NAN], A3D_NAN_IN_INPUT,
              "NaN coordinate in kd-tree input (the reference panics in partial_cmp().unwrap())");
  return A3D_OK;
}

}  // namespace a3d

#ifdef A3D_TAIL_STAMPS
extern "C" int a3d_debug_sel_stamps(unsigned long long out[64], uint32_t next_level, uint32_t next_node) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sel_stamps), 64 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long zero[64] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sel_stamps), zero, sizeof(zero)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sel_stamp_level), &next_level, sizeof(uint32_t)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_sel_stamp_node), &next_node, sizeof(uint32_t)) == hipSuccess ? 0 : 1;
}
#endif
