// The two stable sorts of the SORTING kd-tree build (kdtree_build.hip; diagnostics build only since round 5: the
// product builds by selection, kdtree_select.hip), hand-written for gfx950.
//
//   wide levels   (ranges longer than 4096 points; at most ~n / 4096 of them): ONE device-wide stable LSD radix
//                 sort per level on 64-bit keys `range number << 32 | order-preserving key bits`, 8 bits per pass:
//                 per-block digit histograms -> one exclusive scan (digit-major, block-minor) -> stable scatter.
//                 Inside a block the scatter ranks items round by round (256 consecutive items per round) with a
//                 wave-level multi-split: eight ballots find the lanes that share a digit, a popcount of the lower
//                 lanes is the rank inside the wave, and per-wave counts in LDS order the four waves.
//   narrow levels (every range fits in 4096 points): each block sorts 4096 / CAP ranges at once in LDS with a
//                 bitonic network on 64-bit words `key bits << 32 | position in the range`.  The position makes
//                 every word unique, so the unstable network yields exactly the stable order.
//
// Both give the order of `slice::sort_by(partial_cmp)` (src/kdtree.rs:41-45): keys are canonicalised (-0.0 -> +0.0)
// before the monotone float -> u32 map, equal keys keep their previous order.
#include <cstdlib>
#include <cstring>

#include "kdtree.hpp"

#ifdef A3D_DIAGNOSTICS

using namespace a3d;

namespace {

constexpr uint32_t ITEMS = 4096;        // items per block of the LDS bitonic sort (256 threads x 16)
constexpr uint32_t RADIX_ITEMS = 2048;  // items per block of the radix passes: 245 blocks at 500k keys (the chip has 256 CUs)

__device__ __forceinline__ uint32_t ordered_bits(float v) {
  const uint32_t u = __float_as_uint(v + 0.0f);  // -0.0 -> +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- device-wide LSD radix sort, one 8-bit digit per pass ---------------------------------------------------

// FUSED: the table is block-major (hist[block][digit]) and no scan kernel follows — every scatter block derives its
// own offsets from the table (below).  Otherwise digit-major for the one-block scan.
template <bool FUSED>
__global__ void __launch_bounds__(256)
    radix_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, int shift, uint32_t nblocks,
                      uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * RADIX_ITEMS;
#pragma unroll 4
  for (uint32_t r = 0; r < RADIX_ITEMS / 256; ++r) {
    const uint32_t i = base + r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (FUSED) hist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
  else hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Exclusive scan of `m` counters in place, one block of 1024 threads, each owning one contiguous chunk.
// IN_LDS: the whole array is staged in LDS (coalesced in, coalesced out; m * 4 bytes <= SCAN_LDS_BYTES), so the
// chunk walks do not go to memory with a stride; otherwise the chunks are walked in global memory.
constexpr uint32_t SCAN_LDS_WORDS = 36 * 1024;  // 144 KiB of the CU's 160 KiB
template <bool IN_LDS>
__global__ void __launch_bounds__(1024) exclusive_scan_kernel(uint32_t* __restrict__ data, uint32_t m) {
  extern __shared__ uint32_t staged[];
  __shared__ uint32_t sums[1024];
  const uint32_t t = threadIdx.x;
  if (IN_LDS) {
    for (uint32_t i = t; i < m; i += 1024) staged[i] = data[i];
    __syncthreads();
  }
  uint32_t* src = IN_LDS ? staged : data;
  // an odd chunk length keeps the 1024 chunk walks on different LDS banks
  const uint32_t chunk = ((m + 1023) / 1024) | 1u, lo = min(t * chunk, m), hi = min(lo + chunk, m);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; ++i) s += src[i];
  sums[t] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele over the 1024 chunk sums
    const uint32_t v = t >= off ? sums[t - off] : 0u;
    __syncthreads();
    sums[t] += v;
    __syncthreads();
  }
  uint32_t run = t ? sums[t - 1] : 0u;
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t v = src[i];
    src[i] = run;
    run += v;
  }
  if (IN_LDS) {
    __syncthreads();
    for (uint32_t i = t; i < m; i += 1024) data[i] = staged[i];
  }
}

// FUSED: `offsets` is the raw block-major histogram table.  Thread d of every block walks column d of the table
// (coalesced: 256 consecutive words per row): the counts of the blocks before this one + the digit's total, then one
// 256-wide exclusive scan of the totals in LDS gives the digit's start — the scan kernel of the unfused form (a single
// block, 43 us per pass at 500k points: half of the whole tree build) disappears; the table (nblocks KiB) stays in L2.
template <bool FUSED>
__global__ void __launch_bounds__(256)
    radix_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                         uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                         uint32_t nblocks, const uint32_t* __restrict__ offsets) {
  __shared__ uint32_t base[256];      // where this block's next item with digit d goes
  __shared__ uint32_t wcount[4][256]; // items with digit d in wave w, this round
  const uint32_t t = threadIdx.x, w = t >> 6;
  const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const unsigned long long lower = (1ull << lane) - 1ull;
  if (FUSED) {
    uint32_t below = 0, total = 0;
    uint32_t b = 0;
    for (; b + 8 <= nblocks; b += 8) {  // eight rows in flight
      uint32_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = offsets[(size_t)(b + k) * 256 + t];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        total += v[k];
        below += (b + k < blockIdx.x) ? v[k] : 0u;
      }
    }
    for (; b < nblocks; ++b) {
      const uint32_t v = offsets[(size_t)b * 256 + t];
      total += v;
      below += b < blockIdx.x ? v : 0u;
    }
    // exclusive scan of the 256 digit totals: inside each wave with shuffles, then across the four waves
    uint32_t incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
      if (lane >= (uint32_t)off) incl += up;
    }
    if (lane == 63) wcount[0][w] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t ww = 0; ww < w; ++ww) before += wcount[0][ww];
    base[t] = before + incl - total + below;
    __syncthreads();  // wcount[0][0..3] are rewritten below
  } else {
    base[t] = offsets[(size_t)t * nblocks + blockIdx.x];
  }
  const uint32_t first = blockIdx.x * RADIX_ITEMS;
  for (uint32_t r = 0; r < RADIX_ITEMS / 256; ++r) {
    wcount[0][t] = 0, wcount[1][t] = 0, wcount[2][t] = 0, wcount[3][t] = 0;
    __syncthreads();  // also orders the previous round's base update
    const uint32_t i = first + r * 256 + t;
    const bool valid = i < n;
    const uint64_t key = valid ? keys_in[i] : 0ull;
    const uint32_t d = (uint32_t)(key >> shift) & 255u;
    // lanes of this wave holding the same digit
    unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool b = (d >> bit) & 1u;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(b);
      peers &= b ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__builtin_popcountll(peers & lower);
    if (valid && rank == 0) wcount[w][d] = (uint32_t)__builtin_popcountll(peers);
    __syncthreads();
    if (valid) {
      uint32_t pos = base[d] + rank;
      for (uint32_t ww = 0; ww < w; ++ww) pos += wcount[ww][d];
      keys_out[pos] = key;
      vals_out[pos] = vals_in[i];
    }
    __syncthreads();
    base[t] += wcount[0][t] + wcount[1][t] + wcount[2][t] + wcount[3][t];
  }
}

// ---- narrow levels: bitonic network on (key bits, position) words, 4096 / CAP ranges per block ---------------

// Range of node `j` at `level` (same recursion as kdtree_build.hip)
__device__ __forceinline__ void node_range(uint32_t n, uint32_t level, uint32_t j, uint32_t* start, uint32_t* len,
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

__global__ void __launch_bounds__(256)
    sort_ranges_kernel(const float* __restrict__ points, const uint32_t* __restrict__ idx_in,
                       uint32_t* __restrict__ idx_out, uint32_t n, uint32_t level, int k, uint32_t cap_log2,
                       uint32_t* __restrict__ nan_flag) {
  __shared__ uint64_t slot[ITEMS];
  __shared__ uint32_t seg_start[ITEMS / 32], seg_len[ITEMS / 32];  // CAP >= 32
  const uint32_t cap = 1u << cap_log2, segs = ITEMS >> cap_log2, t = threadIdx.x;
  const uint32_t nodes = 1u << level, j0 = blockIdx.x * segs;
  if (t < segs) {
    uint32_t s = 0, l = 0;
    bool ok = false;
    if (j0 + t < nodes) node_range(n, level, j0 + t, &s, &l, &ok);
    const bool sort = ok && l > 16;  // leaves (and nodes below a leaf) are not sorted
    seg_start[t] = s;
    seg_len[t] = sort ? l : 0u;
  }
  __syncthreads();
  for (uint32_t q = 0; q < ITEMS / 256; ++q) {
    const uint32_t sidx = q * 256 + t, c = sidx >> cap_log2, r = sidx & (cap - 1);
    uint64_t word = ~0ull;  // padding sorts behind every real word
    if (r < seg_len[c]) {
      const float v = points[3 * (size_t)idx_in[seg_start[c] + r] + k];
      if (v != v) atomicOr(nan_flag, 1u);  // partial_cmp().unwrap() would panic (kdtree.rs:43)
      word = ((uint64_t)ordered_bits(v) << 32) | r;
    }
    slot[sidx] = word;
  }
  __syncthreads();
  for (uint32_t kk = 2; kk <= cap; kk <<= 1)
    for (int jl = 31 - __builtin_clz(kk >> 1); jl >= 0; --jl) {  // partner distance j = 2^jl
      const uint32_t j = 1u << jl;
#pragma unroll 4
      for (uint32_t q = 0; q < ITEMS / 512; ++q) {
        const uint32_t p = q * 256 + t;  // pair number
        const uint32_t i = ((p >> jl) << (jl + 1)) | (p & (j - 1)), i2 = i + j;  // j < cap: stays inside a range
        const bool up = ((i & (cap - 1)) & kk) == 0;
        const uint64_t a = slot[i], b = slot[i2];
        if ((a > b) == up) slot[i] = b, slot[i2] = a;
      }
      __syncthreads();
    }
  for (uint32_t q = 0; q < ITEMS / 256; ++q) {
    const uint32_t sidx = q * 256 + t, c = sidx >> cap_log2, r = sidx & (cap - 1);
    if (r < seg_len[c]) idx_out[seg_start[c] + r] = idx_in[seg_start[c] + (uint32_t)slot[sidx]];
  }
}

}  // namespace

namespace a3d {

size_t kdtree_sort_scratch_bytes(uint32_t n) {
  const size_t nblocks = ((size_t)n + RADIX_ITEMS - 1) / RADIX_ITEMS;
  return 256 * nblocks * sizeof(uint32_t);
}

// Stable sort of (keys, vals) by bits [0, end_bit) of the 64-bit keys.  Ping-pongs between the a and b buffers;
// *in_b tells where the result is.
a3d_status kdtree_radix_sort_pairs(hipStream_t s, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a,
                                   uint32_t* vals_b, uint32_t n, int end_bit, uint32_t* hist, bool* in_b) {
  const uint32_t nblocks = (n + RADIX_ITEMS - 1) / RADIX_ITEMS;
  bool flip = false;
  for (int shift = 0; shift < end_bit; shift += 8) {
    const uint64_t* kin = flip ? keys_b : keys_a;
    const uint32_t* vin = flip ? vals_b : vals_a;
    const char* scan_env = A3D_DIAG_ENV("A3D_KDTREE_SCAN");  // diagnostics build: force the > 2 M-key form at test sizes
    const bool force_unfused = scan_env && !strcmp(scan_env, "unfused");
    if (nblocks <= 1024 && !force_unfused) {  // up to 2M keys: every scatter block reads the whole table (at most 1 MiB, L2-resident)
      hipLaunchKernelGGL(radix_hist_kernel<true>, dim3(nblocks), dim3(256), 0, s, kin, n, shift, nblocks, hist);
      hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(nblocks), dim3(256), 0, s, kin, vin, flip ? keys_a : keys_b,
                         flip ? vals_a : vals_b, n, shift, nblocks, hist);
      flip = !flip;
      continue;
    }
    hipLaunchKernelGGL(radix_hist_kernel<false>, dim3(nblocks), dim3(256), 0, s, kin, n, shift, nblocks, hist);
    const uint32_t m = 256u * nblocks;
    if (m <= SCAN_LDS_WORDS) {
      static bool big_lds_allowed = false;  // more than 64 KiB of dynamic LDS has to be requested once
      if (!big_lds_allowed) {
        A3D_HIP_TRY(hipFuncSetAttribute((const void*)exclusive_scan_kernel<true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_LDS_WORDS * sizeof(uint32_t)));
        big_lds_allowed = true;
      }
      hipLaunchKernelGGL(exclusive_scan_kernel<true>, dim3(1), dim3(1024), m * sizeof(uint32_t), s, hist, m);
    } else
      hipLaunchKernelGGL(exclusive_scan_kernel<false>, dim3(1), dim3(1024), 0, s, hist, m);
    hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(nblocks), dim3(256), 0, s, kin, vin, flip ? keys_a : keys_b,
                       flip ? vals_a : vals_b, n, shift, nblocks, hist);
    flip = !flip;
  }
  A3D_HIP_TRY(hipGetLastError());
  *in_b = flip;
  return A3D_OK;
}

// Sorts every range of `level` (all of them at most 2^cap_log2 <= 4096 long) from idx_in into idx_out; positions
// outside the sorted ranges must have been copied by the caller.
a3d_status kdtree_sort_ranges(hipStream_t s, const float* points, const uint32_t* idx_in, uint32_t* idx_out, uint32_t n,
                              uint32_t level, int k, uint32_t cap_log2, uint32_t* nan_flag) {
  const uint32_t segs = ITEMS >> cap_log2, nodes = 1u << level;
  hipLaunchKernelGGL(sort_ranges_kernel, dim3((nodes + segs - 1) / segs), dim3(256), 0, s, points, idx_in, idx_out, n,
                     level, k, cap_log2, nan_flag);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

}  // namespace a3d

#endif  // A3D_DIAGNOSTICS
