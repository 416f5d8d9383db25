#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int M> __device__ __forceinline__ uint32_t lane_xor32(uint32_t v) {
  if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);
  else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);
  else if constexpr (M == 4) {
    int r = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)v, 0x114, 0xF, 0xA, false);
  } else if constexpr (M == 8) {
    int r = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x108, 0xF, 0x3, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)v, 0x118, 0xF, 0xC, false);
  } else if constexpr (M == 16) {
    auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16) ? p[0] : p[1];
  } else {
    auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32) ? p[0] : p[1];
  }
}
__global__ void k(uint32_t* out) {
  uint32_t v = threadIdx.x * 3 + 7;
  out[0 * 64 + threadIdx.x] = lane_xor32<1>(v);
  out[1 * 64 + threadIdx.x] = lane_xor32<2>(v);
  out[2 * 64 + threadIdx.x] = lane_xor32<4>(v);
  out[3 * 64 + threadIdx.x] = lane_xor32<8>(v);
  out[4 * 64 + threadIdx.x] = lane_xor32<16>(v);
  out[5 * 64 + threadIdx.x] = lane_xor32<32>(v);
}
int main() {
  uint32_t* d; hipMalloc(&d, 6 * 64 * 4);
  k<<<1, 64>>>(d);
  uint32_t h[6 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 6; ++m) for (int l = 0; l < 64; ++l) if (h[m * 64 + l] != (uint32_t)((l ^ (1 << m)) * 3 + 7)) { if (bad < 10) printf("m=%d lane %d got %u want %u\n", 1 << m, l, h[m*64+l], (l ^ (1 << m)) * 3 + 7); ++bad; }
  printf("bad=%d\n", bad);
  return bad != 0;
}
