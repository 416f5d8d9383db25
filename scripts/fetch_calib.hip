// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of image_icp_kernel (tuning aid, not part of
// the library): each kernel reads a known number of bytes exactly once, so  bytes / (FETCH_SIZE x 1024)  is the factor
// FETCH_SIZE must be multiplied by for that shape (MI355X_MICROARCH.md gives 2 for 16 B/lane; "other access widths are
// uncalibrated").   hipcc --offload-arch=gfx950 -O3 scripts/fetch_calib.hip -o scripts/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- scripts/fetch_calib ; python3 scripts/fetch_calib_summary.py out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef f32x3 __attribute__((aligned(4))) f32x3_u;
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every kernel: thread i reads element i, i + stride, ... (coalesced across the wave), n elements in all
__global__ void calib_b16(const float4* __restrict__ a, size_t n, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}
__global__ void calib_b12(const float* __restrict__ a, size_t n, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { f32x3 v = *(const f32x3_u*)(a + 3 * i); s += v.x + v.y + v.z; }
  if (s == 12345.678f) out[0] = s;
}
__global__ void calib_b8(const f32x2* __restrict__ a, size_t n, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { f32x2 v = a[i]; s += v.x + v.y; }
  if (s == 12345.678f) out[0] = s;
}
__global__ void calib_b4(const float* __restrict__ a, size_t n, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ void calib_b1(const unsigned char* __restrict__ a, size_t n, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += (float)a[i];
  if (s == 12345.678f) out[0] = s;
}
// the ICP kernel's mix per pixel: 12 + 1 + 1 (source), 12 + 12 + 1 (target, same index), 2 x 8 (two map rows)
__global__ void calib_mix(const float* __restrict__ sp, const unsigned char* __restrict__ sm, const unsigned char* __restrict__ si,
                          const float* __restrict__ tp, const float* __restrict__ tn, const unsigned char* __restrict__ tm,
                          const float* __restrict__ imap, size_t n, unsigned w, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    f32x3 a = *(const f32x3_u*)(sp + 3 * i), b = *(const f32x3_u*)(tp + 3 * i), c = *(const f32x3_u*)(tn + 3 * i);
    size_t r = i / w, col = i % w;
    const float* q = imap + r * (w + 2) + col;
    s += a.x + a.y + a.z + b.x + b.y + b.z + c.x + c.y + c.z + (float)sm[i] + (float)si[i] + (float)tm[i] + q[0] + q[1] + q[w + 2] + q[w + 3];
  }
  if (s == 12345.678f) out[0] = s;
}

int main() {
  const size_t BYTES = 768ull << 20;  // three times the Infinity Cache
  void* buf; float* out;
  CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 0, BYTES));
  const dim3 grid(8192), block(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_b16, grid, block, 0, 0, (const float4*)buf, BYTES / 16, out);
    hipLaunchKernelGGL(calib_b12, grid, block, 0, 0, (const float*)buf, BYTES / 12, out);
    hipLaunchKernelGGL(calib_b8, grid, block, 0, 0, (const f32x2*)buf, BYTES / 8, out);
    hipLaunchKernelGGL(calib_b4, grid, block, 0, 0, (const float*)buf, BYTES / 4, out);
    hipLaunchKernelGGL(calib_b1, grid, block, 0, 0, (const unsigned char*)buf, BYTES / 4, out);  // a quarter of the buffer, byte by byte
    // mix: 48 pairs' level-0 arrays laid out back to back (n pixels): sp, tp, tn 12 n each; sm, si, tm n each; imap
    const unsigned w = 640; const size_t n = 48ull * 640 * 480, rows = n / w;
    char* p = (char*)buf;
    const float* sp = (const float*)p; p += 12 * n; const float* tp = (const float*)p; p += 12 * n; const float* tn = (const float*)p; p += 12 * n;
    const unsigned char* sm = (const unsigned char*)p; p += n; const unsigned char* si = (const unsigned char*)p; p += n; const unsigned char* tm = (const unsigned char*)p; p += n;
    const float* imap = (const float*)p;  // (rows + 2) x (w + 2) floats
    if ((size_t)(p - (char*)buf) + (rows + 2) * (w + 2) * 4 > BYTES) { printf("buffer too small\n"); return 1; }
    hipLaunchKernelGGL(calib_mix, grid, block, 0, 0, sp, sm, si, tp, tn, tm, imap, n, w, out);
  }
  CK(hipDeviceSynchronize());
  const size_t n = 48ull * 640 * 480;
  printf("expected_bytes calib_b16 %zu\nexpected_bytes calib_b12 %zu\nexpected_bytes calib_b8 %zu\nexpected_bytes calib_b4 %zu\nexpected_bytes calib_b1 %zu\nexpected_bytes calib_mix %zu\n",
         BYTES, BYTES / 12 * 12, BYTES, BYTES, BYTES / 4, 39 * n + 4 * (n / 640 + 2) * 642);
  return 0;
}
