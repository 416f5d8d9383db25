// Range images resident in HBM: upload and RangeImage::compute_normals (src/range_image/structure.rs:184-262) as an LDS-tiled stencil.
#include "common.hpp"

using namespace a3d;

namespace {

// ---- compute_normals ----------------------------------------------------------------------------
constexpr int TILE_W = 32, TILE_H = 8;  // 256 threads; LDS tile (TILE_W+2) x (TILE_H+2) with 1-px halo

// get_point(...).unwrap_or_else(zeros): the point if in range and mask == 1, else (0,0,0).
__device__ __forceinline__ V3 masked_point(const float* __restrict__ points, const uint8_t* __restrict__ mask,
                                           int w, int h, int row, int col) {
  if (row < 0 || col < 0 || row >= h || col >= w) return {0.f, 0.f, 0.f};
  int idx = row * w + col;
  if (mask[idx] != 1) return {0.f, 0.f, 0.f};
  return {points[3 * idx], points[3 * idx + 1], points[3 * idx + 2]};
}

__global__ void __launch_bounds__(TILE_W* TILE_H)
    compute_normals_kernel(const float* __restrict__ points, const uint8_t* __restrict__ mask,
                           float* __restrict__ normals, int w, int h) {
  __shared__ float tile[3][TILE_H + 2][TILE_W + 3];  // SoA planes; +3 keeps rows off the same banks
  const int tx = threadIdx.x % TILE_W, ty = threadIdx.x / TILE_W;
  const int col0 = blockIdx.x * TILE_W, row0 = blockIdx.y * TILE_H;
  // cooperative load of the haloed tile, masked (neighbour semantics)
  for (int t = threadIdx.x; t < (TILE_W + 2) * (TILE_H + 2); t += TILE_W * TILE_H) {
    int lx = t % (TILE_W + 2), ly = t / (TILE_W + 2);
    V3 p = masked_point(points, mask, w, h, row0 + ly - 1, col0 + lx - 1);
    tile[0][ly][lx] = p.x;
    tile[1][ly][lx] = p.y;
    tile[2][ly][lx] = p.z;
  }
  __syncthreads();
  const int col = col0 + tx, row = row0 + ty;
  if (col >= w || row >= h) return;
  const int idx = row * w + col;
  // the centre is read raw: its mask is NOT checked (structure.rs:207)
  V3 center{points[3 * idx], points[3 * idx + 1], points[3 * idx + 2]};
  auto at = [&](int ly, int lx) { return V3{tile[0][ly][lx], tile[1][ly][lx], tile[2][ly][lx]}; };
  V3 left = at(ty + 1, tx), right = at(ty + 1, tx + 2);
  V3 top = at(ty, tx + 1), bottom = at(ty + 2, tx + 1);
  const V3 out = normal_from_neighbours(center, left, right, top, bottom);
  normals[3 * idx] = out.x;
  normals[3 * idx + 1] = out.y;
  normals[3 * idx + 2] = out.z;
}

template <typename T>
a3d_status upload_array(a3d_context* ctx, const T* host, size_t count, T** dev) {
  A3D_HIP_TRY(hipMalloc((void**)dev, count * sizeof(T)));
  A3D_HIP_TRY(hipMemcpyAsync(*dev, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return A3D_OK;
}

a3d_status launch_compute_normals(a3d_context* ctx, const float* points, const uint8_t* mask, float* normals,
                                  uint32_t w, uint32_t h) {
  dim3 grid((w + TILE_W - 1) / TILE_W, (h + TILE_H - 1) / TILE_H);
  hipLaunchKernelGGL(compute_normals_kernel, grid, dim3(TILE_W * TILE_H), 0, ctx->stream, points, mask, normals,
                     (int)w, (int)h);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

}  // namespace

namespace a3d {
a3d_status compute_normals_device(a3d_context* ctx, const float* d_points, const uint8_t* d_mask, float* d_normals,
                                  uint32_t w, uint32_t h) {
  return launch_compute_normals(ctx, d_points, d_mask, d_normals, w, h);
}
}  // namespace a3d

extern "C" {

// RangeImages as the host holds them -> HBM.  All levels of a pyramid share ONE arena from the context's pool (no
// hipMalloc / hipFree in a steady stream of calls: each of those synchronises the whole device), every array is one
// asynchronous copy on the context's stream straight from the caller's memory (a DMA at the PCIe rate when that is
// page-locked: a3d_host_alloc), one synchronisation at the end.
a3d_status a3d_range_image_upload_pyramid(a3d_context* ctx, const a3d_range_image_view* views, uint64_t n_levels,
                                          a3d_device_image** out_images) {
  A3D_REQUIRE(ctx && views && out_images && n_levels > 0 && n_levels <= 16, A3D_INVALID_PARAMETER, "bad argument");
  size_t total = 0;
  auto take = [&](size_t bytes) {
    const size_t at = total;
    total += ((bytes + 255) / 256) * 256;
    return at;
  };
  struct Offsets {
    size_t points, mask, normals, intensities, imap;
  };
  std::vector<Offsets> off(n_levels);
  for (uint64_t l = 0; l < n_levels; ++l) {
    const a3d_range_image_view* v = &views[l];
    A3D_REQUIRE(v->points && v->mask, A3D_INVALID_PARAMETER, "RangeImage needs points and mask");
    A3D_REQUIRE(v->width > 0 && v->height > 0 && v->width * v->height < (1ull << 28), A3D_INVALID_PARAMETER,
                "bad image size (at most 2^28 pixels: the kernels address the arrays with 32-bit byte offsets)");
    // the kernels form texel offsets with 24-bit multiplies
    A3D_REQUIRE(v->width < (1ull << 23) && v->height < (1ull << 23), A3D_INVALID_PARAMETER, "image side too long");
    const size_t n = (size_t)v->width * v->height;
    off[l].points = take(n * 12), off[l].mask = take(n);
    off[l].normals = v->normals ? take(n * 12) : 0;
    off[l].intensities = v->intensities ? take(n) : 0;
    off[l].imap = v->intensity_map ? take((size_t)(v->width + 2) * (v->height + 2) * 4) : 0;
  }
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  DeviceArena* arena = new DeviceArena();
  if (ctx_arena_acquire(ctx, total, arena) != A3D_OK) {
    delete arena;
    return A3D_HIP_ERROR;
  }
  char* base = (char*)arena->base;
  hipStream_t s = ctx->stream;
  bool ok = true;
  auto copy = [&](void* dst, const void* src, size_t bytes) {
    if (ok && hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s) != hipSuccess) ok = false;
  };
  for (uint64_t l = 0; l < n_levels; ++l) {
    const a3d_range_image_view* v = &views[l];
    a3d_device_image* im = new a3d_device_image();
    im->ctx = ctx, im->arena = arena;
    ++arena->refs;
    im->width = (uint32_t)v->width, im->height = (uint32_t)v->height;
    im->fx64 = v->fx, im->fy64 = v->fy, im->cx64 = v->cx, im->cy64 = v->cy;
    im->fx = (float)v->fx, im->fy = (float)v->fy, im->cx = (float)v->cx, im->cy = (float)v->cy;
    const size_t n = (size_t)im->width * im->height;
    im->points = (float*)(base + off[l].points), im->mask = (uint8_t*)(base + off[l].mask);
    copy(im->points, v->points, n * 12);
    copy(im->mask, v->mask, n);
    if (v->normals) {
      im->normals = (float*)(base + off[l].normals), im->has_normals = true;
      copy(im->normals, v->normals, n * 12);
    }
    if (v->intensities) {
      im->intensities = (uint8_t*)(base + off[l].intensities), im->has_intensities = true;
      copy(im->intensities, v->intensities, n);
    }
    if (v->intensity_map) {
      im->imap = (float*)(base + off[l].imap), im->has_imap = true;
      copy(im->imap, v->intensity_map, (size_t)(im->width + 2) * (im->height + 2) * 4);
    }
    out_images[l] = im;
  }
  if (hipStreamSynchronize(s) != hipSuccess) ok = false;
  if (!ok) {
    set_error("a3d_range_image_upload: HIP failure: %s", hipGetErrorString(hipGetLastError()));
    for (uint64_t l = 0; l < n_levels; ++l) a3d_range_image_free(out_images[l]), out_images[l] = nullptr;
    return A3D_HIP_ERROR;
  }
  return A3D_OK;
}

a3d_status a3d_range_image_upload(a3d_context* ctx, const a3d_range_image_view* v, a3d_device_image** out) {
  A3D_REQUIRE(ctx && v && out, A3D_INVALID_PARAMETER, "null argument");
  return a3d_range_image_upload_pyramid(ctx, v, 1, out);
}

a3d_status a3d_range_image_free(a3d_device_image* im) {
  if (!im) return A3D_OK;
  // No wait for the context's stream here: whoever enqueued work on this image without waiting for it registered a
  // fence with the image's arena (enqueue-only batch alignments, compute_normals below), every other entry point
  // returns with its work complete, and hipFree (images without an arena) waits for the device by itself.  A thread
  // building the next frames on this context therefore does not stall the thread that frees the previous ones.
  if (im->arena) {  // arrays live in a shared arena: release it with its last user
    if (im->own_normals) hipFree(im->normals);  // (waits for the device by itself)
    if (--im->arena->refs == 0) {
      ctx_arena_release(im->ctx, im->arena);
      delete im->arena;
    }
  } else {
    hipFree(im->points);
    hipFree(im->mask);
    hipFree(im->normals);
    hipFree(im->intensities);
    hipFree(im->imap);
    hipFree(im->colors);
  }
  delete im;
  return A3D_OK;
}

a3d_status a3d_range_image_compute_normals(a3d_device_image* im) {
  if (im) hipSetDevice(im->ctx->device);
  A3D_REQUIRE(im, A3D_INVALID_PARAMETER, "image is null");
  const size_t n = (size_t)im->width * im->height;
  A3D_REQUIRE(im->normals || !im->arena || !im->built, A3D_INVALID_PARAMETER,
              "this image was built without normals (a3d_builder_params.with_normals = 0)");
  if (!im->normals) {  // an uploaded image that came without normals: they get their own allocation, freed with it
    A3D_HIP_TRY(hipMalloc((void**)&im->normals, n * 3 * sizeof(float)));
    im->own_normals = true;
  }
  A3D_TRY(launch_compute_normals(im->ctx, im->points, im->mask, im->normals, im->width, im->height));
  // enqueue-only: the arena must outlive the launch.  The arena's own fence is recorded right behind the launch
  // (an event record costs ~1 us), so that freeing the image later waits for THIS launch only — not for whatever
  // else (the next frames' builds) has been enqueued on the context's stream by then.
  if (im->arena) {
    if (!im->arena->self_fence) im->arena->self_fence = std::make_shared<UseFence>();
    im->arena->self_fence->record(im->ctx->stream);
    attach_fence(im, im->arena->self_fence);
  }
  im->has_normals = true;
  return A3D_OK;
}

a3d_status a3d_range_image_download_normals(a3d_device_image* im, float* out) {
  if (im) hipSetDevice(im->ctx->device);
  A3D_REQUIRE(im && out, A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(im->has_normals, A3D_MISSING_FIELD, "image has no normals");
  const size_t n = (size_t)im->width * im->height;
  A3D_HIP_TRY(hipMemcpyAsync(out, im->normals, n * 3 * sizeof(float), hipMemcpyDeviceToHost, im->ctx->stream));
  A3D_HIP_TRY(hipStreamSynchronize(im->ctx->stream));
  return A3D_OK;
}

a3d_status a3d_compute_normals(a3d_context* ctx, const float* points, const uint8_t* mask, uint64_t width,
                               uint64_t height, float* out_normals) {
  A3D_REQUIRE(ctx && points && mask && out_normals, A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(width > 0 && height > 0 && width * height < (1ull << 30), A3D_INVALID_PARAMETER, "bad image size");
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  const size_t n = width * height;
  float *d_points = nullptr, *d_normals = nullptr;
  uint8_t* d_mask = nullptr;
  a3d_status st = upload_array(ctx, points, n * 3, &d_points);
  if (st == A3D_OK) st = upload_array(ctx, mask, n, &d_mask);
  if (st == A3D_OK && hipMalloc((void**)&d_normals, n * 3 * sizeof(float)) != hipSuccess) st = A3D_HIP_ERROR;
  if (st == A3D_OK) st = launch_compute_normals(ctx, d_points, d_mask, d_normals, (uint32_t)width, (uint32_t)height);
  if (st == A3D_OK &&
      hipMemcpyAsync(out_normals, d_normals, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
    st = A3D_HIP_ERROR;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) st = A3D_HIP_ERROR;
  hipFree(d_points);
  hipFree(d_mask);
  hipFree(d_normals);
  if (st == A3D_HIP_ERROR) set_error("a3d_compute_normals: HIP failure");
  return st;
}

}  // extern "C"
