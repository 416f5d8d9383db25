// Range images resident in HBM: upload and RangeImage::compute_normals (src/range_image/structure.rs:184-262) as an LDS-tiled stencil.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"

using namespace a3d;

namespace {

// ---- compute_normals ----------------------------------------------------------------------------
// One block = a 64 x 16 pixel tile of one frame (blockIdx.z = frame of a batch), 256 threads, four pixels per thread
// (rows ty, ty + 4, ty + 8, ty + 12).  Every point is ONE 12-byte load (global_load_dwordx3: three dword loads at a
// 12-byte lane stride make the L1 look up every line of the wave's span three times) and every normal ONE 12-byte
// store; the tile + a one-pixel halo (16 % more loads) is staged in LDS as SoA planes with the mask applied
// (get_point(...).unwrap_or_else(zeros): neighbours that are out of range or whose mask != 1 read as (0,0,0)); the
// centre is used raw from the thread's own registers — its mask is NOT checked (structure.rs:207).
// 25 algorithmic bytes per pixel (12 + 1 read, 12 written).
// Two shapes: 64 x 16 tiles with four pixels per thread for a batch (16 % halo loads), 32 x 8 tiles with one pixel per
// thread for a frame or two (1200 blocks for a 640 x 480 frame instead of 300: a lone frame is latency-bound).
constexpr int NORMALS_MAX_BATCH = 64;
struct NormalsBatch {
  const float* points[NORMALS_MAX_BATCH];
  const uint8_t* mask[NORMALS_MAX_BATCH];
  float* normals[NORMALS_MAX_BATCH];
};
typedef float nf32x3 __attribute__((ext_vector_type(3)));
typedef nf32x3 __attribute__((aligned(4))) nf32x3_u;

// XCD-aware tile order (1-D grid): workgroups are dealt to the eight XCDs round robin (block b: group b % 8) and every XCD
// has its own L2, so with the plain order the four neighbours of a tile run on four other XCDs and every halo line —
// a 128-byte line for one 12-byte point — is fetched from the fabric again by each of them: 421 MB of reads per 64-frame
// launch against 256 MB of points and masks (`profiles/round4_normals_traffic.json`), at 6.4 TB/s of fabric traffic.
// Here group g takes the g-th CONTIGUOUS eighth of the tiles (whole frames at 64 frames per launch), so a tile's
// neighbours hit in the L2 that already holds their lines.  The remap is bijective for any tile count (CDNA guide, T1).
__device__ __forceinline__ uint32_t xcd_contiguous_tile(uint32_t b, uint32_t n) { return xcd_contiguous_index(b, n); }
template <int NT_W, int NT_H, int NT_PPT>
__global__ void __launch_bounds__(256)
    compute_normals_kernel(NormalsBatch batch, int w, int h, uint32_t tiles_x, uint32_t tiles_y) {
  static_assert(NT_W * NT_H / NT_PPT == 256, "256 threads per block");
  __shared__ float tile[3][NT_H + 2][NT_W + 3];  // SoA planes; +3 keeps rows off the same banks
  const uint32_t t = xcd_contiguous_tile(blockIdx.x, gridDim.x), per_frame = tiles_x * tiles_y;
  const uint32_t frame = t / per_frame, in_frame = t - frame * per_frame, tile_y = in_frame / tiles_x,
                 tile_x = in_frame - tile_y * tiles_x;
  const float* __restrict__ points = batch.points[frame];
  const uint8_t* __restrict__ mask = batch.mask[frame];
  float* __restrict__ normals = batch.normals[frame];
  const int tx = threadIdx.x % NT_W, ty = threadIdx.x / NT_W;  // ty in 0..3
  const int col0 = (int)tile_x * NT_W, row0 = (int)tile_y * NT_H;
  const int col = col0 + tx;
  // ---- the thread's own four pixels: raw centre kept in registers, masked copy into the tile ----
  V3 center[NT_PPT];
  bool inside[NT_PPT];
#pragma unroll
  for (int k = 0; k < NT_PPT; ++k) {
    const int row = row0 + ty + k * (NT_H / NT_PPT);
    inside[k] = col < w && row < h;
    const int idx = inside[k] ? row * w + col : 0;
    const nf32x3 p = *(const nf32x3_u*)(points + 3 * (size_t)idx);  // (unconditional loads: all four in flight at once)
    const uint8_t m = mask[idx];
    center[k] = V3{p.x, p.y, p.z};
    const bool ok = inside[k] && m == 1;
    const int ly = ty + k * (NT_H / NT_PPT) + 1;
    tile[0][ly][tx + 1] = ok ? p.x : 0.f, tile[1][ly][tx + 1] = ok ? p.y : 0.f, tile[2][ly][tx + 1] = ok ? p.z : 0.f;
  }
  // ---- the halo ring: 2 (NT_W + 2) + 2 NT_H = 164 pixels, one per thread of the first 164 ----
  {
    constexpr int RING = 2 * (NT_W + 2) + 2 * NT_H;
    static_assert(RING <= 2 * 256, "the halo ring takes at most two pixels per thread");
    for (int t = (int)threadIdx.x; t < RING; t += 256) {
      int ly, lx;
      if (t < NT_W + 2) ly = 0, lx = t;
      else if (t < 2 * (NT_W + 2)) ly = NT_H + 1, lx = t - (NT_W + 2);
      else ly = 1 + (t - 2 * (NT_W + 2)) / 2, lx = ((t - 2 * (NT_W + 2)) & 1) ? NT_W + 1 : 0;
      const int row = row0 + ly - 1, c = col0 + lx - 1;
      const bool in = row >= 0 && c >= 0 && row < h && c < w;
      const int idx = in ? row * w + c : 0;
      const nf32x3 p = *(const nf32x3_u*)(points + 3 * (size_t)idx);
      const bool ok = in && mask[idx] == 1;
      tile[0][ly][lx] = ok ? p.x : 0.f, tile[1][ly][lx] = ok ? p.y : 0.f, tile[2][ly][lx] = ok ? p.z : 0.f;
    }
  }
  __syncthreads();
  auto at = [&](int ly, int lx) { return V3{tile[0][ly][lx], tile[1][ly][lx], tile[2][ly][lx]}; };
#pragma unroll
  for (int k = 0; k < NT_PPT; ++k) {
    if (!inside[k]) continue;
    const int ly = ty + k * (NT_H / NT_PPT) + 1, lx = tx + 1;
    const V3 out = normal_from_neighbours_dev(center[k], at(ly, lx - 1), at(ly, lx + 1), at(ly - 1, lx), at(ly + 1, lx));
    const int row = row0 + ly - 1;
    // (a streaming store — nothing reads the normals back soon: 102 -> 95 us per 64 frames, 0.60 -> 0.64 of 8 TB/s;
    // streaming LOADS of the points gained nothing at 64 frames and lost at 16)
    __builtin_nontemporal_store(nf32x3{out.x, out.y, out.z}, (nf32x3_u*)(normals + 3 * ((size_t)row * w + col)));
  }
}

a3d_status launch_compute_normals_batch(a3d_context* ctx, const NormalsBatch& batch, uint32_t frames, uint32_t w, uint32_t h) {
  int shape = (uint64_t)frames * w * h >= 4ull * 640 * 480 ? 1 : 0;
  if (const char* env = A3D_DIAG_ENV("A3D_NORMALS_SHAPE")) shape = atoi(env);  // diagnostics build: tile shape sweep
  auto launch = [&](auto kernel, uint32_t tw, uint32_t th) {
    const uint32_t tiles_x = (w + tw - 1) / tw, tiles_y = (h + th - 1) / th;
    hipLaunchKernelGGL(kernel, dim3(tiles_x * tiles_y * frames), dim3(256), 0, ctx->stream, batch, (int)w, (int)h, tiles_x, tiles_y);
  };
  if (shape == 2) launch(compute_normals_kernel<64, 32, 8>, 64, 32);
  else if (shape == 3) launch(compute_normals_kernel<128, 8, 4>, 128, 8);
  else if (shape == 4) launch(compute_normals_kernel<64, 8, 2>, 64, 8);
  else if (shape == 1) launch(compute_normals_kernel<64, 16, 4>, 64, 16);
  else launch(compute_normals_kernel<32, 8, 1>, 32, 8);
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

// Enqueue-only work of a context on images that live in arenas: ONE (lazily recorded) fence per call, shared by the
// arenas of all the images the call touched; it replaces the fence an earlier call left on an arena (a later event on
// the same stream covers the earlier work), so repeated calls do not pile fences up.
void fence_self_work(a3d_context* ctx, a3d_device_image* const* images, uint64_t n) {
  std::shared_ptr<UseFence> fence;
  for (uint64_t i = 0; i < n; ++i) {
    DeviceArena* a = images[i]->arena;
    if (!a) continue;
    if (!fence) {
      fence = std::make_shared<UseFence>();
      fence->record_later(ctx->stream, ctx->device);  // (no event on the enqueue path: UseFence::record_later)
    }
    std::lock_guard<std::mutex> lock(a->fence_mutex);
    bool replaced = false;
    for (auto& f : a->fences)
      if (a->self_fence && f == a->self_fence) f = fence, replaced = true;
    if (!replaced && a->self_fence != fence) a->fences.push_back(fence);
    a->self_fence = fence;
  }
}

template <typename T>
a3d_status upload_array(a3d_context* ctx, const T* host, size_t count, T** dev) {
  A3D_HIP_TRY(hipMalloc((void**)dev, count * sizeof(T)));
  A3D_HIP_TRY(hipMemcpyAsync(*dev, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return A3D_OK;
}

a3d_status launch_compute_normals(a3d_context* ctx, const float* points, const uint8_t* mask, float* normals,
                                  uint32_t w, uint32_t h) {
  NormalsBatch b{};
  b.points[0] = points, b.mask[0] = mask, b.normals[0] = normals;
  return launch_compute_normals_batch(ctx, b, 1, w, h);
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
  return a3d::upload_pyramid(ctx, views, n_levels, out_images, nullptr);
}

}  // extern "C"

namespace a3d {
// level_events == nullptr: every array copied on the context's stream, complete on return (the public call).
// level_events != nullptr (a3d_multiscale_align_host): the copies go on the context's COPY stream, coarsest level first,
// with one event recorded behind each level's arrays (created here, destroyed by the caller) and NO wait: the alignment's
// launches of a level wait for that level's event only, so the coarse levels iterate under the upload of the fine ones.
a3d_status upload_pyramid(a3d_context* ctx, const a3d_range_image_view* views, uint64_t n_levels,
                          a3d_device_image** out_images, hipEvent_t* level_events) {
  A3D_REQUIRE(ctx && views && out_images && n_levels > 0 && n_levels <= 16, A3D_INVALID_PARAMETER, "bad argument");
  for (uint64_t l = 0; l < n_levels; ++l) out_images[l] = nullptr;
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
  hipStream_t s = level_events && ctx->copy_stream ? ctx->copy_stream : ctx->stream;
  bool ok = true;
  auto copy = [&](void* dst, const void* src, size_t bytes) {
    if (ok && hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s) != hipSuccess) ok = false;
  };
  if (level_events)
    for (uint64_t l = 0; l < n_levels; ++l) level_events[l] = nullptr;
  for (uint64_t step = 0; step < n_levels; ++step) {
    const uint64_t l = level_events ? n_levels - 1 - step : step;  // (overlapped: the level the alignment starts with first)
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
    if (level_events && ok) {
      if (hipEventCreateWithFlags(&level_events[l], hipEventDisableTiming) != hipSuccess ||
          hipEventRecord(level_events[l], s) != hipSuccess)
        ok = false;
    }
  }
  if ((!level_events || !ok) && hipStreamSynchronize(s) != hipSuccess) ok = false;
  if (!ok) {
    set_error("a3d_range_image_upload: HIP failure: %s", hipGetErrorString(hipGetLastError()));
    for (uint64_t l = 0; l < n_levels; ++l) {
      if (out_images[l]) a3d_range_image_free(out_images[l]);
      out_images[l] = nullptr;
      if (level_events && level_events[l]) hipEventDestroy(level_events[l]), level_events[l] = nullptr;
    }
    return A3D_HIP_ERROR;
  }
  return A3D_OK;
}
}  // namespace a3d

extern "C" {

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
  // enqueue-only: the arena must outlive the launch.  The arena gets a fence that is recorded when somebody first waits
  // for it (freeing the image, recycling the arena for another stream's build): nothing but the launch on this path.
  fence_self_work(im->ctx, &im, 1);
  im->has_normals = true;
  return A3D_OK;
}

// RangeImage::compute_normals for n resident images of one size in ONE launch per 64 images (enqueue-only, like the
// single-image call): the stencil's launch cost is paid once instead of per frame.
a3d_status a3d_range_image_compute_normals_batch(a3d_device_image* const* images, uint64_t n) {
  A3D_REQUIRE(images || n == 0, A3D_INVALID_PARAMETER, "null argument");
  if (n == 0) return A3D_OK;
  for (uint64_t i = 0; i < n; ++i) {
    A3D_REQUIRE(images[i], A3D_INVALID_PARAMETER, "image is null");
    A3D_REQUIRE(images[i]->ctx == images[0]->ctx && images[i]->width == images[0]->width &&
                    images[i]->height == images[0]->height,
                A3D_INVALID_PARAMETER, "a3d_range_image_compute_normals_batch: the images must share a context and a size");
    A3D_REQUIRE(images[i]->normals || !images[i]->arena || !images[i]->built, A3D_INVALID_PARAMETER,
                "this image was built without normals (a3d_builder_params.with_normals = 0)");
  }
  a3d_context* ctx = images[0]->ctx;
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  const size_t px = (size_t)images[0]->width * images[0]->height;
  for (uint64_t i = 0; i < n; ++i) {  // uploaded without normals: they get their own allocation, freed with the image
    a3d_device_image* im = images[i];
    if (!im->normals) {
      A3D_HIP_TRY(hipMalloc((void**)&im->normals, px * 3 * sizeof(float)));
      im->own_normals = true;
    }
  }
  for (uint64_t first = 0; first < n; first += NORMALS_MAX_BATCH) {
    const uint32_t count = (uint32_t)std::min<uint64_t>(NORMALS_MAX_BATCH, n - first);
    NormalsBatch b{};
    for (uint32_t k = 0; k < count; ++k) {
      a3d_device_image* im = images[first + k];
      b.points[k] = im->points, b.mask[k] = im->mask, b.normals[k] = im->normals;
    }
    A3D_TRY(launch_compute_normals_batch(ctx, b, count, images[0]->width, images[0]->height));
  }
  for (uint64_t i = 0; i < n; ++i) images[i]->has_normals = true;
  fence_self_work(ctx, images, n);  // enqueue-only: one fence behind the launches (see a3d_range_image_compute_normals)
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
