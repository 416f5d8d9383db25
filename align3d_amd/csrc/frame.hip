// RangeImageBuilder::build (src/range_image/builder.rs:74-91) entirely on the device: a frame enters
// HBM as u16 depth + u8 RGB (1.5 MB for 640x480 instead of ~12 MB of f32 pyramid levels) and leaves as a
// resident pyramid:
//   bilateral filter                 src/bilateral/edge_aware_filter.rs:126-135   (bilateral.hip)
//   RangeImage::from_rgbd_image      src/range_image/structure.rs:56-95
//   RangeImage::compute_normals      src/range_image/structure.rs:184-262        (image.hip)
//   RangeImage::pyr_scale_down       src/range_image/structure.rs:309-340, src/range_image/resize.rs:4-104
//   compute_intensity / _map         src/range_image/structure.rs:266-297, src/image/luma.rs:81-83,
//                                    src/intensity_map.rs:37-92
// The RGB blur of the pyramid (image 0.24.7 imageops::blur) is restated from its published algorithm
// like the oracle's: PARITY UNPINNED (no reference test pins its values).
#include <cmath>
#include <memory>

#include "common.hpp"

using namespace a3d;

namespace {

// CameraIntrinsics::backproject (src/camera.rs:101-107) over a depth image; mask = depth > 0.
__global__ void backproject_kernel(const uint16_t* __restrict__ depth, uint32_t w, uint32_t h, float fx, float fy,
                                   float cx, float cy, float scale, float* __restrict__ points,
                                   uint8_t* __restrict__ mask) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const uint32_t y = i / w, x = i % w;
  const uint16_t d = depth[i];
  float px = 0.f, py = 0.f, pz = 0.f;
  if (d > 0) {
    pz = (float)d * scale;
    px = ((float)x - cx) * pz / fx;
    py = ((float)y - cy) * pz / fy;
  }
  points[3 * i] = px, points[3 * i + 1] = py, points[3 * i + 2] = pz;
  mask[i] = d > 0 ? 1 : 0;
}

// rgb_to_luma_u8 (src/image/luma.rs:81-83): (r*0.3 + g*0.59 + b*0.11) as u8 (saturating truncation)
__global__ void luma_kernel(const uint8_t* __restrict__ rgb, uint32_t n, uint8_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float l = (float)rgb[3 * i] * 0.3f + (float)rgb[3 * i + 1] * 0.59f + (float)rgb[3 * i + 2] * 0.11f;
  out[i] = l >= 255.0f ? 255 : (l <= 0.0f ? 0 : (uint8_t)l);
}

// IntensityMap::from_luma_image (src/intensity_map.rs:37-92), one thread per cell of the (h+2) x (w+2) map,
// including the incomplete border: rows h, h+1 copy row h-1 for cols < w-1; cols w, w+1 copy col w-1 for
// rows < h-1; (h, w) and (h+1, w+1) take the last pixel; the other border cells stay 0.
__global__ void intensity_map_kernel(const uint8_t* __restrict__ luma, uint32_t w, uint32_t h, float* __restrict__ map) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t mw = w + 2, mh = h + 2;
  if (i >= mw * mh) return;
  const uint32_t r = i / mw, c = i % mw;
  float v = 0.0f;
  if (r < h && c < w)
    v = (float)luma[r * w + c] / 255.0f;
  else if (r >= h && c + 1 < w)
    v = (float)luma[(h - 1) * w + c] / 255.0f;
  else if (c >= w && r + 1 < h)
    v = (float)luma[r * w + (w - 1)] / 255.0f;
  else if ((r == h && c == w) || (r == h + 1 && c == w + 1))
    v = (float)luma[(h - 1) * w + (w - 1)] / 255.0f;
  map[i] = v;
}

// get_neighborhood_mean_point over every 2x2 block (src/range_image/resize.rs:4-40): among the entries
// whose SOURCE mask is 1, the one nearest to their mean (strict <, first wins).  Used for points
// (writes the destination mask) and for normals (mask output null).
// blockIdx.y = 0 picks the points (and writes the destination mask), blockIdx.y = 1 the normals (if any): one
// launch per level.  The four candidates stay in registers (no dynamically indexed private array).
__global__ void __launch_bounds__(256)
    resize_pick_kernel(const float* __restrict__ src_points, const float* __restrict__ src_normals,
                       const uint8_t* __restrict__ src_mask, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh,
                       float* __restrict__ dst_points, float* __restrict__ dst_normals,
                       uint8_t* __restrict__ dst_mask) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dw * dh) return;
  const bool normals = blockIdx.y == 1;
  const float* __restrict__ src = normals ? src_normals : src_points;
  float* __restrict__ dst = normals ? dst_normals : dst_points;
  const uint32_t dv = i / dw, du = i % dw;
  const float hr = (float)sh / (float)dh, wr = (float)sw / (float)dw;
  const uint32_t sv = (uint32_t)((float)dv * hr), su = (uint32_t)((float)du * wr);
  V3 cand[4];
  bool ok[4];
  int n = 0;
#pragma unroll
  for (uint32_t a = 0; a < 2; ++a)
#pragma unroll
    for (uint32_t b = 0; b < 2; ++b) {
      const uint32_t r = sv + a, c = su + b, q = a * 2 + b;
      const bool in = r < sh && c < sw;
      const uint32_t k = in ? r * sw + c : 0u;
      ok[q] = in && src_mask[k] == 1;
      cand[q] = V3{src[3 * k], src[3 * k + 1], src[3 * k + 2]};
      n += ok[q] ? 1 : 0;
    }
  V3 nearest{0.f, 0.f, 0.f};
  if (n > 0) {
    V3 sum{0.f, 0.f, 0.f};  // valid entries in block order, as the reference's `local` list
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ok[q]) sum = sum + cand[q];
    const V3 mean = sum / (float)n;
    float min_dist = 3.402823466e+38f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d = norm_squared(cand[q] - mean);
      if (ok[q] && d < min_dist) {  // strict <: the first minimum wins
        min_dist = d;
        nearest = cand[q];
      }
    }
  }
  dst[3 * i] = nearest.x, dst[3 * i + 1] = nearest.y, dst[3 * i + 2] = nearest.z;
  if (!normals) dst_mask[i] = n > 0 ? 1 : 0;
}

// One tap table entry per output row / column: first tap, tap count, normalised weights.
constexpr int MAX_TAPS = 14;  // ceil(in + 2 sigma) - floor(in - 2 sigma) <= 14  <=>  sigma <= 3
struct TapRow {
  int32_t left, count;
  float w[MAX_TAPS];
};

// imageops::blur + 2x subsample fused: the pyramid keeps only the even rows and columns of the blurred image,
// so the vertical pass is evaluated at even rows only and never leaves the chip.  One block = one output row x
// BLUR_TILE output columns: the vertical sums (u8 -> f32, the crate's f32 intermediate) of the source columns
// this tile's horizontal taps touch go to LDS, then the horizontal pass, clamp and round-half-away (u8) read
// them back.  Per output the additions run in tap order from 0.0f in both passes, as in the crate.
constexpr uint32_t BLUR_TILE = 64;
constexpr uint32_t BLUR_SPAN = 3 * (2 * BLUR_TILE + MAX_TAPS + 2);  // floats of vertical results a tile can need
__global__ void __launch_bounds__(256)
    blur_halve_kernel(const uint8_t* __restrict__ rgb, uint32_t w, uint32_t dw, const TapRow* __restrict__ taps_v,
                      const TapRow* __restrict__ taps_h, uint8_t* __restrict__ out) {
  __shared__ float s_v[BLUR_SPAN];
  const uint32_t dy = blockIdx.y, dx0 = blockIdx.x * BLUR_TILE, dx1 = min(dx0 + BLUR_TILE, dw) - 1;
  const TapRow* tv = taps_v + dy;  // row 2*dy of the source (table built with stride 2): block-uniform
  const int32_t vleft = tv->left, vcount = tv->count;
  const int32_t cmin = taps_h[dx0].left, cmax = taps_h[dx1].left + taps_h[dx1].count;  // source columns [cmin, cmax)
  const uint32_t span = (uint32_t)(cmax - cmin) * 3u;
  for (uint32_t e = threadIdx.x; e < span; e += 256) {
    const uint8_t* col = rgb + (size_t)vleft * w * 3 + (size_t)cmin * 3 + e;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < MAX_TAPS; ++k)
      if (k < vcount) acc += (float)col[(size_t)k * w * 3] * tv->w[k];
    s_v[e] = acc;
  }
  __syncthreads();
  const uint32_t o = threadIdx.x;  // over BLUR_TILE * 3 outputs
  if (o >= BLUR_TILE * 3) return;
  const uint32_t dx = dx0 + o / 3, ch = o % 3;
  if (dx >= dw) return;
  const TapRow* th = taps_h + dx;
  const int32_t hleft = th->left - cmin, hcount = th->count;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < MAX_TAPS; ++k)
    if (k < hcount) acc += s_v[(uint32_t)(hleft + k) * 3u + ch] * th->w[k];
  acc = fminf(fmaxf(acc, 0.0f), 255.0f);
  out[((size_t)dy * dw + dx) * 3 + ch] = (uint8_t)roundf(acc);
}

// Tap tables of image::imageops::blur's sampling filter (support 2 sigma, weights renormalised over the
// clamped range), computed on the host in f32 exactly as the oracle computes them.
std::vector<TapRow> make_taps(uint32_t size, float sigma, uint32_t stride, uint32_t count) {
  const float support = 2.0f * sigma;
  std::vector<TapRow> rows(count);
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t o = k * stride;
    const float in = (float)o + 0.5f;
    int64_t left = (int64_t)std::floor(in - support);
    left = std::min<int64_t>(std::max<int64_t>(left, 0), (int64_t)size - 1);
    int64_t right = (int64_t)std::ceil(in + support);
    right = std::min<int64_t>(std::max<int64_t>(right, left + 1), (int64_t)size);
    const float c = in - 0.5f;
    TapRow r{};
    r.left = (int32_t)left;
    r.count = (int32_t)std::min<int64_t>(right - left, MAX_TAPS);  // callers reject sigma > 3 (more taps)
    float sum = 0.0f, wv[MAX_TAPS];
    for (int i = 0; i < r.count; ++i) {
      const float x = (float)(left + i) - c;
      wv[i] = 1.0f / (std::sqrt(2.0f * 3.14159265358979323846f) * sigma) * std::exp(-(x * x) / (2.0f * sigma * sigma));
      sum += wv[i];
    }
    for (int i = 0; i < r.count; ++i) r.w[i] = wv[i] / sum;
    rows[k] = r;
  }
  return rows;
}

// Bump allocator over one region (256-byte aligned pieces): the pyramid's arrays share one arena, the
// temporaries share the context's scratch region, so a frame costs one hipMalloc instead of ~40.
struct Carver {
  char* base = nullptr;
  size_t used = 0, capacity = 0;
  template <typename T>
  a3d_status take(T** p, size_t count) {
    const size_t bytes = ((std::max<size_t>(1, count) * sizeof(T) + 255) / 256) * 256;
    A3D_REQUIRE(used + bytes <= capacity, A3D_INVALID_PARAMETER, "internal: arena too small");
    *p = (T*)(base + used);
    used += bytes;
    return A3D_OK;
  }
};
inline size_t padded(size_t bytes) { return ((std::max<size_t>(1, bytes) + 255) / 256) * 256; }

inline dim3 grid_for(size_t n) { return dim3((uint32_t)((n + 255) / 256)); }

// compute_intensity + compute_intensity_map on a resident level that has colours
a3d_status add_intensity(a3d_device_image* im, Carver& arena) {
  const uint32_t w = im->width, h = im->height, n = w * h;
  hipStream_t s = im->ctx->stream;
  A3D_TRY(arena.take(&im->intensities, n));
  A3D_TRY(arena.take(&im->imap, (size_t)(w + 2) * (h + 2)));
  hipLaunchKernelGGL(luma_kernel, grid_for(n), dim3(256), 0, s, im->colors, n, im->intensities);
  hipLaunchKernelGGL(intensity_map_kernel, grid_for((size_t)(w + 2) * (h + 2)), dim3(256), 0, s, im->intensities, w, h,
                     im->imap);
  A3D_HIP_TRY(hipGetLastError());
  im->has_intensities = im->has_imap = true;
  return A3D_OK;
}

// RangeImage::pyr_scale_down(sigma) (structure.rs:309-340)
a3d_status pyr_scale_down(const a3d_device_image* src, float sigma, a3d_device_image* dst, Carver& arena,
                          Carver& scratch) {
  a3d_context* ctx = src->ctx;
  hipStream_t s = ctx->stream;
  const uint32_t sw = src->width, sh = src->height, dw = sw / 2, dh = sh / 2, dn = dw * dh;
  dst->ctx = ctx;
  dst->width = dw, dst->height = dh;
  dst->fx64 = src->fx64 * 0.5, dst->fy64 = src->fy64 * 0.5, dst->cx64 = src->cx64 * 0.5, dst->cy64 = src->cy64 * 0.5;
  dst->fx = (float)dst->fx64, dst->fy = (float)dst->fy64, dst->cx = (float)dst->cx64, dst->cy = (float)dst->cy64;
  A3D_TRY(arena.take(&dst->points, (size_t)dn * 3));
  A3D_TRY(arena.take(&dst->mask, dn));
  if (src->has_normals) {
    A3D_TRY(arena.take(&dst->normals, (size_t)dn * 3));
    dst->has_normals = true;
  }
  hipLaunchKernelGGL(resize_pick_kernel, dim3((dn + 255) / 256, src->has_normals ? 2 : 1), dim3(256), 0, s, src->points,
                     src->normals, src->mask, sw, sh, dw, dh, dst->points, dst->normals, dst->mask);
  if (src->colors) {
    if (sigma <= 0.0f) sigma = 1.0f;
    A3D_REQUIRE(sigma <= 3.0f, A3D_INVALID_PARAMETER, "blur_sigma above 3 is not supported by the device builder");
    // the tap tables depend on (size, sigma) only: computed and uploaded once per context, then reused
    auto taps_for = [&](uint32_t size, uint32_t count, TapRow** out) -> a3d_status {
      uint32_t key[4] = {0x54415053u /* 'TAPS' */, size, count, 0};
      memcpy(&key[3], &sigma, 4);
      for (const auto& t : ctx->tables)
        if (!memcmp(t.key, key, sizeof(key))) {
          *out = (TapRow*)t.d;
          return A3D_OK;
        }
      const std::vector<TapRow> rows = make_taps(size, sigma, 2, count);
      return ctx_cached_table(ctx, key, rows.data(), rows.size() * sizeof(TapRow), (void**)out);
    };
    TapRow *d_tv = nullptr, *d_th = nullptr;
    A3D_TRY(taps_for(sh, dh, &d_tv));
    A3D_TRY(taps_for(sw, dw, &d_th));
    A3D_TRY(arena.take(&dst->colors, (size_t)dn * 3));
    hipLaunchKernelGGL(blur_halve_kernel, dim3((dw + BLUR_TILE - 1) / BLUR_TILE, dh), dim3(256), 0, s, src->colors, sw,
                       dw, d_tv, d_th, dst->colors);
  }
  A3D_HIP_TRY(hipGetLastError());
  return A3D_OK;
}

}  // namespace

extern "C" {

// RangeImageBuilder::default() (builder.rs:16-26)
void a3d_builder_params_default(a3d_builder_params* out) {
  out->with_normals = 1;
  out->with_intensity = 1;
  out->use_bilateral = 0;
  a3d_bilateral_default_sigmas(&out->sigma_space, &out->sigma_color);
  out->pyramid_levels = 3;
  out->blur_sigma = 1.0f;
}

a3d_status a3d_range_image_build_pyramid(a3d_context* ctx, const a3d_builder_params* prm, const uint16_t* depth,
                                         const uint8_t* rgb, uint64_t width, uint64_t height, double fx, double fy,
                                         double cx, double cy, double depth_scale, a3d_device_image** out_levels) {
  A3D_REQUIRE(ctx && prm && depth && rgb && out_levels, A3D_INVALID_PARAMETER, "null argument");
  A3D_REQUIRE(width > 0 && height > 0 && width * height < (1ull << 28), A3D_INVALID_PARAMETER, "bad image size");
  // the kernels form texel offsets with 24-bit multiplies
  A3D_REQUIRE(width < (1ull << 23) && height < (1ull << 23), A3D_INVALID_PARAMETER, "image side too long");
  A3D_REQUIRE(prm->pyramid_levels >= 1 && prm->pyramid_levels <= 16, A3D_INVALID_PARAMETER, "bad pyramid_levels");
  A3D_REQUIRE((width >> (prm->pyramid_levels - 1)) >= 2 && (height >> (prm->pyramid_levels - 1)) >= 2,
              A3D_INVALID_PARAMETER, "image too small for this many pyramid levels");
  A3D_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint32_t w = (uint32_t)width, h = (uint32_t)height, n = w * h;
  // ---- sizes: one arena for everything that stays resident, the context's scratch for the temporaries ----
  const uint64_t L = prm->pyramid_levels;
  size_t arena_bytes = 0, scratch_bytes = 2 * padded((size_t)n * 2);
  for (uint64_t l = 0; l < L; ++l) {
    const size_t wl = w >> l, hl = h >> l, nl = wl * hl;
    arena_bytes += padded(nl * 12) + padded(nl) + padded(nl * 3);                 // points, mask, colors
    if (prm->with_normals) arena_bytes += padded(nl * 12);
    if (prm->with_intensity) arena_bytes += padded(nl) + padded((wl + 2) * (hl + 2) * 4);
  }
  DeviceArena* shared = new DeviceArena();
  if (ctx_arena_acquire(ctx, arena_bytes, shared) != A3D_OK) {
    delete shared;
    set_error("a3d_range_image_build_pyramid: hipMalloc(%zu) failed", arena_bytes);
    return A3D_HIP_ERROR;
  }
  Carver arena{(char*)shared->base, 0, arena_bytes};
  std::vector<a3d_device_image*> levels;
  auto new_level = [&]() {
    a3d_device_image* im = new a3d_device_image();
    im->ctx = ctx;
    im->arena = shared;
    ++shared->refs;
    levels.push_back(im);
    return im;
  };
  bool async_bilateral = false;
  size_t retry_scratch_bytes = 0;
  auto build = [&]() -> a3d_status {
    void* scratch_base = nullptr;
    A3D_TRY(ctx_scratch(ctx, 0, scratch_bytes, &scratch_base));
    Carver scratch{(char*)scratch_base, 0, scratch_bytes};
    uint16_t *d_depth = nullptr, *d_filtered = nullptr;
    a3d_device_image* l0 = new_level();
    l0->width = w, l0->height = h;
    l0->fx64 = fx, l0->fy64 = fy, l0->cx64 = cx, l0->cy64 = cy;
    l0->fx = (float)fx, l0->fy = (float)fy, l0->cx = (float)cx, l0->cy = (float)cy;
    A3D_TRY(scratch.take(&d_depth, n));
    A3D_TRY(scratch.take(&d_filtered, n));
    A3D_TRY(arena.take(&l0->colors, (size_t)n * 3));
    A3D_TRY(arena.take(&l0->points, (size_t)n * 3));
    A3D_TRY(arena.take(&l0->mask, n));
    A3D_HIP_TRY(hipMemcpyAsync(d_depth, depth, (size_t)n * 2, hipMemcpyHostToDevice, s));
    A3D_HIP_TRY(hipMemcpyAsync(l0->colors, rgb, (size_t)n * 3, hipMemcpyHostToDevice, s));
    const uint16_t* d_use = d_depth;
    if (prm->use_bilateral) {  // builder.rs:75-77
      // without a host round trip when the context's grid scratch already exists (every frame but the first)
      A3D_TRY(bilateral_filter_device_async(ctx, d_depth, d_filtered, w, h, prm->sigma_space, prm->sigma_color,
                                            ctx->pinned_words, &async_bilateral));
      if (!async_bilateral)
        A3D_TRY(bilateral_filter_device(ctx, d_depth, d_filtered, w, h, prm->sigma_space, prm->sigma_color, nullptr));
      d_use = d_filtered;
    }
    hipLaunchKernelGGL(backproject_kernel, grid_for(n), dim3(256), 0, s, d_use, w, h, l0->fx, l0->fy, l0->cx, l0->cy,
                       (float)depth_scale, l0->points, l0->mask);
    if (prm->with_normals) {  // level 0 only (builder.rs:79-82); coarser levels inherit picked normals
      A3D_TRY(arena.take(&l0->normals, (size_t)n * 3));
      A3D_TRY(compute_normals_device(ctx, l0->points, l0->mask, l0->normals, w, h));
      l0->has_normals = true;
    }
    for (uint64_t l = 1; l < L; ++l) {  // RangeImage::pyramid (structure.rs:342-351)
      a3d_device_image* prev = levels.back();
      a3d_device_image* next = new_level();
      A3D_TRY(pyr_scale_down(prev, prm->blur_sigma, next, arena, scratch));
    }
    if (prm->with_intensity)
      for (a3d_device_image* lv : levels) A3D_TRY(add_intensity(lv, arena));
    A3D_HIP_TRY(hipGetLastError());
    A3D_HIP_TRY(hipStreamSynchronize(s));
    if (async_bilateral) {  // the filter's scalars arrived with the synchronisation above
      A3D_TRY(bilateral_async_status(ctx->pinned_words, &retry_scratch_bytes));
      if (retry_scratch_bytes) return A3D_HIP_ERROR;  // (not an error: handled below by growing the region)
    }
    return A3D_OK;
  };
  const a3d_status st = build();
  if (st != A3D_OK) {
    hipStreamSynchronize(s);
    for (a3d_device_image* lv : levels) a3d_range_image_free(lv);  // the last one releases the arena
    if (levels.empty()) {
      ctx_arena_release(ctx, shared);
      delete shared;
    }
    if (retry_scratch_bytes) {  // this frame's bilateral grid outgrew the scratch region: grow it, build again
      void* unused = nullptr;
      A3D_TRY(ctx_scratch(ctx, 1, retry_scratch_bytes, &unused));
      return a3d_range_image_build_pyramid(ctx, prm, depth, rgb, width, height, fx, fy, cx, cy, depth_scale,
                                           out_levels);
    }
    return st;
  }
  for (size_t l = 0; l < levels.size(); ++l) out_levels[l] = levels[l];
  return A3D_OK;
}

a3d_status a3d_range_image_size(const a3d_device_image* im, uint64_t* out_width, uint64_t* out_height) {
  A3D_REQUIRE(im && out_width && out_height, A3D_INVALID_PARAMETER, "null argument");
  *out_width = im->width, *out_height = im->height;
  return A3D_OK;
}

a3d_status a3d_range_image_download(a3d_device_image* im, float* points, uint8_t* mask, float* normals,
                                    uint8_t* intensities, float* intensity_map, uint8_t* colors, double out_intrinsics[4]) {
  A3D_REQUIRE(im, A3D_INVALID_PARAMETER, "image is null");
  hipStream_t s = im->ctx->stream;
  const size_t n = (size_t)im->width * im->height;
  auto get = [&](void* dst, const void* src, size_t bytes, const char* what) -> a3d_status {
    if (!dst) return A3D_OK;
    A3D_REQUIRE(src, A3D_MISSING_FIELD, what);
    A3D_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    return A3D_OK;
  };
  A3D_TRY(get(points, im->points, n * 12, "image has no points"));
  A3D_TRY(get(mask, im->mask, n, "image has no mask"));
  A3D_TRY(get(normals, im->has_normals ? im->normals : nullptr, n * 12, "image has no normals"));
  A3D_TRY(get(intensities, im->has_intensities ? im->intensities : nullptr, n, "image has no intensities"));
  A3D_TRY(get(intensity_map, im->has_imap ? im->imap : nullptr, (size_t)(im->width + 2) * (im->height + 2) * 4,
              "image has no intensity map"));
  A3D_TRY(get(colors, im->colors, n * 3, "image has no colors"));
  if (out_intrinsics) out_intrinsics[0] = im->fx64, out_intrinsics[1] = im->fy64, out_intrinsics[2] = im->cx64, out_intrinsics[3] = im->cy64;
  A3D_HIP_TRY(hipStreamSynchronize(s));
  return A3D_OK;
}

}  // extern "C"
