// TEST INFRASTRUCTURE — CPU oracle, part 3: frame preparation on or next to the hot path:
// RangeImage::compute_normals, BilateralFilter<u16>, back-projection, luma, pyramid resize.
// PARITY PINNING: src/range_image/structure.rs:479-485 (270 213 valid points in sample1 frame 0) and
// :453-476 (unit normal at [44,42]) are replayed in tests/test_oracle_kat.py; the bilateral KATs of
// src/bilateral/grid.rs:183-194 depend on the reference's JPEG decoder and only their grid-dimension
// formula is replayed.  orc_rgb_pyr_down restates image-0.24.7's blur: PARITY UNPINNED.
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

#include "a3d_oracle.h"
#include "oracle_math.hpp"

using namespace orc;

namespace {

inline V3 load3(const float* base, uint64_t idx) {
  return {base[3 * idx], base[3 * idx + 1], base[3 * idx + 2]};
}
inline void store3(float* base, uint64_t idx, V3 v) {
  base[3 * idx] = v.x, base[3 * idx + 1] = v.y, base[3 * idx + 2] = v.z;
}

// get_point(...).unwrap_or_else(zeros) (src/range_image/structure.rs:175-181, :208-213)
inline V3 point_or_zero(const float* pts, const uint8_t* mask, uint64_t w, uint64_t h, uint64_t row,
                        uint64_t col) {
  if (col < w && row < h && mask[row * w + col] == 1) return load3(pts, row * w + col);
  return {0, 0, 0};
}

struct Grid {
  uint64_t gh, gw, gd;
  std::vector<double> data;  // [gh][gw][gd][2]
  uint16_t color_min;
};

// BilateralGrid::from_image (src/bilateral/grid.rs:32-88)
Grid grid_from_image(const uint16_t* image, uint64_t w, uint64_t h, double sigma_space,
                     double sigma_color) {
  const uint64_t space_pad = 2, color_pad = 2;
  Grid g;
  g.gh = f64_as_usize((double)(h - 1) / sigma_space) + 1 + 2 * space_pad;
  g.gw = f64_as_usize((double)(w - 1) / sigma_space) + 1 + 2 * space_pad;
  uint16_t mi = UINT16_MAX, ma = 0;
  for (uint64_t i = 0; i < w * h; ++i) {  // min/max over ALL pixels, zeros included (:41-49)
    mi = std::min(mi, image[i]);
    ma = std::max(ma, image[i]);
  }
  g.color_min = mi;
  g.gd = f64_as_usize((double)(uint16_t)(ma - mi) / sigma_color) + 1 + 2 * color_pad;
  const double inv_ss = 1.0 / sigma_space, inv_sc = 1.0 / sigma_color;
  g.data.assign(g.gh * g.gw * g.gd * 2, 0.0);
  for (uint64_t row = 0; row < h; ++row) {
    uint64_t grow = f64_as_usize((double)row * inv_ss + 0.5) + space_pad;
    for (uint64_t col = 0; col < w; ++col) {
      uint64_t gcol = f64_as_usize((double)col * inv_ss + 0.5) + space_pad;
      uint16_t color = image[row * w + col];
      if (color <= 0) continue;  // zero depth is not splatted (:67)
      uint64_t ch = f64_as_usize((double)(uint16_t)(color - mi) * inv_sc + 0.5) + color_pad;
      uint64_t cell = ((grow * g.gw + gcol) * g.gd + ch) * 2;
      g.data[cell] += (double)color;
      g.data[cell + 1] += 1.0;
    }
  }
  return g;
}

// BilateralFilter::convolution (src/bilateral/edge_aware_filter.rs:57-115), literally: flat buffers,
// ping-pong by pointer swap, the channel loop starting at an un-offset pointer.
void convolution(Grid& g) {
  std::vector<double> buffer(g.data.size(), 0.0);
  double* data_ptr = g.data.data();
  double* buffer_ptr = buffer.data();
  const int64_t channel_stride = 2;
  const int64_t col_stride = (int64_t)g.gd * 2;
  const int64_t row_stride = (int64_t)g.gw * col_stride;
  const int64_t offsets[3] = {row_stride, col_stride, channel_stride};
  for (int axis = 0; axis < 3; ++axis) {
    const int64_t plane_offset = offsets[axis];
    for (int rep = 0; rep < 2; ++rep) {
      std::swap(data_ptr, buffer_ptr);
      for (uint64_t row = 1; row + 1 < g.gh; ++row) {
        for (uint64_t col = 1; col + 1 < g.gw; ++col) {
          const double* b = buffer_ptr + (int64_t)row * row_stride + (int64_t)col * col_stride;
          double* d = data_ptr + (int64_t)row * row_stride + (int64_t)col * col_stride;
          for (uint64_t ch = 1; ch < g.gd; ++ch) {
            double pv = b[-plane_offset], pw = b[-plane_offset + 1];
            double cv = b[0], cw = b[1];
            double nv = b[plane_offset], nw = b[plane_offset + 1];
            d[0] = (pv + nv + 2.0 * cv) * 0.25;
            d[1] = (pw + nw + 2.0 * cw) * 0.25;
            b += channel_stride;
            d += channel_stride;
          }
        }
      }
    }
  }
  // six passes: the last write went to the buffer that started as grid.data
}

// BilateralGrid::normalize (src/bilateral/grid.rs:90-104)
void normalize(Grid& g) {
  for (uint64_t c = 0; c < g.gh * g.gw * g.gd; ++c) {
    double count = g.data[2 * c + 1];
    if (count > 0.0) {
      g.data[2 * c] /= count;
      g.data[2 * c + 1] = 1.0;
    }
  }
}

inline uint64_t clampu(uint64_t v, uint64_t lo, uint64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// BilateralGrid::trilinear (src/bilateral/grid.rs:132-162)
double trilinear(const Grid& g, double row, double col, double channel) {
  uint64_t z = clampu(f64_as_usize(channel), 0, g.gd - 1);
  uint64_t zz = clampu(f64_as_usize(channel + 1.0), 0, g.gd - 1);
  double za = channel - (double)z;
  uint64_t y = clampu(f64_as_usize(row), 0, g.gh - 1);
  uint64_t yy = clampu(f64_as_usize(row + 1.0), 0, g.gh - 1);
  double ya = row - (double)y;
  uint64_t x = clampu(f64_as_usize(col), 0, g.gw - 1);
  uint64_t xx = clampu(f64_as_usize(col + 1.0), 0, g.gw - 1);
  double xa = col - (double)x;
  auto at = [&](uint64_t r, uint64_t c, uint64_t k) { return g.data[((r * g.gw + c) * g.gd + k) * 2]; };
  double value = (1.0 - ya) * (1.0 - xa) * (1.0 - za) * at(y, x, z) +
                 (1.0 - ya) * xa * (1.0 - za) * at(y, xx, z) +
                 ya * (1.0 - xa) * (1.0 - za) * at(yy, x, z) +
                 ya * xa * (1.0 - za) * at(yy, xx, z) +
                 (1.0 - ya) * (1.0 - xa) * za * at(y, x, zz) +
                 (1.0 - ya) * xa * za * at(y, xx, zz) +
                 ya * (1.0 - xa) * za * at(yy, x, zz) +
                 ya * xa * za * at(yy, xx, zz);
  return value;
}

// BilateralGrid::slice (src/bilateral/grid.rs:106-130)
a3d_status slice(const Grid& g, const uint16_t* image, uint64_t w, uint64_t h, double sigma_space,
                 double sigma_color, uint16_t* out) {
  const double inv_ss = 1.0 / sigma_space, inv_sc = 1.0 / sigma_color;
  a3d_status st = A3D_OK;
  for (uint64_t row = 0; row < h; ++row)
    for (uint64_t col = 0; col < w; ++col) {
      uint16_t color = image[row * w + col];
      double t = trilinear(g, (double)row * inv_ss + 2.0, (double)col * inv_ss + 2.0,
                           (double)(uint16_t)(color - g.color_min) * inv_sc + 2.0);
      // num::cast::<f64,u16>: Some(trunc) iff -1 < t < 65536, else None -> unwrap panics
      if (t > -1.0 && t < 65536.0) {
        out[row * w + col] = (uint16_t)t;
      } else {
        out[row * w + col] = 0;
        st = A3D_CAST_OVERFLOW;
      }
    }
  return st;
}

// get_neighborhood_mean_point (src/range_image/resize.rs:4-40)
bool neighborhood_mean_point(uint64_t src_v, uint64_t src_u, const uint8_t* mask, const float* pts,
                             uint64_t sw, uint64_t sh, V3* out) {
  V3 local[4];
  int n = 0;
  for (uint64_t i = 0; i < 2; ++i)
    for (uint64_t j = 0; j < 2; ++j) {
      uint64_t r = src_v + i, c = src_u + j;
      if (r >= sh || c >= sw) continue;  // the reference would panic (odd sizes); not reached for even dims
      if (mask[r * sw + c] == 1) local[n++] = load3(pts, r * sw + c);
    }
  if (n == 0) return false;
  V3 sum{0, 0, 0};
  for (int k = 0; k < n; ++k) sum = sum + local[k];
  V3 mean = sum / (float)n;
  float min_dist = std::numeric_limits<float>::max();
  V3 nearest{0, 0, 0};
  for (int k = 0; k < n; ++k) {
    float d = norm_squared(local[k] - mean);
    if (d < min_dist) {
      min_dist = d;
      nearest = local[k];
    }
  }
  *out = nearest;
  return true;
}

}  // namespace

extern "C" {

// RangeImage::compute_normals (src/range_image/structure.rs:184-262)
// One 1024-pixel-aligned range of the pixel loop: the reference runs it as rayon chunks of 1024 (structure.rs:193-201).
static void compute_normals_range(const float* points, const uint8_t* mask, uint64_t w, uint64_t h, float* out,
                                  uint64_t begin, uint64_t end) {
  const float thr_sq = 2.0f * 2.0f;
  std::fill(out + 3 * begin, out + 3 * end, 0.0f);
  for (uint64_t idx = begin; idx < end; ++idx) {
    uint64_t row = idx / w, col = idx % w;
    V3 center = load3(points, idx);  // centre mask is NOT checked (:207)
    V3 left = point_or_zero(points, mask, w, h, row, i32_as_usize((int32_t)col - 1));
    V3 right = point_or_zero(points, mask, w, h, row, col + 1);
    float ld = norm_squared(left - center), rd = norm_squared(right - center);
    float lr_ratio = ld / rd;
    V3 left_to_right;
    if (lr_ratio < thr_sq && lr_ratio > 1.0f / thr_sq)
      left_to_right = right - left;
    else if (ld < rd)
      left_to_right = center - left;
    else
      left_to_right = right - center;
    V3 bottom = point_or_zero(points, mask, w, h, row + 1, col);
    V3 top = point_or_zero(points, mask, w, h, i32_as_usize((int32_t)row - 1), col);
    float bd = norm_squared(bottom - center), td = norm_squared(top - center);
    float bt_ratio = bd / td;
    V3 bottom_to_top;
    if (bt_ratio < thr_sq && bt_ratio > 1.0f / thr_sq)
      bottom_to_top = top - bottom;
    else if (bd < td)
      bottom_to_top = center - bottom;
    else
      bottom_to_top = top - center;
    V3 normal = cross(left_to_right, bottom_to_top);
    float mag = std::sqrt(norm_squared(normal));
    if (mag > 1e-6f) store3(out, idx, normal / mag);
  }
}

void orc_compute_normals(const float* points, const uint8_t* mask, uint64_t w, uint64_t h, float* out) {
  compute_normals_range(points, mask, w, h, out, 0, w * h);
}

// The same with the reference's chunking spread over `threads` threads (pixels are independent: same result).
void orc_compute_normals_mt(const float* points, const uint8_t* mask, uint64_t w, uint64_t h, int32_t threads,
                            float* out) {
  const uint64_t n = w * h, chunks = (n + 1023) / 1024;
  if (threads <= 1) return compute_normals_range(points, mask, w, h, out, 0, n);
  std::vector<std::thread> pool;
  const uint64_t per = (chunks + threads - 1) / threads;
  for (int t = 0; t < threads; ++t) {
    const uint64_t b = std::min(n, (uint64_t)t * per * 1024), e = std::min(n, b + per * 1024);
    if (b < e) pool.emplace_back(compute_normals_range, points, mask, w, h, out, b, e);
  }
  for (auto& th : pool) th.join();
}

a3d_status orc_bilateral_filter_u16(const uint16_t* image, uint64_t w, uint64_t h, double sigma_space,
                                    double sigma_color, uint16_t* out, uint64_t dims[3]) {
  if (!image || !out || w == 0 || h == 0) return A3D_INVALID_PARAMETER;
  Grid g = grid_from_image(image, w, h, sigma_space, sigma_color);
  if (dims) dims[0] = g.gh, dims[1] = g.gw, dims[2] = g.gd;
  convolution(g);
  normalize(g);
  return slice(g, image, w, h, sigma_space, sigma_color, out);
}

a3d_status orc_bilateral_grid_slice_u16(const uint16_t* image, uint64_t w, uint64_t h,
                                        double sigma_space, double sigma_color, uint16_t* out,
                                        uint64_t dims[3]) {
  if (!image || !out || w == 0 || h == 0) return A3D_INVALID_PARAMETER;
  Grid g = grid_from_image(image, w, h, sigma_space, sigma_color);
  if (dims) dims[0] = g.gh, dims[1] = g.gw, dims[2] = g.gd;
  normalize(g);
  return slice(g, image, w, h, sigma_space, sigma_color, out);
}

// RangeImage::from_rgbd_image (src/range_image/structure.rs:56-95) + CameraIntrinsics::backproject
uint64_t orc_backproject_depth(const uint16_t* depth, uint64_t w, uint64_t h, double fx, double fy,
                               double cx, double cy, double depth_scale, float* pts, uint8_t* mask) {
  const float scale = (float)depth_scale;
  uint64_t valid = 0;
  std::fill(pts, pts + 3 * w * h, 0.0f);
  std::fill(mask, mask + w * h, (uint8_t)0);
  for (uint64_t y = 0; y < h; ++y)
    for (uint64_t x = 0; x < w; ++x) {
      uint16_t d = depth[y * w + x];
      if (d > 0) {
        float z = (float)d * scale;
        V3 p{((float)x - (float)cx) * z / (float)fx, ((float)y - (float)cy) * z / (float)fy, z};
        store3(pts, y * w + x, p);
        mask[y * w + x] = 1;
        valid++;
      }
    }
  return valid;
}

// rgb_to_luma_u8 (src/image/luma.rs:81-83): saturating, truncating cast
void orc_rgb_to_luma_u8(const uint8_t* rgb, uint64_t n, uint8_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    float l = (float)rgb[3 * i] * 0.3f + (float)rgb[3 * i + 1] * 0.59f + (float)rgb[3 * i + 2] * 0.11f;
    out[i] = l >= 255.0f ? 255 : (l <= 0.0f ? 0 : (uint8_t)l);
  }
}

// resize_range_points (src/range_image/resize.rs:42-74)
void orc_resize_range_points(const float* sp, const uint8_t* sm, uint64_t sw, uint64_t sh, uint64_t dw,
                             uint64_t dh, float* dp, uint8_t* dm) {
  std::fill(dp, dp + 3 * dw * dh, 0.0f);
  std::fill(dm, dm + dw * dh, (uint8_t)0);
  const float hr = (float)sh / (float)dh, wr = (float)sw / (float)dw;
  for (uint64_t v = 0; v < dh; ++v) {
    uint64_t sv = f32_as_usize((float)v * hr);
    for (uint64_t u = 0; u < dw; ++u) {
      uint64_t su = f32_as_usize((float)u * wr);
      V3 p;
      if (!neighborhood_mean_point(sv, su, sm, sp, sw, sh, &p)) continue;
      dm[v * dw + u] = 1;
      store3(dp, v * dw + u, p);
    }
  }
}

// resize_range_normals (src/range_image/resize.rs:76-104): same pick on the normals, source mask
void orc_resize_range_normals(const float* sn, const uint8_t* sm, uint64_t sw, uint64_t sh, uint64_t dw,
                              uint64_t dh, float* dn) {
  std::fill(dn, dn + 3 * dw * dh, 0.0f);
  const float hr = (float)sh / (float)dh, wr = (float)sw / (float)dw;
  for (uint64_t v = 0; v < dh; ++v) {
    uint64_t sv = f32_as_usize((float)v * hr);
    for (uint64_t u = 0; u < dw; ++u) {
      uint64_t su = f32_as_usize((float)u * wr);
      V3 p;
      if (!neighborhood_mean_point(sv, su, sm, sn, sw, sh, &p)) continue;
      store3(dn, v * dw + u, p);
    }
  }
}

// py_scale_down2 (src/range_image/structure.rs:38-47).  image 0.24.7 imageops::blur(sigma) =
// vertical_sample then horizontal_sample with a Gaussian kernel of support 2*sigma, f32 intermediate,
// weights renormalised over the clamped tap range, result clamped and rounded to nearest.
// PARITY UNPINNED: the crate is not vendored and no reference test pins its values.
void orc_rgb_pyr_down(const uint8_t* rgb, uint64_t w, uint64_t h, float sigma, uint8_t* out) {
  if (sigma <= 0.0f) sigma = 1.0f;
  const float support = 2.0f * sigma;
  auto gaussian = [&](float x) {
    return 1.0f / (std::sqrt(2.0f * 3.14159265358979323846f) * sigma) *
           std::exp(-(x * x) / (2.0f * sigma * sigma));
  };
  std::vector<float> tmp(w * h * 3);
  // vertical pass
  for (uint64_t oy = 0; oy < h; ++oy) {
    float in = ((float)oy + 0.5f);
    int64_t left = (int64_t)std::floor(in - support);
    left = std::min<int64_t>(std::max<int64_t>(left, 0), (int64_t)h - 1);
    int64_t right = (int64_t)std::ceil(in + support);
    right = std::min<int64_t>(std::max<int64_t>(right, left + 1), (int64_t)h);
    float c = in - 0.5f;
    std::vector<float> ws;
    float sum = 0.0f;
    for (int64_t i = left; i < right; ++i) {
      float wgt = gaussian((float)i - c);
      ws.push_back(wgt);
      sum += wgt;
    }
    for (auto& x : ws) x /= sum;
    for (uint64_t x = 0; x < w; ++x)
      for (int ch = 0; ch < 3; ++ch) {
        float t = 0.0f;
        for (int64_t i = left; i < right; ++i) t += (float)rgb[((uint64_t)i * w + x) * 3 + ch] * ws[i - left];
        tmp[(oy * w + x) * 3 + ch] = t;
      }
  }
  // horizontal pass + 2x subsample at even rows/cols
  const uint64_t dw = w / 2, dh = h / 2;
  for (uint64_t dy = 0; dy < dh; ++dy)
    for (uint64_t dx = 0; dx < dw; ++dx) {
      uint64_t ox = dx * 2, oy = dy * 2;
      float in = ((float)ox + 0.5f);
      int64_t left = (int64_t)std::floor(in - support);
      left = std::min<int64_t>(std::max<int64_t>(left, 0), (int64_t)w - 1);
      int64_t right = (int64_t)std::ceil(in + support);
      right = std::min<int64_t>(std::max<int64_t>(right, left + 1), (int64_t)w);
      float c = in - 0.5f;
      float wsum = 0.0f, wgt[16];
      int nt = 0;
      for (int64_t i = left; i < right && nt < 16; ++i) {
        wgt[nt] = gaussian((float)i - c);
        wsum += wgt[nt++];
      }
      for (int ch = 0; ch < 3; ++ch) {
        float t = 0.0f;
        for (int k = 0; k < nt; ++k) t += tmp[(oy * w + (uint64_t)(left + k)) * 3 + ch] * (wgt[k] / wsum);
        t = std::min(std::max(t, 0.0f), 255.0f);
        out[(dy * dw + dx) * 3 + ch] = (uint8_t)std::lround(t);
      }
    }
}

}  // extern "C"
