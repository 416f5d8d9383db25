// align3d.hpp — C++17 host-side mirror of the reference's API for the ICP hot path, header-only,
// over the C ABI of align3d_hip.h.  Names and argument meaning follow the Rust reference:
//
//   icp::IcpParams / MsIcpParams            src/icp/icp_params.rs:8-134
//   icp::multiscale::MultiscaleAlign        src/icp/multiscale.rs:7-68
//   icp::ImageIcp                           src/icp/image_icp.rs:19-165
//   icp::Icp                                src/icp/pcl_icp.rs:15-108
//   kdtree::R3dTree                         src/kdtree.rs:19-106
//   range_image::RangeImage (borrowed view) src/range_image/structure.rs:20-36
//   range_image::RangeImageBuilder          src/range_image/builder.rs:7-92
//   bilateral::BilateralFilter<u16>         src/bilateral/edge_aware_filter.rs:14-135
//   transform::Transform                    src/transform.rs:18
//
// Error mapping: `Result::Err(A3dError::InvalidParameter)` -> align3d::InvalidParameter;
// the reference's panics (`expect`, `unwrap`) -> align3d::Panic carrying the status and the library's text.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "align3d_hip.h"

namespace align3d {

struct Error : std::runtime_error {
  a3d_status status;
  Error(a3d_status s, const std::string& what) : std::runtime_error(what), status(s) {}
};
struct InvalidParameter : Error {  // A3dError::InvalidParameter (src/error.rs:3-9)
  explicit InvalidParameter(const std::string& what) : Error(A3D_INVALID_PARAMETER, what) {}
};
struct Panic : Error {  // what the reference turns into a panic
  using Error::Error;
};

inline void check(a3d_status s) {
  if (s == A3D_OK) return;
  std::string text = std::string(a3d_status_string(s)) + ": " + a3d_last_error();
  if (s == A3D_INVALID_PARAMETER) throw InvalidParameter(text);
  throw Panic(s, text);
}

/// One GPU + one HIP stream; every object below is created on a Context and must not outlive it.
class Context {
 public:
  /// priority < 0: the device's highest stream priority (a frame-builder context beside an aligning one), > 0: lowest.
  explicit Context(int device_index = 0, int priority = 0) {
    check(a3d_context_create_with_priority(device_index, priority, &ctx_));
  }
  /// An aligning context and the builder context that feeds it, created back to back (a3d_context_create_pair: the
  /// builder's kernel stream then sits beside, not behind, the aligner's pair-group streams on the GPU's compute pipes).
  static std::pair<Context, Context> pair(int device_index = 0) {
    a3d_context *a = nullptr, *b = nullptr;
    check(a3d_context_create_pair(device_index, &a, &b));
    return {Context(a, true), Context(b, true)};
  }
  /// A context owned by someone else (MultiContext::device): used like any other, not destroyed here.
  static Context borrowed(a3d_context* raw) {
    Context c(raw, false);
    return c;
  }
  Context(Context&& o) noexcept : ctx_(o.ctx_), owned_(o.owned_) { o.ctx_ = nullptr; }
  ~Context() {
    if (owned_ && ctx_) a3d_context_destroy(ctx_);
  }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  a3d_context* raw() const { return ctx_; }
  void synchronize() const { check(a3d_context_synchronize(ctx_)); }
  /// a3d_context_set_tiling: 0 = throughput tiling (a pair's cut into blocks follows its batch); n > 0 = pinned tiling
  /// (n blocks per pair and level whatever the batch: a pair's pose is bit-identical alone and in any batch).
  void set_tiling(uint32_t tiles_per_pair) const { check(a3d_context_set_tiling(ctx_, tiles_per_pair)); }

 private:
  Context(a3d_context* raw, bool owned) : ctx_(raw), owned_(owned) {}
  a3d_context* ctx_ = nullptr;
  bool owned_ = true;
};

/// Transform = Isometry3<f32>: translation + unit quaternion (i, j, k, w).
struct Transform {
  std::array<float, 3> translation{0.f, 0.f, 0.f};
  std::array<float, 4> rotation_ijkw{0.f, 0.f, 0.f, 1.f};
  static Transform eye() { return {}; }
  a3d_pose to_c() const {
    a3d_pose p;
    for (int i = 0; i < 3; ++i) p.t[i] = translation[i];
    for (int i = 0; i < 4; ++i) p.q[i] = rotation_ijkw[i];
    return p;
  }
  static Transform from_c(const a3d_pose& p) {
    Transform t;
    for (int i = 0; i < 3; ++i) t.translation[i] = p.t[i];
    for (int i = 0; i < 4; ++i) t.rotation_ijkw[i] = p.q[i];
    return t;
  }
};

/// IcpParams: the C struct is the field-for-field mirror; default() is IcpParams::default().
struct IcpParams : a3d_icp_params {
  IcpParams() { a3d_icp_params_default(this); }
  static IcpParams default_() { return IcpParams(); }
  IcpParams& with_max_iterations(uint64_t v) {
    max_iterations = v;
    return *this;
  }
  IcpParams& with_weight(float v) {
    weight = v;
    return *this;
  }
};

/// MsIcpParams: per-level parameters, index 0 = finest level = last to run.
class MsIcpParams {
 public:
  explicit MsIcpParams(std::vector<IcpParams> pyramid) : pyramid_(std::move(pyramid)) {}
  static MsIcpParams repeat(size_t levels, const IcpParams& p) { return MsIcpParams(std::vector<IcpParams>(levels, p)); }
  static MsIcpParams default_() {  // MsIcpParams::default(): 3 levels, 20/20/30 iterations
    a3d_icp_params raw[3];
    a3d_ms_icp_params_default(raw);
    std::vector<IcpParams> v(3);
    for (int i = 0; i < 3; ++i) static_cast<a3d_icp_params&>(v[i]) = raw[i];
    return MsIcpParams(std::move(v));
  }
  MsIcpParams customize(const std::function<void(size_t, IcpParams&)>& f) && {
    for (size_t i = 0; i < pyramid_.size(); ++i) f(i, pyramid_[i]);
    return std::move(*this);
  }
  size_t len() const { return pyramid_.size(); }
  bool is_empty() const { return pyramid_.empty(); }
  IcpParams& operator[](size_t i) { return pyramid_[i]; }
  const IcpParams& operator[](size_t i) const { return pyramid_[i]; }
  auto begin() const { return pyramid_.begin(); }
  auto end() const { return pyramid_.end(); }
  std::vector<a3d_icp_params> to_c() const { return std::vector<a3d_icp_params>(pyramid_.begin(), pyramid_.end()); }

 private:
  std::vector<IcpParams> pyramid_;
};

/// A RangeImage resident in HBM (uploaded from a borrowed host view in the reference's standard layout).
class RangeImage {
 public:
  RangeImage(const Context& ctx, const a3d_range_image_view& host_view) {
    check(a3d_range_image_upload(ctx.raw(), &host_view, &img_));
  }
  /// Adopts a handle produced by the library (RangeImageBuilder::build).
  explicit RangeImage(a3d_device_image* adopted) : img_(adopted) {}
  uint64_t width() const {
    uint64_t w, h;
    check(a3d_range_image_size(img_, &w, &h));
    return w;
  }
  uint64_t height() const {
    uint64_t w, h;
    check(a3d_range_image_size(img_, &w, &h));
    return h;
  }
  ~RangeImage() { a3d_range_image_free(img_); }
  RangeImage(RangeImage&& o) noexcept : img_(o.img_) { o.img_ = nullptr; }
  RangeImage(const RangeImage&) = delete;
  RangeImage& operator=(const RangeImage&) = delete;
  /// RangeImage::compute_normals
  RangeImage& compute_normals() {
    check(a3d_range_image_compute_normals(img_));
    return *this;
  }
  void download_normals(float* out_hw3) { check(a3d_range_image_download_normals(img_, out_hw3)); }
  const a3d_device_image* raw() const { return img_; }

 private:
  a3d_device_image* img_ = nullptr;
};

/// RangeImage::compute_normals on many resident images of one size in one launch per 64 images
/// (a3d_range_image_compute_normals_batch; enqueue-only).
inline void compute_normals_batch(const std::vector<RangeImage*>& images) {
  std::vector<a3d_device_image*> raw;
  raw.reserve(images.size());
  for (RangeImage* im : images) raw.push_back(const_cast<a3d_device_image*>(im->raw()));
  check(a3d_range_image_compute_normals_batch(raw.data(), raw.size()));
}

/// A whole host pyramid (`&[RangeImage]`) in one call: the levels share one pooled arena (a3d_range_image_upload_pyramid).
inline std::vector<RangeImage> upload_pyramid(const Context& ctx, const std::vector<a3d_range_image_view>& host_views) {
  std::vector<a3d_device_image*> raw(host_views.size(), nullptr);
  if (!host_views.empty()) check(a3d_range_image_upload_pyramid(ctx.raw(), host_views.data(), host_views.size(), raw.data()));
  std::vector<RangeImage> out;
  out.reserve(raw.size());
  for (a3d_device_image* im : raw) out.emplace_back(im);
  return out;
}

inline std::vector<const a3d_device_image*> raw_pointers(const std::vector<RangeImage>& v) {
  std::vector<const a3d_device_image*> out;
  for (const auto& r : v) out.push_back(r.raw());
  return out;
}

/// BilateralFilter::<u16>::{default, new}(sigma_space, sigma_color).filter(&image)
class BilateralFilter {
 public:
  double sigma_space, sigma_color;
  BilateralFilter() { a3d_bilateral_default_sigmas(&sigma_space, &sigma_color); }
  BilateralFilter(double ss, double sc) : sigma_space(ss), sigma_color(sc) {}
  std::vector<uint16_t> filter(const Context& ctx, const uint16_t* image, uint64_t width, uint64_t height) const {
    std::vector<uint16_t> out(width * height);
    check(a3d_bilateral_filter_u16(ctx.raw(), image, width, height, sigma_space, sigma_color, out.data(), nullptr));
    return out;
  }
  /// the same on `n_images` images already resident ([n][height][width] u16, device pointers)
  void filter_device(const Context& ctx, const uint16_t* d_images, uint64_t n_images, uint64_t width, uint64_t height,
                     uint16_t* d_out) const {
    check(a3d_bilateral_filter_u16_device(ctx.raw(), d_images, n_images, width, height, sigma_space, sigma_color, d_out));
  }
};

/// CameraIntrinsics (src/camera.rs:9-22)
struct CameraIntrinsics {
  double fx, fy, cx, cy;
  uint64_t width, height;
};

/// RangeImageBuilder::default().with_*(..).build(frame) -> Vec<RangeImage> (builder.rs:16-91), on the device:
/// the frame crosses PCIe as u16 depth + u8 RGB and every pyramid level stays resident in HBM.
class RangeImageBuilder {
 public:
  explicit RangeImageBuilder(const Context& ctx) : ctx_(ctx) { a3d_builder_params_default(&p_); }
  RangeImageBuilder& with_normals(bool v) {
    p_.with_normals = v;
    return *this;
  }
  RangeImageBuilder& with_intensity(bool v) {
    p_.with_intensity = v;
    return *this;
  }
  RangeImageBuilder& with_bilateral_filter(const BilateralFilter* f) {  // Option<BilateralFilter>
    p_.use_bilateral = f != nullptr;
    if (f) p_.sigma_space = f->sigma_space, p_.sigma_color = f->sigma_color;
    return *this;
  }
  RangeImageBuilder& pyramid_levels(uint64_t n) {
    p_.pyramid_levels = n;
    return *this;
  }
  RangeImageBuilder& blur_sigma(float s) {
    p_.blur_sigma = s;
    return *this;
  }
  /// depth: [height][width] u16; rgb: [height][width][3] u8; depth_scale as RgbdImage::depth_scale.
  std::vector<RangeImage> build(const CameraIntrinsics& k, const uint16_t* depth, const uint8_t* rgb,
                                double depth_scale) const {
    std::vector<a3d_device_image*> raw(p_.pyramid_levels, nullptr);
    check(a3d_range_image_build_pyramid(ctx_.raw(), &p_, depth, rgb, k.width, k.height, k.fx, k.fy, k.cx, k.cy,
                                        depth_scale, raw.data()));
    std::vector<RangeImage> out;
    out.reserve(raw.size());
    for (a3d_device_image* im : raw) out.emplace_back(im);
    return out;
  }
  /// The same for n frames of one stream (same size and camera) in one launch sequence
  /// (a3d_range_image_build_pyramids): returns pyramids[frame][level].
  std::vector<std::vector<RangeImage>> build_many(const CameraIntrinsics& k, const std::vector<const uint16_t*>& depth,
                                                   const std::vector<const uint8_t*>& rgb, double depth_scale) const {
    if (depth.size() != rgb.size()) throw InvalidParameter("A3D_INVALID_PARAMETER: one depth and one rgb image per frame");
    std::vector<std::vector<RangeImage>> out;
    if (depth.empty()) return out;
    std::vector<a3d_device_image*> raw(depth.size() * p_.pyramid_levels, nullptr);
    check(a3d_range_image_build_pyramids(ctx_.raw(), &p_, depth.size(), depth.data(), rgb.data(), k.width, k.height, k.fx,
                                         k.fy, k.cx, k.cy, depth_scale, raw.data()));
    for (size_t f = 0; f < depth.size(); ++f) {
      out.emplace_back();
      for (uint64_t l = 0; l < p_.pyramid_levels; ++l) out.back().emplace_back(raw[f * p_.pyramid_levels + l]);
    }
    return out;
  }

 private:
  const Context& ctx_;
  a3d_builder_params p_;
};

/// ImageIcp::new(params, &target); initial_transform; align(&source)
class ImageIcp {
 public:
  ImageIcp(const Context& ctx, const IcpParams& params, const RangeImage& target)
      : params(params), ctx_(ctx), target_(target) {}
  IcpParams params;
  Transform initial_transform = Transform::eye();
  Transform align(const RangeImage& source) const {
    a3d_pose init = initial_transform.to_c(), out;
    check(a3d_image_icp_align(ctx_.raw(), &params, target_.raw(), source.raw(), &init, &out));
    return Transform::from_c(out);
  }

 private:
  const Context& ctx_;
  const RangeImage& target_;
};

/// MultiscaleAlign::new(params, &target_pyramid) -> Result ; align(&source_pyramid) -> Transform
class MultiscaleAlign {
 public:
  MultiscaleAlign(const Context& ctx, const MsIcpParams& params, const std::vector<RangeImage>& target_pyramid) {
    auto p = params.to_c();
    auto t = raw_pointers(target_pyramid);
    check(a3d_multiscale_new(ctx.raw(), p.data(), p.size(), t.data(), t.size(), &ms_));  // throws InvalidParameter
  }
  ~MultiscaleAlign() { a3d_multiscale_free(ms_); }
  MultiscaleAlign(const MultiscaleAlign&) = delete;
  MultiscaleAlign& operator=(const MultiscaleAlign&) = delete;
  Transform align(const std::vector<RangeImage>& source_pyramid) const {
    auto s = raw_pointers(source_pyramid);
    a3d_pose out;
    check(a3d_multiscale_align(ms_, s.data(), s.size(), &out));
    return Transform::from_c(out);
  }
  /// The reference's literal call: the source pyramid as borrowed HOST arrays (views); uploaded and aligned in one call,
  /// the coarse levels iterating under the upload of the fine ones (a3d_multiscale_align_host).
  Transform align_host(const std::vector<a3d_range_image_view>& source_pyramid) const {
    a3d_pose out;
    check(a3d_multiscale_align_host(ms_, source_pyramid.data(), source_pyramid.size(), &out));
    return Transform::from_c(out);
  }

 private:
  a3d_multiscale* ms_ = nullptr;
};

/// R3dTree::new(&points) ; nearest(&query) -> (index, squared distance)
class R3dTree {
 public:
  R3dTree(const Context& ctx, const float* points_n3, uint64_t n) { check(a3d_kdtree_new(ctx.raw(), points_n3, n, &t_)); }
  /// Over points already resident in HBM (device pointer, [n][3] f32): a3d_kdtree_new_device.
  static R3dTree from_device(const Context& ctx, const void* d_points_n3, uint64_t n) {
    R3dTree t;
    check(a3d_kdtree_new_device(ctx.raw(), d_points_n3, n, &t.t_));
    return t;
  }
  R3dTree(R3dTree&& o) noexcept : t_(o.t_) { o.t_ = nullptr; }
  ~R3dTree() { a3d_kdtree_free(t_); }
  R3dTree(const R3dTree&) = delete;
  R3dTree& operator=(const R3dTree&) = delete;
  std::pair<uint64_t, float> nearest(const std::array<float, 3>& q) const {
    uint64_t idx;
    float d;
    check(a3d_kdtree_nearest(t_, q.data(), 1, &idx, &d));
    return {idx, d};
  }
  void nearest(const float* queries_m3, uint64_t m, uint64_t* out_idx, float* out_sqr) const {
    check(a3d_kdtree_nearest(t_, queries_m3, m, out_idx, out_sqr));
  }

 private:
  R3dTree() = default;
  a3d_kdtree* t_ = nullptr;
};

/// Icp::new(params, &target_cloud) ; align(&source_cloud)
class Icp {
 public:
  Icp(const Context& ctx, const IcpParams& params, const a3d_point_cloud_view& target) {
    check(a3d_pcl_icp_new(ctx.raw(), &params, &target, &icp_));
  }
  ~Icp() { a3d_pcl_icp_free(icp_); }
  Icp(const Icp&) = delete;
  Icp& operator=(const Icp&) = delete;
  Transform align(const a3d_point_cloud_view& source) const {
    a3d_pose out;
    check(a3d_pcl_icp_align(icp_, &source, &out));
    return Transform::from_c(out);
  }
  /// Clouds already resident in HBM: the views hold DEVICE pointers (a3d_pcl_icp_new_device / _align_device).
  static Icp from_device(const Context& ctx, const IcpParams& params, const a3d_point_cloud_view& d_target) {
    Icp icp;
    check(a3d_pcl_icp_new_device(ctx.raw(), &params, &d_target, &icp.icp_));
    return icp;
  }
  Icp(Icp&& o) noexcept : icp_(o.icp_) { o.icp_ = nullptr; }
  Transform align_device(const a3d_point_cloud_view& d_source) const {
    a3d_pose out;
    check(a3d_pcl_icp_align_device(icp_, &d_source, &out));
    return Transform::from_c(out);
  }

 private:
  Icp() = default;
  a3d_pcl_icp* icp_ = nullptr;
};

/// P independent MultiscaleAlign jobs in one launch sequence (the per-GPU shard of a batch of frame pairs):
/// the same parameters for every pair, pyramids[pair][level]; align() returns one Transform per pair and
/// throws Panic if any pair's solve failed (status() then tells which).
class MultiscaleAlignBatch {
 public:
  MultiscaleAlignBatch(const Context& ctx, const MsIcpParams& params,
                       const std::vector<const std::vector<RangeImage>*>& target_pyramids,
                       const std::vector<const std::vector<RangeImage>*>& source_pyramids)
      : n_pairs_(target_pyramids.size()) {
    if (target_pyramids.size() != source_pyramids.size() || target_pyramids.empty())
      throw InvalidParameter("A3D_INVALID_PARAMETER: one target and one source pyramid per pair are required");
    const size_t levels = target_pyramids[0]->size();
    std::vector<const a3d_device_image*> t, s;
    for (size_t p = 0; p < n_pairs_; ++p) {
      if (target_pyramids[p]->size() != levels || source_pyramids[p]->size() != levels)
        throw InvalidParameter("A3D_INVALID_PARAMETER: every pyramid of a batch must have the same number of levels");
      for (size_t l = 0; l < levels; ++l) {
        t.push_back((*target_pyramids[p])[l].raw());
        s.push_back((*source_pyramids[p])[l].raw());
      }
    }
    auto prm = params.to_c();
    check(a3d_multiscale_batch_new(ctx.raw(), prm.data(), prm.size(), n_pairs_, levels, t.data(), s.data(), &b_));
  }
  ~MultiscaleAlignBatch() { a3d_multiscale_batch_free(b_); }
  MultiscaleAlignBatch(const MultiscaleAlignBatch&) = delete;
  MultiscaleAlignBatch& operator=(const MultiscaleAlignBatch&) = delete;
  /// The same batch object on other pyramids (same pair and level counts): nothing is allocated or freed.
  void rebind(const std::vector<const std::vector<RangeImage>*>& target_pyramids,
              const std::vector<const std::vector<RangeImage>*>& source_pyramids) {
    if (target_pyramids.size() != n_pairs_ || source_pyramids.size() != n_pairs_)
      throw InvalidParameter("A3D_INVALID_PARAMETER: rebind needs the batch's own number of pairs");
    std::vector<const a3d_device_image*> t, s;
    for (size_t p = 0; p < n_pairs_; ++p) {
      for (const auto& r : *target_pyramids[p]) t.push_back(r.raw());
      for (const auto& r : *source_pyramids[p]) s.push_back(r.raw());
    }
    if (t.size() != s.size() || t.size() % n_pairs_) throw InvalidParameter("A3D_INVALID_PARAMETER: ragged pyramids");
    check(a3d_multiscale_batch_rebind(b_, t.data(), s.data()));
  }
  std::vector<Transform> align() {
    std::vector<a3d_pose> poses(n_pairs_);
    status_.assign(n_pairs_, 0);
    check(a3d_multiscale_batch_align(b_, poses.data(), nullptr, status_.data()));
    std::vector<Transform> out;
    for (const a3d_pose& p : poses) out.push_back(Transform::from_c(p));
    return out;
  }
  /// One pass without synchronising the host; results() reads it (and waits for it alone).  Two batches alternating
  /// over a stream of rounds — rebind + enqueue the next while the previous computes — keep the GPU busy.
  void enqueue() { check(a3d_multiscale_batch_align(b_, nullptr, nullptr, nullptr)); }
  std::vector<Transform> results() {
    std::vector<a3d_pose> poses(n_pairs_);
    status_.assign(n_pairs_, 0);
    check(a3d_multiscale_batch_results(b_, poses.data(), status_.data()));
    std::vector<Transform> out;
    for (const a3d_pose& p : poses) out.push_back(Transform::from_c(p));
    return out;
  }
  const std::vector<int32_t>& status() const { return status_; }

 private:
  size_t n_pairs_;
  a3d_multiscale_batch* b_ = nullptr;
  std::vector<int32_t> status_;
};

/// One context per entry of a device list (a3d_multi_context): device(i) is the context the frames of the pairs
/// entry i owns are built on (shard(n_pairs, i) tells which pairs those are: contiguous blocks).
class MultiContext {
 public:
  explicit MultiContext(const std::vector<int32_t>& device_ids) {
    check(a3d_multi_context_create(device_ids.data(), device_ids.size(), &mc_));
  }
  ~MultiContext() { a3d_multi_context_destroy(mc_); }
  MultiContext(const MultiContext&) = delete;
  MultiContext& operator=(const MultiContext&) = delete;
  size_t size() const { return a3d_multi_context_size(mc_); }
  Context device(size_t i) const { return Context::borrowed(a3d_multi_context_device(mc_, i)); }
  std::pair<uint64_t, uint64_t> shard(uint64_t n_pairs, size_t i) const {
    uint64_t lo = 0, hi = 0;
    check(a3d_multi_shard_range(n_pairs, size(), i, &lo, &hi));
    return {lo, hi};
  }
  a3d_multi_context* raw() const { return mc_; }

 private:
  a3d_multi_context* mc_ = nullptr;
};

/// P independent MultiscaleAlign jobs over the devices of a MultiContext (a3d_multiscale_batch_new_multi): pyramids in
/// global pair order, each pair's images resident on the device that owns it; align() gathers every device's poses.
class MultiscaleAlignMultiBatch {
 public:
  MultiscaleAlignMultiBatch(const MultiContext& mc, const MsIcpParams& params,
                            const std::vector<const std::vector<RangeImage>*>& target_pyramids,
                            const std::vector<const std::vector<RangeImage>*>& source_pyramids)
      : n_pairs_(target_pyramids.size()) {
    if (target_pyramids.size() != source_pyramids.size() || target_pyramids.empty())
      throw InvalidParameter("A3D_INVALID_PARAMETER: one target and one source pyramid per pair are required");
    const size_t levels = target_pyramids[0]->size();
    std::vector<const a3d_device_image*> t, s;
    for (size_t p = 0; p < n_pairs_; ++p)
      for (size_t l = 0; l < levels; ++l) {
        t.push_back(target_pyramids[p]->at(l).raw());
        s.push_back(source_pyramids[p]->at(l).raw());
      }
    auto prm = params.to_c();
    check(a3d_multiscale_batch_new_multi(mc.raw(), prm.data(), prm.size(), n_pairs_, levels, t.data(), s.data(), &b_));
  }
  ~MultiscaleAlignMultiBatch() { a3d_multiscale_multi_batch_free(b_); }
  MultiscaleAlignMultiBatch(const MultiscaleAlignMultiBatch&) = delete;
  MultiscaleAlignMultiBatch& operator=(const MultiscaleAlignMultiBatch&) = delete;
  /// One Transform per pair; matrices (nullable) receives the gathered [n_pairs][16] row-major 4x4 poses.
  std::vector<Transform> align(std::vector<float>* matrices = nullptr) {
    std::vector<a3d_pose> poses(n_pairs_);
    status_.assign(n_pairs_, 0);
    if (matrices) matrices->assign(n_pairs_ * 16, 0.0f);
    check(a3d_multiscale_multi_batch_align(b_, poses.data(), matrices ? matrices->data() : nullptr, status_.data(), nullptr));
    std::vector<Transform> out;
    for (const a3d_pose& p : poses) out.push_back(Transform::from_c(p));
    return out;
  }
  const std::vector<int32_t>& status() const { return status_; }

 private:
  size_t n_pairs_;
  a3d_multiscale_multi_batch* b_ = nullptr;
  std::vector<int32_t> status_;
};

}  // namespace align3d
