// TEST INFRASTRUCTURE — CPU oracle, part 1: math KAT hooks, IntensityMap, ImageIcp, MultiscaleAlign.
// A restatement of the reference's algorithm, cited per function; never loaded by the product.
// PARITY PINNING: checked in tests/test_oracle_kat.py against every known-answer test the reference
// holds for this path (gaussnewton.rs:141-167, transform.rs:321-411, camera.rs:210-243,
// intensity_map.rs:229-262 property form, image_icp.rs:181-200 smoke threshold).
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

#include "a3d_oracle.h"
#include "oracle_math.hpp"

using namespace orc;

namespace {

inline Pose from_c(const a3d_pose* p) {
  return Pose{{p->t[0], p->t[1], p->t[2]}, {p->q[0], p->q[1], p->q[2], p->q[3]}};
}
inline void to_c(const Pose& p, a3d_pose* o) {
  o->t[0] = p.t.x;
  o->t[1] = p.t.y;
  o->t[2] = p.t.z;
  o->q[0] = p.q.i;
  o->q[1] = p.q.j;
  o->q[2] = p.q.k;
  o->q[3] = p.q.w;
}
inline void gn_to_c(const GnF32& s, a3d_gn_state* o) {
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) o->hessian[i * 6 + j] = s.H[i][j];
    o->gradient[i] = s.g[i];
  }
  o->squared_residual_sum = s.ssq;
  o->count = s.count;
}
inline GnF32 gn_from_c(const a3d_gn_state* o) {
  GnF32 s;
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) s.H[i][j] = o->hessian[i * 6 + j];
    s.g[i] = o->gradient[i];
  }
  s.ssq = o->squared_residual_sum;
  s.count = o->count;
  return s;
}

inline V3 load3(const float* base, uint64_t idx) {
  return {base[3 * idx], base[3 * idx + 1], base[3 * idx + 2]};
}

// RangeImage::get_point (src/range_image/structure.rs:175-181)
inline bool get_point(const a3d_range_image_view& im, uint64_t row, uint64_t col, V3* out) {
  if (col < im.width && row < im.height && im.mask[row * im.width + col] == 1) {
    *out = load3(im.points, row * im.width + col);
    return true;
  }
  return false;
}

// IntensityMap::bilinear (src/intensity_map.rs:150-169); map is [(h+2)][(w+2)]
inline float imap_at(const float* map, uint64_t mw, uint64_t mh, uint64_t r, uint64_t c) {
  if (r >= mh || c >= mw) return std::numeric_limits<float>::quiet_NaN();  // reference would panic
  return map[r * mw + c];
}
inline float bilinear(const float* map, uint64_t mw, uint64_t mh, float u, float v) {
  uint64_t ui = f32_as_usize(u), vi = f32_as_usize(v);
  float u_frac = u - (float)ui, v_frac = v - (float)vi;
  float v00 = imap_at(map, mw, mh, vi, ui), v10 = imap_at(map, mw, mh, vi, ui + 1);
  float v01 = imap_at(map, mw, mh, vi + 1, ui), v11 = imap_at(map, mw, mh, vi + 1, ui + 1);
  float u0 = v00 * (1.0f - u_frac) + v10 * u_frac;
  float u1 = v01 * (1.0f - u_frac) + v11 * u_frac;
  return u0 * (1.0f - v_frac) + u1 * v_frac;
}
// IntensityMap::bilinear_grad (src/intensity_map.rs:184-210)
inline void bilinear_grad(const float* map, uint64_t mw, uint64_t mh, float u, float v, float* value,
                          float* gu, float* gv) {
  const float H = 0.005f;
  const float H_INV = 1.0f / H;
  float val = bilinear(map, mw, mh, u, v);
  float uh = bilinear(map, mw, mh, u + H, v);
  float vh = bilinear(map, mw, mh, u, v + H);
  *value = val;
  *gu = (uh - val) * H_INV;
  *gv = (vh - val) * H_INV;
}

// The per-pixel body of ImageIcp::align (src/icp/image_icp.rs:101-139) over flat pixels [begin, end).
template <typename Acc>
void image_icp_chunk(const a3d_icp_params& prm, const a3d_range_image_view& tgt,
                     const a3d_range_image_view& src, const Pose& T, uint64_t begin, uint64_t end,
                     GaussNewton6<Acc>& geom, GaussNewton6<Acc>& color) {
  const float fx = (float)tgt.fx, fy = (float)tgt.fy, cx = (float)tgt.cx, cy = (float)tgt.cy;
  const float max_color_distance_sqr = prm.max_color_distance * prm.max_color_distance;
  const float max_distance_sqr = prm.max_distance * prm.max_distance;
  const uint64_t mw = tgt.width + 2, mh = tgt.height + 2;
  for (uint64_t i = begin; i < end; ++i) {
    if (src.mask[i] == 0) continue;
    V3 p = transform_vector(T, load3(src.points, i));
    // CameraIntrinsics::project (src/camera.rs:64-70)
    float z = p.z;
    float u = p.x * fx / z + cx;
    float v = p.y * fy / z + cy;
    int32_t u_int = f32_as_i32(u + 0.5f), v_int = f32_as_i32(v + 0.5f);
    uint64_t row = i32_as_usize(v_int), col = i32_as_usize(u_int);
    V3 q;
    if (!get_point(tgt, row, col, &q)) continue;
    if (norm_squared(q - p) > max_distance_sqr) continue;
    V3 n = load3(tgt.normals, row * tgt.width + col);
    // extra_math::angle_between_normals(&p, &n) (src/extra_math.rs:13-15) on the POINT p; NaN passes.
    float ang = std::fabs(std::acos(dot(p, n)));
    if (ang >= prm.max_normal_angle) continue;
    // PointPlaneDistance::jacobian (src/icp/cost_function.rs:33-41)
    float residual = dot(q - p, n);
    V3 tw = cross(p, n);
    float J[6] = {n.x, n.y, n.z, tw.x, tw.y, tw.z};
    geom.step(residual, J);
    // colour part (image_icp.rs:130-138)
    float tc, du, dv;
    bilinear_grad(tgt.intensity_map, mw, mh, u, v, &tc, &du, &dv);
    float sc = (float)src.intensities[i] * 0.003921569f;
    // CameraIntrinsics::project_grad (src/camera.rs:82-89)
    float zz = z * z;
    float dfx = fx / z, dcx = -p.x * fx / zz;
    float dfy = fy / z, dcy = -p.y * fy / zz;
    V3 g{du * dfx, dv * dfy, du * dcx + dv * dcy};
    float rc = sc - tc;
    V3 twc = cross(p, g);
    float Jc[6] = {g.x, g.y, g.z, twc.x, twc.y, twc.z};
    if (rc * rc <= max_color_distance_sqr) color.step(rc, Jc);
  }
}

a3d_status check_image_inputs(const a3d_range_image_view* target, const a3d_range_image_view* source) {
  if (!target || !source || !target->points || !target->mask || !source->points || !source->mask)
    return A3D_INVALID_PARAMETER;
  if (!target->intensity_map || !target->normals || !source->intensities) return A3D_MISSING_FIELD;
  return A3D_OK;
}

constexpr uint64_t BATCH_SIZE = 4096;  // image_icp.rs:74

// Chunk merge order.  The reference collects the per-chunk sub-optimisers through rayon's `par_bridge()`
// (image_icp.rs:96,143), which does NOT preserve the order of the chunks, and then adds them in the order of that
// Vec (:145-148): any permutation of the chunks is a result the reference can produce.  0 = chunk order; any
// other value seeds one permutation per pass (splitmix64 Fisher-Yates), advanced after every pass.
thread_local uint64_t g_merge_order_state = 0;

inline uint64_t splitmix64_next(uint64_t& x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// One full pass: chunk sub-optimisers (image_icp.rs:76-143) merged in chunk order (:145-148).
template <typename Acc>
void image_icp_pass(const a3d_icp_params& prm, const a3d_range_image_view& tgt,
                    const a3d_range_image_view& src, const Pose& T, int threads,
                    GaussNewton6<Acc>& geom, GaussNewton6<Acc>& color) {
  const uint64_t n = src.width * src.height;
  const uint64_t n_chunks = (n + BATCH_SIZE - 1) / BATCH_SIZE;
  std::vector<GaussNewton6<Acc>> sub_geom(n_chunks), sub_color(n_chunks);
  auto work = [&](uint64_t c0, uint64_t c1) {
    for (uint64_t c = c0; c < c1; ++c)
      image_icp_chunk<Acc>(prm, tgt, src, T, c * BATCH_SIZE, std::min(n, (c + 1) * BATCH_SIZE),
                           sub_geom[c], sub_color[c]);
  };
  if (threads <= 1) {
    work(0, n_chunks);
  } else {
    std::vector<std::thread> pool;
    uint64_t per = (n_chunks + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
      uint64_t c0 = std::min(n_chunks, (uint64_t)t * per), c1 = std::min(n_chunks, c0 + per);
      if (c0 < c1) pool.emplace_back(work, c0, c1);
    }
    for (auto& th : pool) th.join();
  }
  std::vector<uint64_t> order(n_chunks);
  for (uint64_t c = 0; c < n_chunks; ++c) order[c] = c;
  if (g_merge_order_state != 0)
    for (uint64_t c = n_chunks; c > 1; --c) std::swap(order[c - 1], order[splitmix64_next(g_merge_order_state) % c]);
  for (uint64_t k = 0; k < n_chunks; ++k) {
    color.add(sub_color[order[k]]);
    geom.add(sub_geom[order[k]]);
  }
}

a3d_status image_icp_align_impl(const a3d_icp_params& prm, const a3d_range_image_view& tgt,
                                const a3d_range_image_view& src, const Pose& init, int threads,
                                Pose* out, float* trace) {
  Pose optim = init;
  float best_residual = std::numeric_limits<float>::infinity();
  Pose best = optim;
  for (uint64_t it = 0; it < prm.max_iterations; ++it) {
    GaussNewton6<float> geom, color;
    image_icp_pass<float>(prm, tgt, src, optim, threads, geom, color);
    GnF32 g = to_f32(geom), c = to_f32(color);
    add_weighted(g, c, prm.weight, prm.color_weight);  // image_icp.rs:150
    float residual = mean_squared_residual(g);         // :151
    float update[6];
    if (!solve(g, update)) return A3D_SOLVE_FAILED;    // :152 unwrap
    optim = compose(exp_se3(update), optim);           // :153
    if (trace) {
      float* tr = trace + 8 * it;
      tr[0] = residual;
      tr[1] = optim.t.x, tr[2] = optim.t.y, tr[3] = optim.t.z;
      tr[4] = optim.q.i, tr[5] = optim.q.j, tr[6] = optim.q.k, tr[7] = optim.q.w;
    }
    if (residual < best_residual) {  // :158-161 (the transform AFTER the update is stored)
      best_residual = residual;
      best = optim;
    }
  }
  *out = best;
  return A3D_OK;
}

}  // namespace

extern "C" {

void orc_pose_eye(a3d_pose* out) { to_c(pose_eye(), out); }
void orc_exp_se3(const float u[6], a3d_pose* out) { to_c(exp_se3(u), out); }
void orc_compose(const a3d_pose* a, const a3d_pose* b, a3d_pose* out) {
  to_c(compose(from_c(a), from_c(b)), out);
}
void orc_inverse(const a3d_pose* a, a3d_pose* out) { to_c(inverse(from_c(a)), out); }
void orc_transform_points(const a3d_pose* p, const float* in, uint64_t n, float* out) {
  Pose T = from_c(p);
  for (uint64_t i = 0; i < n; ++i) {
    V3 r = transform_vector(T, load3(in, i));
    out[3 * i] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
  }
}
void orc_transform_normals(const a3d_pose* p, const float* in, uint64_t n, float* out) {
  Pose T = from_c(p);
  for (uint64_t i = 0; i < n; ++i) {
    V3 r = transform_normal(T, load3(in, i));
    out[3 * i] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
  }
}
void orc_pose_to_matrix(const a3d_pose* p, float m[16]) {
  // UnitQuaternion::to_rotation_matrix (nalgebra): ww+ii-jj-kk etc.
  const float i = p->q[0], j = p->q[1], k = p->q[2], w = p->q[3];
  float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
  float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f, ik = i * k * 2.0f;
  float jk = j * k * 2.0f, wi = w * i * 2.0f;
  float R[9] = {ww + ii - jj - kk, ij - wk, wj + ik, wk + ij, ww - ii + jj - kk,
                jk - wi,           ik - wj, wi + jk, ww - ii - jj + kk};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) m[r * 4 + c] = R[r * 3 + c];
    m[r * 4 + 3] = p->t[r];
  }
  m[12] = m[13] = m[14] = 0.0f;
  m[15] = 1.0f;
}
void orc_pose_from_matrix(const float m[16], a3d_pose* out) {
  // Transform::from_matrix4 (src/transform.rs:112-118).  nalgebra first projects the 3x3 block onto
  // SO(3) iteratively; the ground-truth matrices this is used for are rotations to 1e-7, so the
  // standard trace-based extraction is used (ground truth only feeds smoke thresholds, not parity).
  double r00 = m[0], r01 = m[1], r02 = m[2], r10 = m[4], r11 = m[5], r12 = m[6], r20 = m[8],
         r21 = m[9], r22 = m[10];
  double tr = r00 + r11 + r22, w, i, j, k;
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0) * 2;
    w = 0.25 * s, i = (r21 - r12) / s, j = (r02 - r20) / s, k = (r10 - r01) / s;
  } else if (r00 > r11 && r00 > r22) {
    double s = std::sqrt(1.0 + r00 - r11 - r22) * 2;
    w = (r21 - r12) / s, i = 0.25 * s, j = (r01 + r10) / s, k = (r02 + r20) / s;
  } else if (r11 > r22) {
    double s = std::sqrt(1.0 + r11 - r00 - r22) * 2;
    w = (r02 - r20) / s, i = (r01 + r10) / s, j = 0.25 * s, k = (r12 + r21) / s;
  } else {
    double s = std::sqrt(1.0 + r22 - r00 - r11) * 2;
    w = (r10 - r01) / s, i = (r02 + r20) / s, j = (r12 + r21) / s, k = 0.25 * s;
  }
  double n = std::sqrt(w * w + i * i + j * j + k * k);
  out->q[0] = (float)(i / n), out->q[1] = (float)(j / n), out->q[2] = (float)(k / n);
  out->q[3] = (float)(w / n);
  out->t[0] = m[3], out->t[1] = m[7], out->t[2] = m[11];
}
void orc_transform_metrics(const a3d_pose* lhs, const a3d_pose* rhs, float* angle, float* translation) {
  Pose d = compose(inverse(from_c(lhs)), from_c(rhs));
  *angle = qangle(d.q);
  *translation = std::sqrt(norm_squared(d.t));
}
void orc_project(double fx, double fy, double cx, double cy, const float p[3], float uv[2]) {
  float z = p[2];
  uv[0] = p[0] * (float)fx / z + (float)cx;
  uv[1] = p[1] * (float)fy / z + (float)cy;
}
void orc_project_grad(double fx, double fy, const float p[3], float o[4]) {
  float z = p[2], zz = z * z;
  o[0] = (float)fx / z, o[1] = -p[0] * (float)fx / zz;
  o[2] = (float)fy / z, o[3] = -p[1] * (float)fy / zz;
}
void orc_backproject(double fx, double fy, double cx, double cy, float x, float y, float z, float o[3]) {
  o[0] = (x - (float)cx) * z / (float)fx;
  o[1] = (y - (float)cy) * z / (float)fy;
  o[2] = z;
}

void orc_gn_steps(const float* r, const float* J, uint64_t n, a3d_gn_state* out) {
  GaussNewton6<float> gn;
  for (uint64_t i = 0; i < n; ++i) gn.step(r[i], J + 6 * i);
  gn_to_c(to_f32(gn), out);
}
void orc_gn_add_weighted(a3d_gn_state* self, const a3d_gn_state* other, float w1, float w2) {
  GnF32 a = gn_from_c(self), b = gn_from_c(other);
  add_weighted(a, b, w1, w2);
  gn_to_c(a, self);
}
void orc_gn_weight(a3d_gn_state* self, float w) {
  GnF32 a = gn_from_c(self);
  weight(a, w);
  gn_to_c(a, self);
}
float orc_gn_mean_squared_residual(const a3d_gn_state* s) { return mean_squared_residual(gn_from_c(s)); }
int32_t orc_gn_solve(const a3d_gn_state* s, float out[6]) { return solve(gn_from_c(s), out) ? 1 : 0; }

// IntensityMap::from_luma_image -> zeros + fill (src/intensity_map.rs:29-92)
void orc_intensity_map_fill(const uint8_t* luma, uint64_t w, uint64_t h, float* map) {
  const uint64_t mw = w + 2, mh = h + 2;
  std::fill(map, map + mw * mh, 0.0f);
  for (uint64_t r = 0; r < h; ++r)
    for (uint64_t c = 0; c < w; ++c) map[r * mw + c] = (float)luma[r * w + c] / 255.0f;
  if (h == 0 || w == 0) return;
  for (uint64_t c = 0; c + 1 < w; ++c) {  // "border X": cols 0..w-2 only (:60-65)
    float b = map[(h - 1) * mw + c];
    for (int k = 0; k < 2; ++k) map[(h + k) * mw + c] = b;
  }
  for (uint64_t r = 0; r + 1 < h; ++r) {  // rows 0..h-2 only (:67-72)
    float b = map[r * mw + (w - 1)];
    for (int k = 0; k < 2; ++k) map[r * mw + (w + k)] = b;
  }
  float last = (float)luma[(h - 1) * w + (w - 1)] / 255.0f;
  for (int k = 0; k < 2; ++k) map[(h + k) * mw + (w + k)] = last;  // (:74-77)
}
void orc_intensity_map_bilinear_grad(const float* map, uint64_t w, uint64_t h, float u, float v,
                                     float out3[3]) {
  bilinear_grad(map, w + 2, h + 2, u, v, &out3[0], &out3[1], &out3[2]);
}

void orc_set_chunk_merge_order(uint64_t seed) { g_merge_order_state = seed; }

a3d_status orc_image_icp_accumulate(const a3d_icp_params* prm, const a3d_range_image_view* target,
                                    const a3d_range_image_view* source, const a3d_pose* pose,
                                    int32_t accum_f64, a3d_gn_state* out_geom, a3d_gn_state* out_color) {
  a3d_status st = check_image_inputs(target, source);
  if (st != A3D_OK) return st;
  Pose T = pose ? from_c(pose) : pose_eye();
  if (accum_f64) {
    GaussNewton6<double> g, c;
    image_icp_pass<double>(*prm, *target, *source, T, 1, g, c);
    gn_to_c(to_f32(g), out_geom);
    gn_to_c(to_f32(c), out_color);
  } else {
    GaussNewton6<float> g, c;
    image_icp_pass<float>(*prm, *target, *source, T, 1, g, c);
    gn_to_c(to_f32(g), out_geom);
    gn_to_c(to_f32(c), out_color);
  }
  return A3D_OK;
}

a3d_status orc_image_icp_align(const a3d_icp_params* prm, const a3d_range_image_view* target,
                               const a3d_range_image_view* source, const a3d_pose* init_pose,
                               int32_t threads, a3d_pose* out_pose, float* trace) {
  a3d_status st = check_image_inputs(target, source);
  if (st != A3D_OK) return st;
  Pose out;
  st = image_icp_align_impl(*prm, *target, *source, init_pose ? from_c(init_pose) : pose_eye(), threads,
                            &out, trace);
  if (st == A3D_OK) to_c(out, out_pose);
  return st;
}

a3d_status orc_multiscale_align(const a3d_icp_params* params, uint64_t n_params,
                                const a3d_range_image_view* target_pyramid, uint64_t n_target,
                                const a3d_range_image_view* source_pyramid, uint64_t n_source,
                                int32_t threads, a3d_pose* out_pose) {
  if (n_params != n_target) return A3D_INVALID_PARAMETER;  // multiscale.rs:30-34
  uint64_t n = std::min(n_params, std::min(n_target, n_source));  // izip! truncates (:54-58)
  Pose optim = pose_eye();
  for (uint64_t k = n; k-- > 0;) {  // .rev(): coarsest first
    a3d_status st = check_image_inputs(&target_pyramid[k], &source_pyramid[k]);
    if (st != A3D_OK) return st;
    Pose res;
    st = image_icp_align_impl(params[k], target_pyramid[k], source_pyramid[k], optim, threads, &res,
                              nullptr);
    if (st != A3D_OK) return st;
    optim = res;
  }
  to_c(optim, out_pose);
  return A3D_OK;
}

}  // extern "C"
