// TEST INFRASTRUCTURE — CPU oracle for the align3d ICP hot path.  Not part of the product:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// Scalar math shared by the oracle's translation units: Rust cast semantics, nalgebra 0.30.1
// vector/quaternion/isometry arithmetic (nalgebra is not vendored under the reference; its published
// algorithm is restated here and pinned through the reference's own KATs in src/transform.rs:321-411),
// the SE3 exponential (src/transform.rs:44-108) and GaussNewton<6> (src/optim/gaussnewton.rs:9-134).
//
// Build with -ffp-contract=off: Rust never contracts a*b+c into an fma.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

// ---- Rust `as` casts -------------------------------------------------------------------------
// float -> int casts saturate and map NaN to 0 (Rust reference, "Numeric cast").
inline int32_t f32_as_i32(float x) {
  if (std::isnan(x)) return 0;
  if (x >= 2147483648.0f) return INT32_MAX;
  if (x <= -2147483648.0f) return INT32_MIN;
  return (int32_t)x;
}
inline uint64_t f32_as_usize(float x) {
  if (std::isnan(x)) return 0;
  if (x <= 0.0f) return 0;
  if (x >= 18446744073709551616.0f) return UINT64_MAX;
  return (uint64_t)x;
}
inline uint64_t f64_as_usize(double x) {
  if (std::isnan(x)) return 0;
  if (x <= 0.0) return 0;
  if (x >= 18446744073709551616.0) return UINT64_MAX;
  return (uint64_t)x;
}
// i32 -> usize sign-extends: negative values become huge and fail every `< dim` test.
inline uint64_t i32_as_usize(int32_t x) { return (uint64_t)(int64_t)x; }

// ---- Vector3<f32> ----------------------------------------------------------------------------
struct V3 {
  float x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// nalgebra dot for 3-vectors: (a0*b0 + a1*b1) + a2*b2
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float norm_squared(V3 a) { return dot(a, a); }
inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// ---- Isometry3<f32> --------------------------------------------------------------------------
struct Quat {
  float i, j, k, w;
};
struct Pose {
  V3 t;
  Quat q;
};
inline Pose pose_eye() { return {{0, 0, 0}, {0, 0, 0, 1}}; }

// UnitQuaternion * Vector3 (nalgebra quaternion_ops): t = 2 (q_v x v); v' = t*w + q_v x t + v
inline V3 rotate(const Quat& q, V3 v) {
  V3 qv{q.i, q.j, q.k};
  V3 t = cross(qv, v) * 2.0f;
  V3 c = cross(qv, t);
  return (t * q.w + c) + v;
}
// Transform::transform_vector (src/transform.rs:138-140): rotation * rhs + translation
inline V3 transform_vector(const Pose& p, V3 v) { return rotate(p.q, v) + p.t; }
// Transform::transform_normal (src/transform.rs:151-153)
inline V3 transform_normal(const Pose& p, V3 v) { return rotate(p.q, v); }

// Quaternion * Quaternion (Hamilton product, nalgebra operand order)
inline Quat qmul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.i * b.i - a.j * b.j - a.k * b.k;
  r.i = a.w * b.i + a.i * b.w + a.j * b.k - a.k * b.j;
  r.j = a.w * b.j - a.i * b.k + a.j * b.w + a.k * b.i;
  r.k = a.w * b.k + a.i * b.j - a.j * b.i + a.k * b.w;
  return r;
}
// UnitQuaternion::from_quaternion = Unit::new_normalize: 4-vector norm (a+c)+(b+d), then divide.
inline Quat qnormalize(Quat q) {
  float a = q.i * q.i, b = q.j * q.j, c = q.k * q.k, d = q.w * q.w;
  a += c;
  b += d;
  float n = std::sqrt(a + b);
  return {q.i / n, q.j / n, q.k / n, q.w / n};
}
// Isometry3 * Isometry3 (src/transform.rs:205-220): t = t1 + R1 t2 ; q = q1 q2 (not renormalised)
inline Pose compose(const Pose& a, const Pose& b) {
  Pose r;
  r.t = a.t + rotate(a.q, b.t);
  r.q = qmul(a.q, b.q);
  return r;
}
inline Pose inverse(const Pose& a) {
  Pose r;
  r.q = {-a.q.i, -a.q.j, -a.q.k, a.q.w};
  V3 rt = rotate(r.q, a.t);
  r.t = {-rt.x, -rt.y, -rt.z};
  return r;
}
// UnitQuaternion::angle: 2 atan2(|v|, |w|)
inline float qangle(const Quat& q) {
  float n = std::sqrt(norm_squared(V3{q.i, q.j, q.k}));
  return 2.0f * std::atan2(n, std::fabs(q.w));
}

// ---- Transform::exp (src/transform.rs:44-108) -------------------------------------------------
inline void exp_so3(V3 omega, float* theta_out, Quat* q_out) {
  const float EPS = 1e-8f;
  float theta_sq = norm_squared(omega);
  float theta, imag, real;
  if (theta_sq < EPS * EPS) {
    float theta_po4 = theta_sq * theta_sq;
    theta = 0.0f;
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_po4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_po4;
  } else {
    theta = std::sqrt(theta_sq);
    float half = 0.5f * theta;
    imag = std::sin(half) / theta;
    real = std::cos(half);
  }
  *theta_out = theta;
  *q_out = qnormalize(Quat{imag * omega.x, imag * omega.y, imag * omega.z, real});
}

struct M3 {
  float m[3][3];
};
inline M3 m3_scale(const M3& a, float s) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] * s;
  return r;
}
inline M3 m3_add(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
// nalgebra gemv/gemm are column-axpy: y_i = ((a_i0 x_0) + a_i1 x_1) + a_i2 x_2
inline V3 m3_mulv(const M3& a, V3 v) {
  float x[3] = {v.x, v.y, v.z};
  float y[3];
  for (int i = 0; i < 3; ++i) {
    float acc = a.m[i][0] * x[0];
    acc = a.m[i][1] * x[1] + acc;
    acc = a.m[i][2] * x[2] + acc;
    y[i] = acc;
  }
  return {y[0], y[1], y[2]};
}
inline M3 m3_mul(const M3& a, const M3& b) {
  M3 r;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) {
      float acc = a.m[i][0] * b.m[0][j];
      acc = a.m[i][1] * b.m[1][j] + acc;
      acc = a.m[i][2] * b.m[2][j] + acc;
      r.m[i][j] = acc;
    }
  return r;
}

// xyz_so3 = [rho_x, rho_y, rho_z, omega_x, omega_y, omega_z]
inline Pose exp_se3(const float u[6]) {
  const float EPS = 1e-8f;
  V3 omega{u[3], u[4], u[5]};
  float theta;
  Quat q;
  exp_so3(omega, &theta, &q);
  float theta_sq = theta * theta;
  M3 I{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  M3 W{{{0, -omega.z, omega.y}, {omega.z, 0, -omega.x}, {-omega.y, omega.x, 0}}};
  M3 V;
  if (theta_sq < EPS) {
    V = m3_add(I, m3_scale(W, 0.5f));
  } else {
    M3 W2 = m3_mul(W, W);
    float a = (1.0f - std::cos(theta)) / theta_sq;
    float b = (theta - std::sin(theta)) / (theta_sq * theta);
    V = m3_add(m3_add(I, m3_scale(W, a)), m3_scale(W2, b));
  }
  Pose p;
  p.t = m3_mulv(V, V3{u[0], u[1], u[2]});
  p.q = q;
  return p;
}

// ---- GaussNewton<6> (src/optim/gaussnewton.rs) ------------------------------------------------
// Acc = float is the reference; Acc = double keeps the per-sample f32 products but sums them in
// f64, which makes the sums independent of summation order (used to judge the GPU reduction).
template <typename Acc>
struct GaussNewton6 {
  Acc H[6][6];
  Acc g[6];
  Acc ssq;
  uint64_t count;
  GaussNewton6() { reset(); }
  void reset() {
    std::memset(H, 0, sizeof(H));
    std::memset(g, 0, sizeof(g));
    ssq = 0;
    count = 0;
  }
  // gaussnewton.rs:47-77
  void step(float residual, const float J[6]) {
    for (int i = 0; i < 6; ++i) {
      float ival = J[i];
      g[i] += (Acc)(ival * residual);
      H[i][i] += (Acc)(ival * ival);
      for (int j = i + 1; j < 6; ++j) {
        float mul = ival * J[j];
        H[i][j] += (Acc)mul;
        H[j][i] += (Acc)mul;
      }
    }
    ssq += (Acc)(residual * residual);
    count += 1;
  }
  // gaussnewton.rs:101-106
  void add(const GaussNewton6& o) {
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j < 6; ++j) H[i][j] += o.H[i][j];
      g[i] += o.g[i];
    }
    ssq += o.ssq;
    count += o.count;
  }
};

struct GnF32 {
  float H[6][6];
  float g[6];
  float ssq;
  uint64_t count;
};
template <typename Acc>
inline GnF32 to_f32(const GaussNewton6<Acc>& a) {
  GnF32 r;
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) r.H[i][j] = (float)a.H[i][j];
    r.g[i] = (float)a.g[i];
  }
  r.ssq = (float)a.ssq;
  r.count = a.count;
  return r;
}
// gaussnewton.rs:115-121 — H by w^2, g and ssq by w, counts summed (inconsistent by design).
inline void add_weighted(GnF32& self, const GnF32& other, float w1, float w2) {
  float w1s = w1 * w1, w2s = w2 * w2;
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) self.H[i][j] = self.H[i][j] * w1s + other.H[i][j] * w2s;
    self.g[i] = self.g[i] * w1 + other.g[i] * w2;
  }
  self.ssq = self.ssq * w1 + other.ssq * w2;
  self.count += other.count;
}
// gaussnewton.rs:124-128
inline void weight(GnF32& self, float w) {
  float ws = w * w;
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) self.H[i][j] *= ws;
    self.g[i] *= w;
  }
  self.ssq *= w;
}
// gaussnewton.rs:131-133
inline float mean_squared_residual(const GnF32& s) { return s.ssq / (float)s.count; }

// gaussnewton.rs:84-93 with nalgebra's Cholesky<f64> (left-looking, column by column) and its
// forward / adjoint substitution.  Returns false for solve() == None.
inline bool solve(const GnF32& s, float out[6]) {
  if (s.count == 0) return false;
  double L[6][6], b[6];
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) L[i][j] = (double)s.H[i][j];
    b[i] = (double)s.g[i];
  }
  for (int j = 0; j < 6; ++j) {
    for (int k = 0; k < j; ++k) {
      double factor = -L[j][k];
      for (int i = j; i < 6; ++i) L[i][j] = factor * L[i][k] + L[i][j];
    }
    double diag = L[j][j];
    if (diag == 0.0) return false;
    if (!(diag >= 0.0)) return false;  // try_sqrt: negative or NaN
    double denom = std::sqrt(diag);
    L[j][j] = denom;
    for (int i = j + 1; i < 6; ++i) L[i][j] /= denom;
  }
  // solve_lower_triangular (column oriented)
  for (int i = 0; i < 6; ++i) {
    double coeff = b[i] / L[i][i];
    b[i] = coeff;
    for (int r = i + 1; r < 6; ++r) b[r] = -coeff * L[r][i] + b[r];
  }
  // ad_solve_lower_triangular: L^T x = b, dot-product form
  for (int i = 5; i >= 0; --i) {
    double d = 0.0;
    for (int r = i + 1; r < 6; ++r) d += L[r][i] * b[r];
    b[i] = (b[i] - d) / L[i][i];
  }
  for (int i = 0; i < 6; ++i) out[i] = (float)b[i];
  return true;
}

}  // namespace orc
