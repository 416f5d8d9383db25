// Scalar math shared by the HIP kernels and the host-side pose code of libalign3d_hip.so.
// Every expression keeps the reference's evaluation order (Rust/nalgebra never contract a*b+c), and
// the library is compiled with -ffp-contract=off, so a per-sample value computed here is the
// bit-identical f32 the reference computes; only sums over samples are re-associated.
//
//   Transform::transform_vector / transform_normal   src/transform.rs:138-153
//   Isometry3 * Isometry3                            src/transform.rs:205-220
//   Transform::exp(Se3)                              src/transform.rs:44-108
//   CameraIntrinsics::project / project_grad         src/camera.rs:64-89
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define A3D_HD __host__ __device__ __forceinline__

namespace a3d {

struct V3 {
  float x, y, z;
};
A3D_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
A3D_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
A3D_HD V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
A3D_HD V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// nalgebra 3-vector dot: (a0 b0 + a1 b1) + a2 b2
A3D_HD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
A3D_HD float norm_squared(V3 a) { return dot(a, a); }
A3D_HD V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct Quat {
  float i, j, k, w;
};
struct Pose {
  V3 t;
  Quat q;
};
A3D_HD Pose pose_eye() { return {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 1.f}}; }

// UnitQuaternion * Vector3: t = 2 (q_v x v); v' = (t w + q_v x t) + v
A3D_HD V3 rotate(const Quat& q, V3 v) {
  V3 qv{q.i, q.j, q.k};
  V3 t = cross(qv, v) * 2.0f;
  V3 c = cross(qv, t);
  return (t * q.w + c) + v;
}
A3D_HD V3 transform_vector(const Pose& p, V3 v) { return rotate(p.q, v) + p.t; }
A3D_HD V3 transform_normal(const Pose& p, V3 v) { return rotate(p.q, v); }

A3D_HD Quat qmul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.i * b.i - a.j * b.j - a.k * b.k;
  r.i = a.w * b.i + a.i * b.w + a.j * b.k - a.k * b.j;
  r.j = a.w * b.j - a.i * b.k + a.j * b.w + a.k * b.i;
  r.k = a.w * b.k + a.i * b.j - a.j * b.i + a.k * b.w;
  return r;
}
A3D_HD Pose compose(const Pose& a, const Pose& b) {
  Pose r;
  r.t = a.t + rotate(a.q, b.t);
  r.q = qmul(a.q, b.q);
  return r;
}

// Unit::new_normalize on the 4-vector (i, j, k, w): norm^2 = (i^2 + k^2) + (j^2 + w^2)
A3D_HD Quat qnormalize(Quat q) {
  float a = q.i * q.i, b = q.j * q.j, c = q.k * q.k, d = q.w * q.w;
  a += c;
  b += d;
  float n = sqrtf(a + b);
  return {q.i / n, q.j / n, q.k / n, q.w / n};
}

// f32 sin/cos as the reference calls them (f32::sin / f32::cos).  The device versions are accurate to
// ~1 ulp, the reference's libm to <1 ulp; a 1-ulp difference here moves a pose by ~1e-9.
A3D_HD float sin_f32(float x) { return sinf(x); }
A3D_HD float cos_f32(float x) { return cosf(x); }

// The four transcendental values Transform::exp needs: sin and cos of theta / 2 and of theta, theta = |omega|.
struct Se3Trig {
  float theta, sin_half, cos_half, sin_theta, cos_theta;
};
A3D_HD float se3_theta(const float u[6]) { return sqrtf(norm_squared(V3{u[3], u[4], u[5]})); }

// Transform::exp(&LieGroup::Se3(u)), u = [rho, omega], with the transcendental values supplied by the caller
// (the iteration tail evaluates sin/cos of theta/2 and theta on two lanes at once).
A3D_HD Pose exp_se3_trig(const float u[6], const Se3Trig& tg) {
  const float EPS = 1e-8f;
  V3 omega{u[3], u[4], u[5]};
  float theta_sq0 = norm_squared(omega);
  float theta, imag, real;
  if (theta_sq0 < EPS * EPS) {
    float po4 = theta_sq0 * theta_sq0;
    theta = 0.0f;
    imag = 0.5f - (1.0f / 48.0f) * theta_sq0 + (1.0f / 3840.0f) * po4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq0 + (1.0f / 384.0f) * po4;
  } else {
    theta = tg.theta;
    imag = tg.sin_half / theta;
    real = tg.cos_half;
  }
  Quat q = qnormalize(Quat{imag * omega.x, imag * omega.y, imag * omega.z, real});
  float theta_sq = theta * theta;
  // V = I + a W + b W^2, W = [omega]x ; translation = V rho (column-axpy order)
  float W[3][3] = {{0.f, -omega.z, omega.y}, {omega.z, 0.f, -omega.x}, {-omega.y, omega.x, 0.f}};
  float V[3][3];
  if (theta_sq < EPS) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) V[r][c] = (r == c ? 1.0f : 0.0f) + W[r][c] * 0.5f;
  } else {
    float W2[3][3];
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) {
        float acc = W[r][0] * W[0][c];
        acc = W[r][1] * W[1][c] + acc;
        acc = W[r][2] * W[2][c] + acc;
        W2[r][c] = acc;
      }
    float a = (1.0f - tg.cos_theta) / theta_sq;
    float b = (theta - tg.sin_theta) / (theta_sq * theta);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) V[r][c] = ((r == c ? 1.0f : 0.0f) + W[r][c] * a) + W2[r][c] * b;
  }
  Pose p;
  float rho[3] = {u[0], u[1], u[2]}, t[3];
  for (int r = 0; r < 3; ++r) {
    float acc = V[r][0] * rho[0];
    acc = V[r][1] * rho[1] + acc;
    acc = V[r][2] * rho[2] + acc;
    t[r] = acc;
  }
  p.t = {t[0], t[1], t[2]};
  p.q = q;
  return p;
}

A3D_HD Pose exp_se3(const float u[6]) {
  Se3Trig tg;
  tg.theta = se3_theta(u);
  const float half = 0.5f * tg.theta;
  tg.sin_half = sin_f32(half), tg.cos_half = cos_f32(half);
  tg.sin_theta = sin_f32(tg.theta), tg.cos_theta = cos_f32(tg.theta);
  return exp_se3_trig(u, tg);
}

// Isometry3 -> Matrix4 (row-major), UnitQuaternion::to_rotation_matrix
A3D_HD void pose_to_matrix(const Pose& p, float m[16]) {
  const float i = p.q.i, j = p.q.j, k = p.q.k, w = p.q.w;
  float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
  float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f, ik = i * k * 2.0f;
  float jk = j * k * 2.0f, wi = w * i * 2.0f;
  m[0] = ww + ii - jj - kk, m[1] = ij - wk, m[2] = wj + ik, m[3] = p.t.x;
  m[4] = wk + ij, m[5] = ww - ii + jj - kk, m[6] = jk - wi, m[7] = p.t.y;
  m[8] = ik - wj, m[9] = wi + jk, m[10] = ww - ii - jj + kk, m[11] = p.t.z;
  m[12] = 0.f, m[13] = 0.f, m[14] = 0.f, m[15] = 1.f;
}

// GaussNewton::solve (src/optim/gaussnewton.rs:84-93): f64 Cholesky (nalgebra's left-looking
// column form) + forward / adjoint substitution.  H is the full symmetric 6x6.  false == None.
A3D_HD bool gn_solve6(const float H[36], const float g[6], float out[6]) {
  double L[6][6], b[6];
  for (int r = 0; r < 6; ++r) {
    for (int c = 0; c < 6; ++c) L[r][c] = (double)H[r * 6 + c];
    b[r] = (double)g[r];
  }
  for (int j = 0; j < 6; ++j) {
    for (int k = 0; k < j; ++k) {
      double factor = -L[j][k];
      for (int r = j; r < 6; ++r) L[r][j] = factor * L[r][k] + L[r][j];
    }
    double diag = L[j][j];
    if (diag == 0.0) return false;
    if (!(diag >= 0.0)) return false;
    double denom = sqrt(diag);
    L[j][j] = denom;
    for (int r = j + 1; r < 6; ++r) L[r][j] /= denom;
  }
  for (int i = 0; i < 6; ++i) {
    double coeff = b[i] / L[i][i];
    b[i] = coeff;
    for (int r = i + 1; r < 6; ++r) b[r] = -coeff * L[r][i] + b[r];
  }
  for (int i = 5; i >= 0; --i) {
    double d = 0.0;
    for (int r = i + 1; r < 6; ++r) d += L[r][i] * b[r];
    b[i] = (b[i] - d) / L[i][i];
  }
  for (int i = 0; i < 6; ++i) out[i] = (float)b[i];
  return true;
}

// Index of (r, c), r <= c, in the packed upper triangle of a symmetric 6x6 (21 entries).
A3D_HD constexpr int tri6(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

#if defined(__HIPCC__)
// ---- IEEE division with a shared reciprocal (device only) ---------------------------------------------------
// Several quotients over ONE denominator (the projection's x / z and y / z in the ICP kernels, a normal's three
// components over its length, a back-projection's division by fx).  a / z below is the arithmetic of the compiler's own
// correctly rounded f32 division (reciprocal estimate, one Newton step, quotient, two fma corrections) minus its
// range scaling, with the refined reciprocal computed once per denominator.  It returns the correctly rounded quotient for operands in `div_fast_ok` range (checked on
// 2e8 random pairs against IEEE division, and on the device by a3d_selftest_division); anything else takes
// the plain `/`.
struct DivBy {
  float negz, y;
};
__device__ __forceinline__ DivBy div_prepare(float z) {
  const float r = __builtin_amdgcn_rcpf(z);
  const float e = __builtin_fmaf(-z, r, 1.0f);
  return {-z, __builtin_fmaf(e, r, r)};
}
__device__ __forceinline__ float div_by(float a, const DivBy d) {
  const float q = a * d.y;
  const float r1 = __builtin_fmaf(d.negz, q, a);
  const float q1 = __builtin_fmaf(r1, d.y, q);
  const float r2 = __builtin_fmaf(d.negz, q1, a);
  return __builtin_fmaf(r2, d.y, q1);
}
// (bitwise on purpose: short-circuit forms compile to a chain of exec-mask branches in the pixel loop)
__device__ __forceinline__ bool div_den_ok(float z) { return (fabsf(z) > 1e-9f) & (fabsf(z) < 1e9f); }
__device__ __forceinline__ bool div_num_ok(float a) { return (a == 0.0f) | ((fabsf(a) > 1e-20f) & (fabsf(a) < 1e9f)); }


// The per-pixel body of RangeImage::compute_normals (src/range_image/structure.rs:207-257) as the kernels evaluate it:
// the same decisions and the same bits as normal_from_neighbours below (the reference's operations in the reference's
// order), with the two ratio tests and the three quotients n / |n| rewritten so that they cost a third of the
// instructions (the stand-alone stencil is VALU-issue bound, not byte-bound):
//  * `ld / rd < 4 && ld / rd > 1/4` == `ld < 4 rd && 4 ld > rd` for every pair of f32 (4 x is exact; a quotient below
//    4 is at most pred(4) = 4 (1 - 2^-24), which is representable, so round-to-nearest cannot lift it to 4; a quotient
//    above 1/4 is at least (1/4) succ(rd) / rd > the midpoint of 1/4 and succ(1/4), so it cannot be rounded down to 1/4;
//    rd == 0, NaN, inf, overflow and underflow give `false` / the same truth value on both sides);
//  * n / |n|: three IEEE quotients through ONE refined reciprocal (div_prepare / div_by: bit-identical to `/` inside
//    their operand range, checked on the device by a3d_selftest_division), plain `/` for the rare lanes outside it.
__device__ __forceinline__ V3 normal_from_neighbours_dev(V3 center, V3 left, V3 right, V3 top, V3 bottom) {
  const float ld = norm_squared(left - center), rd = norm_squared(right - center);
  V3 left_to_right;
  if ((ld < 4.0f * rd) & (4.0f * ld > rd))
    left_to_right = right - left;
  else if (ld < rd)
    left_to_right = center - left;
  else
    left_to_right = right - center;
  const float bd = norm_squared(bottom - center), td = norm_squared(top - center);
  V3 bottom_to_top;
  if ((bd < 4.0f * td) & (4.0f * bd > td))
    bottom_to_top = top - bottom;
  else if (bd < td)
    bottom_to_top = center - bottom;
  else
    bottom_to_top = top - center;
  const V3 n = cross(left_to_right, bottom_to_top);
  const float mag = sqrtf(norm_squared(n));
  V3 out{0.f, 0.f, 0.f};
  const bool keep = mag > 1e-6f;
  const bool fast = div_den_ok(mag) & div_num_ok(n.x) & div_num_ok(n.y) & div_num_ok(n.z);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(keep & !fast) != 0ull, 0)) {
    if (keep) out = n / mag;
  } else if (keep) {
    const DivBy d = div_prepare(mag);
    // (a zero numerator keeps its sign: -0 / |n| = -0, which the fused corrections of div_by would turn into +0)
    out = V3{n.x == 0.0f ? n.x : div_by(n.x, d), n.y == 0.0f ? n.y : div_by(n.y, d), n.z == 0.0f ? n.z : div_by(n.z, d)};
  }
  return out;
}

#endif  // __HIPCC__

// The per-pixel body of RangeImage::compute_normals (src/range_image/structure.rs:207-257): neighbours that are
// out of range or masked out arrive as (0,0,0); ratio comparisons with NaN / inf are false like the reference's.
A3D_HD V3 normal_from_neighbours(V3 center, V3 left, V3 right, V3 top, V3 bottom) {
  const float thr_sq = 2.0f * 2.0f;
  float ld = norm_squared(left - center), rd = norm_squared(right - center);
  float lr_ratio = ld / rd;
  V3 left_to_right;
  if (lr_ratio < thr_sq && lr_ratio > 1.0f / thr_sq)
    left_to_right = right - left;
  else if (ld < rd)
    left_to_right = center - left;
  else
    left_to_right = right - center;
  float bd = norm_squared(bottom - center), td = norm_squared(top - center);
  float bt_ratio = bd / td;
  V3 bottom_to_top;
  if (bt_ratio < thr_sq && bt_ratio > 1.0f / thr_sq)
    bottom_to_top = top - bottom;
  else if (bd < td)
    bottom_to_top = center - bottom;
  else
    bottom_to_top = top - center;
  V3 n = cross(left_to_right, bottom_to_top);
  float mag = sqrtf(norm_squared(n));
  V3 out{0.f, 0.f, 0.f};
  if (mag > 1e-6f) out = n / mag;
  return out;
}

}  // namespace a3d
