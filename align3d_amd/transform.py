"""Transform = Isometry3<f32> (src/transform.rs:18): translation + unit quaternion (i, j, k, w)."""
import numpy as np

from ._abi import PoseC


class Transform:
    def __init__(self, t=(0.0, 0.0, 0.0), q=(0.0, 0.0, 0.0, 1.0)):
        self.t = np.asarray(t, np.float32).copy()
        self.q = np.asarray(q, np.float32).copy()

    @staticmethod
    def eye():
        """Transform::eye (src/transform.rs:29-34)."""
        return Transform()

    @staticmethod
    def from_c(p):
        return Transform(p.t[:], p.q[:])

    def to_c(self):
        p = PoseC()
        p.t[:] = [float(x) for x in self.t]
        p.q[:] = [float(x) for x in self.q]
        return p

    # -- Isometry3 algebra in f32, nalgebra's formulas (src/transform.rs:138-153, :191, :205-220) -------------
    @staticmethod
    def _rotate(q, v):
        """UnitQuaternion * Vector3: t = 2 (q_v x v); v' = (t w + q_v x t) + v — every operation rounded to f32, in
        nalgebra's order (numpy f32 scalars: the odometry loop calls this once per frame, array temporaries cost 10x)."""
        f = np.float32
        qi, qj, qk, qw = f(q[0]), f(q[1]), f(q[2]), f(q[3])
        vx, vy, vz = f(v[0]), f(v[1]), f(v[2])
        two = f(2)
        tx, ty, tz = (qj * vz - qk * vy) * two, (qk * vx - qi * vz) * two, (qi * vy - qj * vx) * two
        cx, cy, cz = qj * tz - qk * ty, qk * tx - qi * tz, qi * ty - qj * tx
        return np.array([(tx * qw + cx) + vx, (ty * qw + cy) + vy, (tz * qw + cz) + vz], np.float32)

    def transform_vector(self, v):
        return (self._rotate(self.q, v) + self.t).astype(np.float32)

    def __mul__(self, rhs):
        """self * rhs (rhs applied first): t = t1 + R1 t2, q = q1 q2 (Hamilton product, not renormalised)."""
        f = np.float32
        a0, a1, a2, a3 = f(self.q[0]), f(self.q[1]), f(self.q[2]), f(self.q[3])
        b0, b1, b2, b3 = f(rhs.q[0]), f(rhs.q[1]), f(rhs.q[2]), f(rhs.q[3])
        q = (a3 * b0 + a0 * b3 + a1 * b2 - a2 * b1,
             a3 * b1 - a0 * b2 + a1 * b3 + a2 * b0,
             a3 * b2 + a0 * b1 - a1 * b0 + a2 * b3,
             a3 * b3 - a0 * b0 - a1 * b1 - a2 * b2)
        return Transform(self.t + self._rotate(self.q, rhs.t), q)

    def inverse(self):
        qi = np.array([-self.q[0], -self.q[1], -self.q[2], self.q[3]], np.float32)
        return Transform(-self._rotate(qi, self.t), qi)

    @staticmethod
    def from_matrix4(m):
        """Transform::from_matrix4 (src/transform.rs:112-118) for a rigid 4x4 (trace-based extraction)."""
        m = np.asarray(m, np.float64)
        r = m[:3, :3]
        tr = np.trace(r)
        if tr > 0:
            s = np.sqrt(tr + 1.0) * 2
            w, i, j, k = 0.25 * s, (r[2, 1] - r[1, 2]) / s, (r[0, 2] - r[2, 0]) / s, (r[1, 0] - r[0, 1]) / s
        elif r[0, 0] > r[1, 1] and r[0, 0] > r[2, 2]:
            s = np.sqrt(1.0 + r[0, 0] - r[1, 1] - r[2, 2]) * 2
            w, i, j, k = (r[2, 1] - r[1, 2]) / s, 0.25 * s, (r[0, 1] + r[1, 0]) / s, (r[0, 2] + r[2, 0]) / s
        elif r[1, 1] > r[2, 2]:
            s = np.sqrt(1.0 + r[1, 1] - r[0, 0] - r[2, 2]) * 2
            w, i, j, k = (r[0, 2] - r[2, 0]) / s, (r[0, 1] + r[1, 0]) / s, 0.25 * s, (r[1, 2] + r[2, 1]) / s
        else:
            s = np.sqrt(1.0 + r[2, 2] - r[0, 0] - r[1, 1]) * 2
            w, i, j, k = (r[1, 0] - r[0, 1]) / s, (r[0, 2] + r[2, 0]) / s, (r[1, 2] + r[2, 1]) / s, 0.25 * s
        q = np.array([i, j, k, w], np.float64)
        q /= np.linalg.norm(q)
        return Transform(m[:3, 3], q)

    def matrix(self):
        """4x4 homogeneous matrix (From<&Transform> for Matrix4, src/transform.rs:229-234)."""
        i, j, k, w = [np.float32(x) for x in self.q]
        two = np.float32(2)
        R = np.array(
            [
                [w * w + i * i - j * j - k * k, i * j * two - w * k * two, w * j * two + i * k * two],
                [w * k * two + i * j * two, w * w - i * i + j * j - k * k, j * k * two - w * i * two],
                [i * k * two - w * j * two, w * i * two + j * k * two, w * w - i * i - j * j + k * k],
            ],
            np.float32,
        )
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = R
        m[:3, 3] = self.t
        return m

    def angle(self):
        """Transform::angle (src/transform.rs:196-198): rotation angle in radians."""
        n = np.float32(np.sqrt(np.float32(np.sum(self.q[:3].astype(np.float32) ** 2))))
        return float(np.float32(2) * np.arctan2(n, np.abs(self.q[3])))

    def translation(self):
        return self.t.copy()

    def __repr__(self):
        return f"Transform(t={self.t.tolist()}, q_ijkw={self.q.tolist()})"
