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
