"""Trajectory, TrajectoryBuilder (src/trajectory.rs) and TransformMetrics (src/metrics.rs): the small
host-side pieces the odometry loop needs around the hot path."""
import math

import numpy as np

from ._abi import InvalidParameter
from .transform import Transform


class Trajectory:
    """Camera-to-world poses + timestamps (src/trajectory.rs:7-12)."""

    def __init__(self):
        self.camera_to_world = []
        self.times = []

    def push(self, camera_to_world, time):
        self.camera_to_world.append(camera_to_world)
        self.times.append(float(time))

    def len(self):
        return len(self.camera_to_world)

    __len__ = len

    def is_empty(self):
        return not self.camera_to_world

    def get_relative_transform(self, from_index, dest_index):
        """trajectory.rs:47-53: dest^-1 * from"""
        return self.camera_to_world[dest_index].inverse() * self.camera_to_world[from_index]

    def iter(self):
        return zip(self.camera_to_world, self.times)

    def first_frame_at_origin(self):
        """trajectory.rs:64-78"""
        out = Trajectory()
        if self.is_empty():
            return out
        first_inv = self.camera_to_world[0].inverse()
        for t, time in self.iter():
            out.push(first_inv * t, time)
        return out

    def slice(self, start, end):
        out = Trajectory()
        for t, time in list(self.iter())[start:end]:
            out.push(t, time)
        return out

    def last(self):
        return None if self.is_empty() else (self.camera_to_world[-1], self.times[-1])

    def __getitem__(self, i):
        return self.camera_to_world[i]


class TrajectoryBuilder:
    """src/trajectory.rs:129-182"""

    def __init__(self):
        self.trajectory = Trajectory()
        self._last = Transform.eye()
        self._last_time = 0.0

    @staticmethod
    def with_start(start_transform, start_time):
        b = TrajectoryBuilder()
        b.trajectory.push(start_transform, start_time)
        b._last = start_transform
        b._last_time = float(start_time)
        return b

    def accumulate(self, now_to_previous, timestamp=None):
        """trajectory.rs:164-168: last = now_to_previous * last (sic: left multiplication)."""
        self._last = now_to_previous * self._last
        self._last_time = float(timestamp) if timestamp is not None else self._last_time + 1.0
        self.trajectory.push(self._last, self._last_time)

    def build(self):
        return self.trajectory

    def current_camera_to_world(self):
        return None if self.trajectory.is_empty() else self.trajectory.last()[0]


class TransformMetrics:
    """src/metrics.rs:5-69"""

    def __init__(self, lfs=None, rhs=None):
        self.angle = 0.0
        self.translation = 0.0
        if lfs is not None:
            diff = lfs.inverse() * rhs
            self.angle = diff.angle()
            self.translation = float(np.sqrt(np.float32(np.sum(diff.t.astype(np.float32) ** 2))))

    @staticmethod
    def new(lfs, rhs):
        return TransformMetrics(lfs, rhs)

    @staticmethod
    def mean_trajectory_error(pred, gt):
        """metrics.rs:33-52"""
        if pred.len() != gt.len():
            raise InvalidParameter("Pred and GT trajectories have different lengths.")
        acc = TransformMetrics()
        for (p, _), (g, _) in zip(pred.iter(), gt.iter()):
            m = TransformMetrics.new(p, g)
            acc.angle += m.angle
            acc.translation += m.translation
        acc.angle /= pred.len()
        acc.translation /= pred.len()
        return acc

    def total(self):
        return self.angle + self.translation

    def __str__(self):
        return f"angle: {math.degrees(self.angle):.2f}°, translation: {self.translation:.5f}"
