"""Seeded synthetic RGB-D input for benchmarks and large-size tests: an analytic room
(five planes + three spheres, depth about 1.4-4.2 m) with a sinusoid texture, about 12 % of the pixels
invalid in contiguous holes (sample1 has 12.0 %), rendered from a camera that moves like the sample
sequences do (0.17-0.56 degrees and 2-6 mm per frame).  Everything derives from a counter-based
generator (splitmix64), so any process regenerates the same frames from the seed."""
import math

import numpy as np

from .range_image import CameraIntrinsics

SAMPLE_INTRINSICS = (544.4732666015625, 544.4732666015625, 320.0, 240.0)  # sample1 (slamtb.rs:166-169)
DEPTH_SCALE = 0.001


def splitmix64(seed, n):
    x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)).astype(np.uint64)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def uniform01(seed, n):
    """f64 uniforms in [0, 1) from the top 53 bits."""
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform01_f32(seed, n):
    """f32 uniforms in [0, 1) from the top 24 bits (kd-tree workloads: benches/bench_kdtree.rs shape)."""
    return ((splitmix64(seed, n) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def _rodrigues(axis, angle):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


class Scene:
    def __init__(self, seed):
        u = uniform01(seed * 7919 + 17, 64)
        self.planes = [  # (axis, offset): the plane  p[axis] == offset
            (1, 1.2 + 0.1 * u[0]),
            (1, -1.4 - 0.1 * u[1]),
            (0, -2.2 - 0.2 * u[2]),
            (0, 2.5 + 0.2 * u[3]),
            (2, 4.2 + 0.2 * u[4]),
        ]
        self.spheres = [
            (np.array([-0.9 + 0.2 * u[5], 0.6, 1.9 + 0.2 * u[6]]), 0.5),
            (np.array([-0.3 + 0.2 * u[7], 0.45, 2.9 + 0.2 * u[8]]), 0.55),
            (np.array([0.9 + 0.2 * u[9], -0.2, 3.1 + 0.2 * u[10]]), 0.45),
        ]
        self.tex_dir = (u[11:29].reshape(6, 3) - 0.5) * 2.0
        self.tex_freq = 4.0 + 22.0 * u[29:35]  # rad / m: wavelengths 0.24 - 1.6 m
        self.tex_phase = 2 * math.pi * u[35:41]
        self.hole_phase = 2 * math.pi * u[41:44]
        self.seed = seed

    def render(self, R, t, width=640, height=480, intr=SAMPLE_INTRINSICS, noise_seed=0, invalid_fraction=0.12):
        """Camera-to-world pose (R, t) -> (depth u16, rgb u8)."""
        fx, fy, cx, cy = intr
        us = (np.arange(width, dtype=np.float64) - cx) / fx
        vs = (np.arange(height, dtype=np.float64) - cy) / fy
        d_cam = np.stack(np.broadcast_arrays(us[None, :], vs[:, None], np.ones((height, width))), -1)
        d = d_cam @ R.T
        o = t
        best = np.full((height, width), np.inf)
        for axis, off in self.planes:
            with np.errstate(divide="ignore", invalid="ignore"):
                tt = (off - o[axis]) / d[..., axis]
            tt = np.where(tt > 1e-3, tt, np.inf)
            best = np.minimum(best, tt)
        for c, r in self.spheres:
            oc = o - c
            a = np.sum(d * d, -1)
            b = 2 * (d @ oc)
            cc = oc @ oc - r * r
            disc = b * b - 4 * a * cc
            with np.errstate(invalid="ignore"):
                tt = (-b - np.sqrt(disc)) / (2 * a)
            tt = np.where((disc > 0) & (tt > 1e-3), tt, np.inf)
            best = np.minimum(best, tt)
        hit = o + best[..., None] * d  # world point; camera depth == best because d_cam.z == 1
        tex = np.zeros((height, width))
        for k in range(6):
            tex += np.sin(self.tex_freq[k] * (hit @ self.tex_dir[k]) + self.tex_phase[k])
        base = 128.0 + 17.0 * tex
        noise = (uniform01(self.seed * 31 + noise_seed, width * height * 3).reshape(height, width, 3) - 0.5) * 8.0
        rgb = np.stack([base * 1.05 - 4, base, base * 0.9 + 9], -1) + noise
        rgb = np.clip(np.floor(rgb), 0, 255).astype(np.uint8)
        hole = (np.sin(3.1 * hit[..., 0] + self.hole_phase[0]) * np.sin(2.7 * hit[..., 1] + self.hole_phase[1])
                * np.sin(2.3 * hit[..., 2] + self.hole_phase[2]))
        thr = np.quantile(hole, 1.0 - invalid_fraction)
        depth = np.where(np.isfinite(best) & (hole <= thr), np.clip(np.round(best / DEPTH_SCALE), 0, 65535), 0)
        return depth.astype(np.uint16), rgb


def trajectory(seed, n_frames):
    """Camera-to-world poses: per step a rotation of 0.17-0.56 degrees about a seeded axis and a
    translation of 2-6 mm (the motion statistics of the reference's sample1 sequence)."""
    u = uniform01(seed * 104729 + 5, 8 * n_frames).reshape(n_frames, 8)
    R, t = np.eye(3), np.zeros(3)
    poses = [(R.copy(), t.copy())]
    for k in range(1, n_frames):
        axis = u[k, 0:3] - 0.5 + 1e-3
        ang = math.radians(0.17 + (0.56 - 0.17) * u[k, 3])
        dirv = u[k, 4:7] - 0.5 + 1e-3
        step = (0.002 + 0.004 * u[k, 7]) * dirv / np.linalg.norm(dirv)
        R = R @ _rodrigues(axis, ang)
        t = t + step
        poses.append((R.copy(), t.copy()))
    return poses


def relative_pose(pose_target, pose_source):
    """4x4 transform taking source-camera points into the target camera frame (what ICP estimates)."""
    Rt, tt = pose_target
    Rs, ts = pose_source
    m = np.eye(4)
    m[:3, :3] = Rt.T @ Rs
    m[:3, 3] = Rt.T @ (ts - tt)
    return m


def camera(width=640, height=480):
    fx, fy, cx, cy = SAMPLE_INTRINSICS
    return CameraIntrinsics(fx, fy, cx, cy, width, height)


def frame_stream(seed, n_frames, width=640, height=480, first=0, count=None):
    """Frames [first, first + count) (default: all) of the n_frames-long stream `seed`: (depth u16, rgb u8) of one
    scene along one trajectory + the camera-to-world poses of those frames.  A frame depends on (seed, its index)
    only, so ranks that render disjoint slices of one stream agree on every frame."""
    scene = Scene(seed)
    count = n_frames - first if count is None else count
    poses = trajectory(seed, n_frames)[first:first + count]
    frames = [scene.render(R, t, width, height, noise_seed=first + k) for k, (R, t) in enumerate(poses)]
    return frames, poses
