"""RgbdDataset readers the odometry harness needs: the SlamTb layout the reference's tests use
(src/io/dataset/slamtb.rs:61-154: frames.json + PNGs) and the seeded synthetic stream."""
import json
import os

import numpy as np

from . import synth
from .range_image import CameraIntrinsics
from .trajectory import Trajectory
from .transform import Transform


class SlamTbDataset:
    def __init__(self, base_dir):
        self.base_dir = base_dir
        with open(os.path.join(base_dir, "frames.json")) as f:
            self.frames = json.load(f)["root"]

    @staticmethod
    def load(base_dir):
        return SlamTbDataset(base_dir)

    def len(self):
        return len(self.frames)

    __len__ = len

    def camera(self, index):
        info = self.frames[index]["info"]
        k = info["kcam"]["matrix"]
        w, h = info["kcam"]["image_size"]
        return CameraIntrinsics(k[0][0], k[1][1], k[0][2], k[1][2], w, h)

    def depth_scale(self, index):
        return self.frames[index]["info"]["depth_scale"]

    def get(self, index):
        """(camera, depth u16 [h][w], rgb u8 [h][w][3], depth_scale)"""
        from PIL import Image

        fr = self.frames[index]
        depth = np.array(Image.open(os.path.join(self.base_dir, fr["depth_image"]))).astype(np.uint16)
        rgb = np.array(Image.open(os.path.join(self.base_dir, fr["rgb_image"])).convert("RGB"), np.uint8)
        return self.camera(index), depth, rgb, self.depth_scale(index)

    def trajectory(self):
        t = Trajectory()
        for i, fr in enumerate(self.frames):
            m = fr["info"]["rt_cam"]["matrix"]
            t.push(Transform.from_matrix4(np.array(m)) if len(m) == 4 else Transform.eye(), float(i))
        return t


class SyntheticDataset:
    """n_frames of the analytic room along a seeded trajectory (align3d_amd.synth), with ground truth."""

    def __init__(self, seed, n_frames, width=640, height=480):
        self._frames, self._poses = synth.frame_stream(seed, n_frames, width, height)
        self._camera = synth.camera(width, height)

    def len(self):
        return len(self._frames)

    __len__ = len

    def get(self, index):
        d, rgb = self._frames[index]
        return self._camera, d, rgb, synth.DEPTH_SCALE

    def trajectory(self):
        t = Trajectory()
        for i, (R, tr) in enumerate(self._poses):
            m = np.eye(4)
            m[:3, :3] = R
            m[:3, 3] = tr
            t.push(Transform.from_matrix4(m), float(i))
        return t
