"""Fixture loading: the SlamTb-format sample frames the reference's own tests use
(tests/golden/rgbd/*, copied data files — MIT, otaviog/align3d) and seeded synthetic inputs."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class SlamTbSample:
    """Reader for the frames.json + PNG layout (src/io/dataset/slamtb.rs:61-154)."""

    def __init__(self, name):
        self.dir = os.path.join(GOLDEN, "rgbd", name)
        with open(os.path.join(self.dir, "frames.json")) as f:
            self.frames = json.load(f)["root"]
        self.ids = [int(fr["depth_image"].split("_")[1]) for fr in self.frames]

    def _frame(self, frame_id):
        return self.frames[self.ids.index(frame_id)]

    def intrinsics(self, frame_id):
        k = self._frame(frame_id)["info"]["kcam"]["matrix"]
        return k[0][0], k[1][1], k[0][2], k[1][2]

    def depth_scale(self, frame_id):
        return self._frame(frame_id)["info"]["depth_scale"]

    def load(self, frame_id):
        from PIL import Image

        fr = self._frame(frame_id)
        depth = np.array(Image.open(os.path.join(self.dir, fr["depth_image"])))
        assert depth.dtype in (np.uint16, np.int32), depth.dtype
        depth = depth.astype(np.uint16)
        rgb = np.array(Image.open(os.path.join(self.dir, fr["rgb_image"])).convert("RGB"), np.uint8)
        return depth, rgb

    def rt_cam(self, frame_id):
        return np.array(self._frame(frame_id)["info"]["rt_cam"]["matrix"], np.float32)


def splitmix64(seed, n):
    """Counter-based generator shared with the product's synthetic workloads (same constants)."""
    x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)).astype(np.uint64)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def uniform01(seed, n):
    """f32 uniform in [0,1) from the top 24 bits."""
    return ((splitmix64(seed, n) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(
        np.float32
    )
