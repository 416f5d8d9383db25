"""RgbdDataset readers the odometry harness needs: the SlamTb layout the reference's tests use
(src/io/dataset/slamtb.rs:61-154: frames.json + PNGs), the TUM RGB-D and IndoorLidar (IL-RGBD) layouts the
reference's odometry example takes (src/io/dataset/tum.rs, indoor_lidar.rs), SubsetDataset (core.rs:64-93)
and the seeded synthetic stream.  `get(i)` returns (camera, depth u16 [h][w], rgb u8 [h][w][3], depth_scale):
the arguments of RangeImageBuilder.build / build_device."""
import glob as _glob
import json
import os
import re

import numpy as np

from . import synth
from ._abi import InvalidParameter
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


class DatasetError(Exception):
    """DatasetError::{Io, Parser, Image} (src/io/dataset/core.rs:8-13); `kind` names the variant."""

    def __init__(self, kind, text):
        super().__init__(f"{kind} error: {text}")
        self.kind = kind


def _read_images(rgb_path, depth_path):
    from PIL import Image

    try:
        rgb = np.array(Image.open(rgb_path).convert("RGB"), np.uint8)  # into_rgb8
        depth = np.array(Image.open(depth_path))  # into_luma16
    except OSError as e:
        raise DatasetError("Image", str(e)) from e
    if depth.dtype == np.uint8:  # an 8-bit image widens as image-rs does: v * 257
        depth = depth.astype(np.uint16) * 257
    return np.ascontiguousarray(depth.astype(np.uint16)), rgb


# The Kinect intrinsics both readers hard-code (tum.rs:170-177, indoor_lidar.rs:108-115).
def _kinect_camera():
    return CameraIntrinsics(525.0, 525.0, 319.5, 239.5, 640, 480)


def _tum_read_file_list(path):
    """tum.rs:22-40: '#' lines skipped; tokens split at each single ',', TAB or ' ' (so a doubled separator
    yields an empty second token, as in the reference); (f64 timestamp, name)."""
    out = []
    try:
        with open(path) as f:
            for line in f.read().splitlines():
                if line.strip().startswith("#"):
                    continue
                tokens = re.split("[,\t ]", line)
                try:
                    out.append((float(tokens[0].strip()), tokens[1].strip()))
                except (ValueError, IndexError) as e:  # parse::<f64>().unwrap() / tokens[1] panic
                    raise DatasetError("Parser", f"{path}: bad line {line!r}") from e
    except OSError as e:
        raise DatasetError("Io", str(e)) from e
    return out


def _tum_associate(first, second):
    """tum.rs:42-68: two-cursor walk; a pair matches when |t1 - t2| < 0.02 s, otherwise the older entry is
    dropped.  Returns (t1, v1, t2, v2) tuples."""
    i = j = 0
    out = []
    while i < len(first) and j < len(second):
        t1, v1 = first[i]
        t2, v2 = second[j]
        if abs(t1 - t2) < 0.02:
            out.append((t1, v1, t2, v2))
            i += 1
            j += 1
        elif t1 < t2:
            i += 1
        else:
            j += 1
    return out


def _tum_load_trajectory(path):
    """tum.rs:70-100: 'timestamp tx ty tz qx qy qz qw' -> Transform::new(xyz, Quaternion::new(qw, qx, qy, qz)),
    components cast f64 -> f32 first; the quaternion is normalised (UnitQuaternion::from_quaternion)."""
    out = []
    try:
        with open(path) as f:
            for line in f.read().splitlines():
                if line.strip().startswith("#"):
                    continue
                try:
                    tok = [float(t) for t in line.split()]
                    q = np.array([tok[4], tok[5], tok[6], tok[7]], np.float32)  # i, j, k, w
                    t = np.array(tok[1:4], np.float32)
                except (ValueError, IndexError) as e:
                    raise DatasetError("Parser", f"{path}: bad line {line!r}") from e
                # Quaternion::norm (f32): sqrt of the 4-term sum of squares
                q = q / np.sqrt(np.float32(np.sum(q * q, dtype=np.float32)))
                out.append((tok[0], Transform(tuple(float(x) for x in t), tuple(float(x) for x in q))))
    except OSError as e:
        raise DatasetError("Io", str(e)) from e
    return out


class TumRgbdDataset:
    """TumRgbdDataset (src/io/dataset/tum.rs:15-178): rgb.txt / depth.txt / groundtruth.txt, depth frames
    associated with colour frames and (separately) with ground-truth poses inside a 0.02 s window,
    depth scale 1/5000, fixed 525/525/319.5/239.5 intrinsics.  As in the reference the frame list follows the
    depth-colour association and the trajectory the depth-pose association, each indexed from 0."""

    def __init__(self, base_dir):
        self.base_dir = base_dir
        rgb_files = _tum_read_file_list(os.path.join(base_dir, "rgb.txt"))
        depth_files = _tum_read_file_list(os.path.join(base_dir, "depth.txt"))
        assoc = _tum_associate(depth_files, rgb_files)
        self.rgb_images = [e[3] for e in assoc]
        self.depth_images = [e[1] for e in assoc]
        poses = _tum_load_trajectory(os.path.join(base_dir, "groundtruth.txt"))
        self._trajectory = Trajectory()
        for e in _tum_associate(depth_files, poses):
            self._trajectory.push(e[3], float(np.float32(e[2])))  # the pose's own timestamp, as f32

    @staticmethod
    def load(base_dir):
        return TumRgbdDataset(base_dir)

    def len(self):
        return len(self.rgb_images)

    __len__ = len

    def is_empty(self):
        return self.len() == 0

    def camera(self, index):
        self._trajectory[index]  # the reference indexes the trajectory here and panics when it is shorter
        return _kinect_camera()

    def depth_scale(self, index):
        return 1.0 / 5000.0

    def get(self, index):
        depth, rgb = _read_images(os.path.join(self.base_dir, self.rgb_images[index]),
                                  os.path.join(self.base_dir, self.depth_images[index]))
        return self.camera(index), depth, rgb, self.depth_scale(index)

    def trajectory(self):
        return self._trajectory


class IndoorLidarDataset:
    """IndoorLidarDataset (src/io/dataset/indoor_lidar.rs:19-118): image/*.jpg + depth/*.png in glob
    (alphabetical) order and `<dir stem>.log`, whose non-empty lines come in blocks of five: a header line,
    then the four rows of the camera-to-world matrix (parsed as f32).  Depth scale 0.001.
    JPEG decoding goes through PIL/libjpeg here and through image-rs there; the two decoders may differ by a
    level in places (unpinned, like the RGB pyramid blur)."""

    def __init__(self, base_dir):
        self.rgb_images = sorted(_glob.glob(os.path.join(_glob.escape(base_dir), "image", "*.jpg")))
        self.depth_images = sorted(_glob.glob(os.path.join(_glob.escape(base_dir), "depth", "*.png")))
        if len(self.rgb_images) != len(self.depth_images):
            raise DatasetError("Parser", "Number of RGB and depth images do not match")
        stem = os.path.splitext(os.path.basename(os.path.normpath(base_dir)))[0]  # Path::file_stem
        try:
            with open(os.path.join(base_dir, stem + ".log")) as f:
                lines = [ln.strip() for ln in f.read().splitlines()]
        except OSError as e:
            raise DatasetError("Io", str(e)) from e
        lines = [ln for ln in lines if ln]
        self._trajectory = Trajectory()
        for n in range(0, len(lines), 5):
            m = np.zeros((4, 4), np.float32)
            for i, row in enumerate(lines[n + 1:n + 5]):
                tok = row.split()
                if len(tok) > 4:
                    raise DatasetError("Parser", f"matrix row with {len(tok)} entries: {row!r}")
                try:
                    for j, t in enumerate(tok):
                        m[i, j] = np.float32(t)
                except ValueError as e:
                    raise DatasetError("Parser", f"bad matrix row {row!r}") from e
            self._trajectory.push(Transform.from_matrix4(m), float(n // 5))

    @staticmethod
    def load(base_dir):
        return IndoorLidarDataset(base_dir)

    def len(self):
        return len(self.rgb_images)

    __len__ = len

    def is_empty(self):
        return self.len() == 0

    def camera(self, index):
        self._trajectory[index]
        return _kinect_camera()

    def depth_scale(self, index):
        return 0.001

    def get(self, index):
        depth, rgb = _read_images(self.rgb_images[index], self.depth_images[index])
        return self.camera(index), depth, rgb, self.depth_scale(index)

    def trajectory(self):
        return self._trajectory


class SubsetDataset:
    """SubsetDataset (src/io/dataset/core.rs:64-93): a dataset seen through a list of indices; the
    trajectory is re-timed 0, 1, 2, ..."""

    def __init__(self, dataset, indices):
        self.dataset = dataset
        self.indices = list(indices)

    @staticmethod
    def new(dataset, indices):
        return SubsetDataset(dataset, indices)

    def len(self):
        return len(self.indices)

    __len__ = len

    def is_empty(self):
        return self.len() == 0

    def get(self, index):
        return self.dataset.get(self.indices[index])

    def camera(self, index):
        return self.dataset.camera(self.indices[index])

    def trajectory(self):
        orig = self.dataset.trajectory()
        if orig is None:
            return None
        t = Trajectory()
        for i, index in enumerate(self.indices):
            t.push(orig.camera_to_world[index], float(i))
        return t


def load_dataset(fmt, path):
    """examples/src/lib.rs `load_dataset(format, path)`: "ilrgbd" | "tum" (the reference's two), plus "slamtb"."""
    if fmt == "tum":
        return TumRgbdDataset.load(path)
    if fmt == "ilrgbd":
        return IndoorLidarDataset.load(path)
    if fmt == "slamtb":
        return SlamTbDataset.load(path)
    raise InvalidParameter(f"Invalid dataset format: {fmt}")
