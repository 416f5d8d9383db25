"""Dataset readers either side of the hot path: TUM RGB-D (src/io/dataset/tum.rs), IndoorLidar
(indoor_lidar.rs), SubsetDataset (core.rs).  The reference's own TUM test is #[ignore]d (it needs the downloaded
dataset), so the cases here are small directories written in those formats by the test itself."""
import numpy as np
import pytest
from PIL import Image

from align3d_amd import (DatasetError, IndoorLidarDataset, InvalidParameter, SubsetDataset, TumRgbdDataset,
                         load_dataset)
from align3d_amd.dataset import _tum_associate


def _write_images(directory, rgb_name, depth_name, seed):
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    depth = rng.integers(0, 40000, size=(480, 640), dtype=np.uint16)
    (directory / rgb_name).parent.mkdir(parents=True, exist_ok=True)
    (directory / depth_name).parent.mkdir(parents=True, exist_ok=True)
    Image.fromarray(rgb).save(directory / rgb_name)
    Image.fromarray(depth).save(directory / depth_name)
    return rgb, depth


def test_tum_associate_window_and_skips():
    a = [(0.00, "a0"), (0.05, "a1"), (0.10, "a2"), (0.20, "a3")]
    b = [(0.019, "b0"), (0.03, "b1"), (0.121, "b2"), (0.2199, "b3")]
    # a0~b0 (0.019 < 0.02); a1 vs b1: 0.02 apart is NOT < 0.02 -> b1 older, dropped; a1 vs b2 -> a1 dropped;
    # a2 vs b2: 0.021 -> a2 dropped; a3 vs b2 -> b2 dropped; a3~b3
    assert _tum_associate(a, b) == [(0.00, "a0", 0.019, "b0"), (0.20, "a3", 0.2199, "b3")]


def test_tum_dataset(tmp_path):
    imgs = [_write_images(tmp_path, f"rgb/{i}.png", f"depth/{i}.png", i) for i in range(3)]
    (tmp_path / "rgb.txt").write_text("# color images\n# timestamp filename\n"
                                      "1.000 rgb/0.png\n1.033 rgb/1.png\n1.500 rgb/unmatched.png\n1.066 rgb/2.png\n")
    (tmp_path / "depth.txt").write_text("# depth maps\n1.010 depth/0.png\n1.040 depth/1.png\n1.070\tdepth/2.png\n")
    # third pose has a non-unit quaternion: Transform::new normalises it
    (tmp_path / "groundtruth.txt").write_text("# timestamp tx ty tz qx qy qz qw\n"
                                              "0.5 9 9 9 0 0 0 1\n"
                                              "1.005 0.1 0.2 0.3 0 0 0 1\n"
                                              "1.045 0.4 0.5 0.6 0 0.7071068 0 0.7071068\n"
                                              "1.075 1 2 3 0 0 2 0\n")
    ds = TumRgbdDataset.load(str(tmp_path))
    # 1.500 is skipped only when it becomes the older entry: after (1.040,1.033) the cursors sit at depth 1.070 vs
    # rgb 1.500 -> depth is older and is dropped, so the last pair never forms (the reference's two-cursor walk)
    assert ds.len() == 2 and not ds.is_empty()
    assert ds.rgb_images == ["rgb/0.png", "rgb/1.png"] and ds.depth_images == ["depth/0.png", "depth/1.png"]
    cam, depth, rgb, scale = ds.get(1)
    assert (cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height) == (525.0, 525.0, 319.5, 239.5, 640, 480)
    assert scale == 1.0 / 5000.0
    assert np.array_equal(rgb, imgs[1][0]) and np.array_equal(depth, imgs[1][1]) and depth.dtype == np.uint16
    traj = ds.trajectory()
    assert traj.len() == 3  # depth-pose association is independent of the depth-colour one
    assert traj.times == [float(np.float32(t)) for t in (1.005, 1.045, 1.075)]
    assert np.allclose(traj[0].t, [0.1, 0.2, 0.3]) and np.allclose(traj[0].q, [0, 0, 0, 1])
    assert np.allclose(traj[1].q, [0, 0.70710678, 0, 0.70710678], atol=1e-7)
    assert np.allclose(traj[2].q, [0, 0, 1, 0]) and np.allclose(traj[2].t, [1, 2, 3])
    assert load_dataset("tum", str(tmp_path)).len() == 2


def test_tum_errors(tmp_path):
    with pytest.raises(DatasetError) as e:
        TumRgbdDataset.load(str(tmp_path))
    assert e.value.kind == "Io"
    (tmp_path / "rgb.txt").write_text("1.0  rgb/double_space.png\n\n")  # empty line -> parse panic in the reference
    with pytest.raises(DatasetError) as e:
        TumRgbdDataset.load(str(tmp_path))
    assert e.value.kind == "Parser"
    with pytest.raises(InvalidParameter):
        load_dataset("kitti", str(tmp_path))


def _il_log(poses):
    out = []
    for n, m in enumerate(poses):
        out.append(f"{n} {n} {n + 1}")
        out += [" ".join(f"{v:.8f}" for v in row) for row in m]
    return "\n".join(out) + "\n\n"


def test_indoor_lidar_dataset(tmp_path):
    base = tmp_path / "apartment"
    base.mkdir()
    depths = []
    for i in (2, 0, 1):  # written out of order: the reader sorts by name
        rng = np.random.default_rng(i)
        rgb = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
        depth = rng.integers(0, 40000, size=(480, 640), dtype=np.uint16)
        (base / "image").mkdir(exist_ok=True)
        (base / "depth").mkdir(exist_ok=True)
        Image.fromarray(rgb).save(base / "image" / f"{i:06d}.jpg")
        Image.fromarray(depth).save(base / "depth" / f"{i:06d}.png")
        depths.append((i, depth))
    c, s = np.cos(0.3), np.sin(0.3)
    poses = [np.eye(4), np.array([[c, -s, 0, 1.0], [s, c, 0, 2.0], [0, 0, 1, 3.0], [0, 0, 0, 1]]),
             np.array([[1, 0, 0, -1.0], [0, c, -s, 0.5], [0, s, c, 0.25], [0, 0, 0, 1]])]
    (base / "apartment.log").write_text(_il_log(poses))
    ds = IndoorLidarDataset.load(str(base) + "/")  # trailing slash: file_stem is still "apartment"
    assert ds.len() == 3
    cam, depth, rgb, scale = ds.get(2)
    assert scale == 0.001 and (cam.fx, cam.cx, cam.cy) == (525.0, 319.5, 239.5)
    assert np.array_equal(depth, dict(depths)[2]) and rgb.shape == (480, 640, 3) and rgb.dtype == np.uint8
    traj = ds.trajectory()
    assert traj.len() == 3 and traj.times == [0.0, 1.0, 2.0]
    for t, m in zip(traj.camera_to_world, poses):
        assert np.allclose(t.matrix(), m, atol=1e-6)
    sub = SubsetDataset.new(ds, [2, 0])
    assert sub.len() == 2 and np.array_equal(sub.get(0)[1], dict(depths)[2])
    st = sub.trajectory()
    assert st.times == [0.0, 1.0] and np.allclose(st[0].matrix(), poses[2], atol=1e-6)
    assert load_dataset("ilrgbd", str(base)).len() == 3


def test_indoor_lidar_errors(tmp_path):
    base = tmp_path / "loft"
    (base / "image").mkdir(parents=True)
    (base / "depth").mkdir()
    Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save(base / "image" / "0.jpg")
    with pytest.raises(DatasetError) as e:
        IndoorLidarDataset.load(str(base))
    assert e.value.kind == "Parser" and "do not match" in str(e.value)
    Image.fromarray(np.zeros((4, 4), np.uint16)).save(base / "depth" / "0.png")
    with pytest.raises(DatasetError) as e:  # no loft.log
        IndoorLidarDataset.load(str(base))
    assert e.value.kind == "Io"
