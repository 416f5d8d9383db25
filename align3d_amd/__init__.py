"""align3d_amd — MI355X-native implementation of align3d's ICP hot path.

Host-side mirror of the reference's public API for that path (same names and argument meaning:
IcpParams, MsIcpParams, ImageIcp, MultiscaleAlign, Icp, R3dTree, RangeImage, BilateralFilter,
Transform) over the C ABI of libalign3d_hip.so (include/align3d_hip.h).  All compute runs in
hand-written HIP kernels for gfx950; there is no CPU fallback."""
import os as _os

# Streams beyond GPU_MAX_HW_QUEUES (HIP default: 4) share hardware queues and run one after the other; a batch
# uses three, a context has four, an aligner + builder pair eight, PyTorch / RCCL in the same process their own.
# Must be set before the HIP runtime initialises, which importing this package does not do yet.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    import sys as _sys

    _torch = _sys.modules.get("torch")
    try:
        _late = bool(_torch is not None and _torch.cuda.is_initialized())
    except Exception:
        _late = False
    if _late:  # the runtime has read its settings already: say so instead of silently losing a third of the throughput
        import warnings as _warnings

        _warnings.warn("align3d_amd: HIP was initialised before this import, so GPU_MAX_HW_QUEUES=16 cannot be applied; "
                       "with the default of 4 hardware queues the pair groups of a batch alignment share queues and "
                       "run one after the other (about -35 %).  Export GPU_MAX_HW_QUEUES=16 before the process starts.")
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"

from ._abi import A3dError, InvalidParameter, load_library  # noqa: F401
from .bilateral import BilateralFilter  # noqa: F401
from .context import Context  # noqa: F401
from .icp import DevicePointCloud, Icp, ImageIcp, MultiscaleAlign, MultiscaleAlignBatch, PointCloud  # noqa: F401
from .icp_params import IcpParams, MsIcpParams  # noqa: F401
from .kdtree import R3dTree  # noqa: F401
from .multi import MultiContext, MultiscaleAlignMultiBatch, device_count  # noqa: F401
from .range_image import (CameraIntrinsics, DeviceRangeImage, RangeImage, RangeImageBuilder,  # noqa: F401
                          compute_normals_batch)
from .transform import Transform  # noqa: F401
from .dataset import (DatasetError, IndoorLidarDataset, SlamTbDataset, SubsetDataset, SyntheticDataset,  # noqa: F401
                      TumRgbdDataset, load_dataset)
from .odometry import run_odometry, run_odometry_batched  # noqa: F401
from .trajectory import Trajectory, TrajectoryBuilder, TransformMetrics  # noqa: F401
