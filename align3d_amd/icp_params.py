"""IcpParams / MsIcpParams (src/icp/icp_params.rs:8-134), same fields, defaults and builder API."""
import ctypes as C
import dataclasses

from . import _abi


def _default_c():
    p = _abi.IcpParamsC()
    _abi.load_library().a3d_icp_params_default(C.byref(p))
    return p


@dataclasses.dataclass
class IcpParams:
    max_iterations: int = None
    weight: float = None
    color_weight: float = None
    max_point_to_plane_distance: float = None
    max_distance: float = None
    max_normal_angle: float = None
    max_color_distance: float = None

    def __post_init__(self):
        # IcpParams::default() (icp_params.rs:33-43): the values come from the library so that the f32
        # constants (18 degrees in radians) are the ones the kernels see.
        d = _default_c()
        for f in dataclasses.fields(self):
            if getattr(self, f.name) is None:
                setattr(self, f.name, getattr(d, f.name))

    @staticmethod
    def default():
        return IcpParams()

    def to_c(self):
        return _abi.IcpParamsC(
            int(self.max_iterations),
            self.weight,
            self.color_weight,
            self.max_point_to_plane_distance,
            self.max_distance,
            self.max_normal_angle,
            self.max_color_distance,
        )

    @staticmethod
    def from_c(c):
        return IcpParams(
            c.max_iterations,
            c.weight,
            c.color_weight,
            c.max_point_to_plane_distance,
            c.max_distance,
            c.max_normal_angle,
            c.max_color_distance,
        )


class MsIcpParams:
    """Per-level parameters; index 0 is the finest level, which runs last (icp_params.rs:127)."""

    def __init__(self, pyramid):
        self.pyramid = list(pyramid)

    @staticmethod
    def new(pyramid):
        return MsIcpParams(pyramid)

    @staticmethod
    def repeat(levels, params):
        return MsIcpParams([dataclasses.replace(params) for _ in range(levels)])

    def customize(self, f):
        for i, p in enumerate(self.pyramid):
            f(i, p)
        return self

    @staticmethod
    def default():
        """MsIcpParams::default() (icp_params.rs:112-133)."""
        arr = (_abi.IcpParamsC * 3)()
        _abi.load_library().a3d_ms_icp_params_default(arr)
        return MsIcpParams([IcpParams.from_c(arr[i]) for i in range(3)])

    def len(self):
        return len(self.pyramid)

    __len__ = len

    def is_empty(self):
        return not self.pyramid

    def iter(self):
        return iter(self.pyramid)

    __iter__ = iter

    def __getitem__(self, i):
        return self.pyramid[i]

    def __setitem__(self, i, v):
        self.pyramid[i] = v

    def to_c_array(self):
        arr = (_abi.IcpParamsC * max(1, len(self.pyramid)))()
        for i, p in enumerate(self.pyramid):
            arr[i] = p.to_c()
        return arr
