"""raytrace_amd: MI355X-native per-pixel ray / kd-tree traversal and shading core for
RayTrace-format scenes.  The product is the two C-ABI libraries under raytrace_amd/lib
(sources in raytrace_amd/csrc, headers in include/); this package is the thin host-side
mirror of the reference's scene model and offload boundary over them."""
from . import capi, scenes  # noqa: F401
from .scene import DeviceScene, HostScene, PreparedScene, debug_fp64, device_count, device_info, tonemap, write_bmp  # noqa: F401
