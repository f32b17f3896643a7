"""MI355X-native point-cloud -> framebuffer projector (hot path of RTRenderer's ProjectCloud).

Host-side mirror of the reference interface over the C ABI of include/rtr.h; the HIP
extension (lib/librtr_hip.so, built from csrc/) is the only compute implementation.
"""
from . import _lib
from ._lib import RtrError, RtrParams, build, LIB_PATH, SYMBOLS, EMPTY_DEPTH
from .camera import CameraCalibration, compose_projection, benchmark_calibration, orbit_pose, orbit_projection
from .projector import Projector, ProjectCloud, DeviceBuffer
from .sharded import ShardedProjector, shard_range
from . import formats, sharded

__all__ = ["RtrError", "RtrParams", "build", "LIB_PATH", "SYMBOLS", "EMPTY_DEPTH", "CameraCalibration",
           "compose_projection", "benchmark_calibration", "orbit_pose", "orbit_projection", "Projector",
           "ProjectCloud", "DeviceBuffer", "ShardedProjector", "shard_range"]
