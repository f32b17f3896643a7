"""ctypes binding of librtr_hip.so (the C ABI in include/rtr.h).

There is deliberately no fallback: if the HIP extension is missing or no GPU is
visible the calls fail loudly (RtrError / OSError).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librtr_hip.so")
if os.environ.get("RTR_LIB_VARIANT"):  # e.g. "xp": the timing-experiment build (`make -C csrc experiment`)
    LIB_PATH = os.path.join(_HERE, "lib", "librtr_hip_%s.so" % os.environ["RTR_LIB_VARIANT"])
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/rtr.h declares (tests check the library exports all of them)
SYMBOLS = [
    "rtr_abi_version", "rtr_create", "rtr_destroy", "rtr_last_error", "rtr_default_params", "rtr_set_params",
    "rtr_get_params", "rtr_set_stream", "rtr_synchronize", "rtr_upload_points", "rtr_generate_synthetic",
    "rtr_num_points", "rtr_download_points", "rtr_compose_projection", "rtr_set_resolution", "rtr_project",
    "rtr_project_filtered", "rtr_render", "rtr_clear", "rtr_min_depth_pass", "rtr_accumulate_pass", "rtr_resolve",
    "rtr_filter", "rtr_device_buffer", "rtr_download_buffer", "rtr_timing_enable", "rtr_timing_reset",
    "rtr_timing_get", "rtr_set_option", "rtr_stream_probe", "rtr_resolve_range", "rtr_reorder_points", "rtr_reset_stream",
    "rtr_device_count", "rtr_p2p_export", "rtr_p2p_open", "rtr_p2p_close", "rtr_p2p_min_depth", "rtr_p2p_sum_resolve", "rtr_p2p_status",
    "rtr_p2p_render", "rtr_frame_stats", "rtr_get_option", "rtr_host_output_buffers", "rtr_project_async", "rtr_wait",
    "rtr_p2p_render_owned",
]

RTR_OK, RTR_ERR_INVALID, RTR_ERR_HIP, RTR_ERR_NO_OUTPUT, RTR_ERR_UNSUPPORTED, RTR_ERR_INTERNAL = 0, -1, -2, -3, -4, -5
BUF_DEPTH, BUF_ACCUM, BUF_IMAGE, BUF_TENSOR, BUF_MASK, BUF_MINMAX = range(6)
K_CLEAR, K_MIN_DEPTH, K_ACCUMULATE, K_RESOLVE, K_FILTER, K_PROBE, K_TILE, K_BIN = range(8)
P2P_HANDLES_BYTES = 9 * 64  # sizeof(rtr_p2p_handles)
KERNEL_NAMES = ["clear", "min_depth", "accumulate", "resolve", "filter", "probe", "tile", "bin"]
SCENES = {"uniform_box": 0, "room_shell": 1}
EMPTY_DEPTH = 0x7F7FFFFF


class RtrParams(C.Structure):
    _fields_ = [("depth_window", C.c_float), ("filter_strength", C.c_float),
                ("gradient_threshold", C.c_float), ("levels", C.c_int32)]


class RtrError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("rtr error %d: %s" % (code, text))
        self.code = code


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the in-tree extension (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s"] + (["-B"] if force else [])
    subprocess.check_call(args)
    return LIB_PATH


ABI_VERSION = 2  # include/rtr.h RTR_ABI_VERSION
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("HIP extension %s is missing: run `python __graft_entry__.py` (build()) first; "
                      "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64, sz, i32 = C.c_void_p, C.c_uint64, C.c_size_t, C.c_int
    L.rtr_abi_version.restype = i32
    if L.rtr_abi_version() != ABI_VERSION:  # caller-allocated structs (P2P_HANDLES_BYTES) would be mis-sized
        raise OSError("%s has ABI version %d, this binding was written for %d: rebuild (`python __graft_entry__.py`)"
                      % (LIB_PATH, L.rtr_abi_version(), ABI_VERSION))
    L.rtr_device_count.argtypes = []
    L.rtr_create.argtypes = [C.POINTER(vp), i32]
    L.rtr_destroy.argtypes = [vp]
    L.rtr_last_error.argtypes = [vp]
    L.rtr_last_error.restype = C.c_char_p
    L.rtr_default_params.argtypes = [C.POINTER(RtrParams)]
    L.rtr_default_params.restype = None
    L.rtr_set_params.argtypes = [vp, C.POINTER(RtrParams)]
    L.rtr_get_params.argtypes = [vp, C.POINTER(RtrParams)]
    L.rtr_set_stream.argtypes = [vp, vp]
    L.rtr_reset_stream.argtypes = [vp]
    L.rtr_synchronize.argtypes = [vp]
    L.rtr_upload_points.argtypes = [vp, vp, sz, vp, sz, sz]
    L.rtr_generate_synthetic.argtypes = [vp, i32, u64, u64, u64, u64]
    L.rtr_num_points.argtypes = [vp, C.POINTER(u64)]
    L.rtr_download_points.argtypes = [vp, vp, vp, u64, u64]
    L.rtr_compose_projection.argtypes = [vp, vp, vp]
    L.rtr_set_resolution.argtypes = [vp, i32, i32]
    L.rtr_project.argtypes = [vp, vp, vp, vp]
    L.rtr_project_filtered.argtypes = [vp, vp, vp, vp]
    L.rtr_render.argtypes = [vp, vp, i32]
    L.rtr_clear.argtypes = [vp]
    L.rtr_min_depth_pass.argtypes = [vp, vp]
    L.rtr_accumulate_pass.argtypes = [vp, vp]
    L.rtr_resolve.argtypes = [vp]
    L.rtr_filter.argtypes = [vp]
    L.rtr_reorder_points.argtypes = [vp]
    L.rtr_resolve_range.argtypes = [vp, vp, u64, u64]
    L.rtr_device_buffer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz)]
    L.rtr_download_buffer.argtypes = [vp, i32, vp, sz]
    L.rtr_timing_enable.argtypes = [vp, i32]
    L.rtr_timing_reset.argtypes = [vp]
    L.rtr_set_option.argtypes = [vp, C.c_char_p, i32]
    L.rtr_stream_probe.argtypes = [vp, vp]
    L.rtr_timing_get.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(u64)]
    L.rtr_p2p_export.argtypes = [vp, vp]
    L.rtr_p2p_open.argtypes = [vp, i32, i32, vp]
    L.rtr_p2p_close.argtypes = [vp]
    L.rtr_p2p_min_depth.argtypes = [vp]
    L.rtr_p2p_sum_resolve.argtypes = [vp]
    L.rtr_p2p_status.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.rtr_p2p_render.argtypes = [vp, vp, i32]
    L.rtr_p2p_render_owned.argtypes = [vp, vp, i32, i32]
    L.rtr_frame_stats.argtypes = [vp, vp]
    L.rtr_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i32)]
    L.rtr_host_output_buffers.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp)]
    L.rtr_project_async.argtypes = [vp, vp, i32, i32]
    L.rtr_wait.argtypes = [vp, i32]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("rtr_last_error", "rtr_default_params"):
            fn.restype = i32
    _lib = L
    return L
