"""ctypes/numpy wrapper around the CPU oracle (oracle/rtr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
PARITY UNPINNED (see rtr_oracle.c header): the reference has no fixtures.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librtr_oracle.so")
EMPTY_DEPTH = 0x7F7FFFFF
SCENES = {"uniform_box": 0, "room_shell": 1}


class Params(C.Structure):
    _fields_ = [("depth_window", C.c_float), ("filter_strength", C.c_float),
                ("gradient_threshold", C.c_float), ("levels", C.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "rtr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_selftest.restype = C.c_int
        L.orc_project_point.restype = C.c_int64
        L.orc_project_point.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_generate.restype = C.c_int
        L.orc_generate.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_filter.restype = C.c_int
        L.orc_project_mt.restype = C.c_int
        L.orc_envelope_points.restype = C.c_int
        L.orc_project_variant.restype = C.c_int
        _lib = L
        assert L.orc_selftest() == 1, "oracle build violates the arithmetic contract"
    return _lib


def default_params():
    p = Params()
    lib().orc_default_params(C.byref(p))
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _P(P):
    P = np.ascontiguousarray(P, dtype=np.float32).reshape(16)
    return P


def compose_projection(K, E):
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    E = np.ascontiguousarray(E, dtype=np.float64).reshape(16)
    P = np.empty(16, np.float32)
    lib().orc_compose_projection(_ptr(K), _ptr(E), _ptr(P))
    return P


def project_point(P, x, y, z, W, H):
    """-> (pixel id or -1, depth bits)"""
    P = _P(P)
    bits = C.c_uint32(0)
    pix = lib().orc_project_point(_ptr(P), x, y, z, W, H, C.addressof(bits))
    return int(pix), int(bits.value)


def generate(scene, seed, first, count, total):
    """-> (xyzw float32 [count,4], rgba uint8 [count,4])"""
    sc = SCENES[scene] if isinstance(scene, str) else int(scene)
    xyzw = np.empty((count, 4), np.float32)
    rgba = np.empty((count, 4), np.uint8)
    rc = lib().orc_generate(sc, seed, first, count, total, _ptr(xyzw), _ptr(rgba))
    if rc != 0:
        raise ValueError("orc_generate failed")
    return xyzw, rgba


def _cloud(xyz, rgb):
    xyz = np.asarray(xyz)
    assert xyz.dtype == np.float32 and xyz.ndim == 2 and xyz.shape[1] in (3, 4)
    xyz = np.ascontiguousarray(xyz)
    if rgb is not None:
        rgb = np.asarray(rgb)
        assert rgb.dtype == np.uint8 and rgb.ndim == 2 and rgb.shape[1] in (3, 4) and rgb.shape[0] == xyz.shape[0]
        rgb = np.ascontiguousarray(rgb)
    return xyz, rgb


def clear(W, H):
    depth = np.full(W * H, EMPTY_DEPTH, np.uint32)
    acc = np.zeros(W * H * 4, np.uint32)
    return depth, acc


def min_depth_pass(xyz, P, W, H, depth):
    xyz, _ = _cloud(xyz, None)
    P = _P(P)
    lib().orc_min_depth_pass(_ptr(xyz), C.c_size_t(xyz.strides[0]), C.c_size_t(xyz.shape[0]), _ptr(P),
                             W, H, _ptr(depth))
    return depth


def accumulate_pass(xyz, rgb, P, W, H, depth, acc, window=None):
    xyz, rgb = _cloud(xyz, rgb)
    P = _P(P)
    w = default_params().depth_window if window is None else window
    lib().orc_accumulate_pass(_ptr(xyz), C.c_size_t(xyz.strides[0]), _ptr(rgb), C.c_size_t(rgb.strides[0]),
                              C.c_size_t(xyz.shape[0]), _ptr(P), W, H, _ptr(depth), _ptr(acc), C.c_float(w))
    return acc


def resolve(acc, W, H):
    img = np.empty(W * H * 3, np.uint8)
    lib().orc_resolve(_ptr(acc), C.c_size_t(W * H), _ptr(img))
    return img.reshape(H, W, 3)


def project(xyz, rgb, P, W, H, params=None):
    """Naive single-thread host loop (project_cloud.cu:314-329).
    -> dict(depth_bits uint32 [H,W], acc uint32 [H,W,4], img uint8 [H,W,3])"""
    xyz, rgb = _cloud(xyz, rgb)
    P = _P(P)
    prm = params or default_params()
    depth = np.empty(W * H, np.uint32)
    acc = np.empty(W * H * 4, np.uint32)
    img = np.empty(W * H * 3, np.uint8)
    lib().orc_project(_ptr(xyz), C.c_size_t(xyz.strides[0]), _ptr(rgb), C.c_size_t(rgb.strides[0]),
                      C.c_size_t(xyz.shape[0]), _ptr(P), W, H, C.byref(prm), _ptr(depth), _ptr(acc), _ptr(img))
    return {"depth_bits": depth.reshape(H, W), "acc": acc.reshape(H, W, 4), "img": img.reshape(H, W, 3)}


class MTProjector:
    """Multi-thread CPU projector (the timed CPU baseline); scratch is reused."""

    def __init__(self, W, H, nthreads):
        self.W, self.H, self.nthreads = W, H, nthreads
        self.scratch = np.empty(nthreads * W * H * 4, np.uint32)
        self.depth = np.empty(W * H, np.uint32)
        self.acc = np.empty(W * H * 4, np.uint32)
        self.img = np.empty(W * H * 3, np.uint8)

    def project(self, xyz, rgb, P, params=None):
        xyz, rgb = _cloud(xyz, rgb)
        P = _P(P)
        prm = params or default_params()
        rc = lib().orc_project_mt(_ptr(xyz), C.c_size_t(xyz.strides[0]), _ptr(rgb), C.c_size_t(rgb.strides[0]),
                                  C.c_size_t(xyz.shape[0]), _ptr(P), self.W, self.H, C.byref(prm),
                                  self.nthreads, _ptr(self.scratch), _ptr(self.depth), _ptr(self.acc),
                                  _ptr(self.img))
        assert rc == 0
        return {"depth_bits": self.depth.reshape(self.H, self.W), "acc": self.acc.reshape(self.H, self.W, 4),
                "img": self.img.reshape(self.H, self.W, 3)}


MM_VARIANTS = {0: "contract (right products fused, left to right)", 1: "first add fused with the left product",
               2: "nothing contracted (-fmad=false)", 3: "re-associated: all fused, right to left",
               4: "re-associated: m3 folded into the first fma"}
DV_VARIANTS = {0: "contract: x * RN(1/z)", 1: "IEEE x / z", 2: "x * (RN(1/z) - 2 ulp)", 3: "x * (RN(1/z) - 1 ulp)",
               4: "x * (RN(1/z) + 1 ulp)", 5: "x * (RN(1/z) + 2 ulp)"}


def envelope_points(xyz, P, W, H, mm, dv, nthreads=8):
    """Per-point comparison of arithmetic variant (mm, dv) with the contract (rtr_oracle.c, "ENVELOPE").
    -> dict(accepted, either, flips, max_depth_ulp)"""
    xyz, _ = _cloud(xyz, None)
    P = _P(P)
    out = np.zeros(4, np.uint64)
    rc = lib().orc_envelope_points(_ptr(xyz), C.c_size_t(xyz.strides[0]), C.c_size_t(xyz.shape[0]), _ptr(P), W, H,
                                   int(mm), int(dv), int(nthreads), _ptr(out))
    assert rc == 0
    return {"accepted": int(out[0]), "either": int(out[1]), "flips": int(out[2]), "max_depth_ulp": int(out[3])}


def project_variant(xyz, rgb, P, W, H, mm, dv, params=None):
    """A whole frame under arithmetic variant (mm, dv): same outputs as project()."""
    xyz, rgb = _cloud(xyz, rgb)
    P = _P(P)
    prm = params or default_params()
    depth = np.empty(W * H, np.uint32)
    acc = np.empty(W * H * 4, np.uint32)
    img = np.empty(W * H * 3, np.uint8)
    rc = lib().orc_project_variant(_ptr(xyz), C.c_size_t(xyz.strides[0]), _ptr(rgb), C.c_size_t(rgb.strides[0]),
                                   C.c_size_t(xyz.shape[0]), _ptr(P), W, H, C.byref(prm), int(mm), int(dv),
                                   _ptr(depth), _ptr(acc), _ptr(img))
    assert rc == 0
    return {"depth_bits": depth.reshape(H, W), "acc": acc.reshape(H, W, 4), "img": img.reshape(H, W, 3)}


def filter(depth_bits, img, params=None, want_tensor=True):
    """applyDepthFilter (project_cloud.cu:331-392) on copies of the inputs.
    -> dict(depth float32 [H,W], img uint8 [H,W,3], mask uint8 [H,W],
            tensor uint16 (fp16 bits) [5,H,W], minmax uint32[2])"""
    H, W = depth_bits.shape
    prm = params or default_params()
    d = np.ascontiguousarray(depth_bits, dtype=np.uint32).copy()
    im = np.ascontiguousarray(img, dtype=np.uint8).copy()
    mask = np.empty(W * H, np.uint8)
    tensor = np.empty(5 * W * H, np.uint16) if want_tensor else None
    mm = np.empty(2, np.uint32)
    rc = lib().orc_filter(_ptr(d), _ptr(im), W, H, C.byref(prm), _ptr(mask), _ptr(tensor), _ptr(mm))
    if rc != 0:
        raise ValueError("orc_filter: unsupported dimensions (need W %% 2^levels == 0): %dx%d" % (W, H))
    return {"depth": d.view(np.float32).reshape(H, W), "img": im.reshape(H, W, 3), "mask": mask.reshape(H, W),
            "tensor": None if tensor is None else tensor.reshape(5, H, W), "minmax": mm}


def f32_to_f16(x):
    return int(lib().orc_f32_to_f16(C.c_float(x)))


def f16_to_f32(h):
    return float(lib().orc_f16_to_f32(C.c_uint16(h)))
