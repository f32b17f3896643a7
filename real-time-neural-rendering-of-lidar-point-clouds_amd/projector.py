"""Host-side mirror of the reference's ProjectCloud on top of the C ABI.

`Projector` is a 1:1 wrapper of include/rtr.h.  `ProjectCloud` keeps the reference's
public interface (reference: src/RTRenderer/include/project_cloud.h:11-19 and
src/project_cloud.cu:268-434): same method names, argument meaning (calibration,
world->camera extrinsics, optional caller-allocated colour / depth outputs) and return
codes (1 ok, -1 when both outputs are None).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .camera import compose_projection


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """A context-owned device buffer exposed through __cuda_array_interface__ so that
    torch.as_tensor(buf, device='cuda') aliases it (zero copy) for RCCL collectives."""

    def __init__(self, ptr, shape, typestr):
        self.ptr, self.shape, self.typestr = ptr, tuple(shape), typestr
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


class Projector:
    """Thin object wrapper over the rtr_* C ABI (one context = one GPU)."""

    def __init__(self, device=0):
        self._lib = L.lib()
        self._ctx = C.c_void_p()
        rc = self._lib.rtr_create(C.byref(self._ctx), int(device))
        if rc != 0:
            raise L.RtrError(rc, (self._lib.rtr_last_error(None) or b"").decode())
        self.device = int(device)
        self.W = self.H = 0

    # -- plumbing
    def _chk(self, rc):
        if rc != 0:
            raise L.RtrError(rc, (self._lib.rtr_last_error(self._ctx) or b"").decode())

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.rtr_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._chk(self._lib.rtr_synchronize(self._ctx))

    def set_stream(self, hip_stream_ptr):
        """Run on the given hipStream_t handle (0 / None = HIP's default stream)."""
        self._chk(self._lib.rtr_set_stream(self._ctx, C.c_void_p(hip_stream_ptr or 0)))

    def reset_stream(self):
        self._chk(self._lib.rtr_reset_stream(self._ctx))

    @property
    def params(self):
        p = L.RtrParams()
        self._chk(self._lib.rtr_get_params(self._ctx, C.byref(p)))
        return p

    def set_params(self, **kw):
        p = self.params
        for k, v in kw.items():
            setattr(p, k, v)
        self._chk(self._lib.rtr_set_params(self._ctx, C.byref(p)))

    def set_option(self, key, value):
        self._chk(self._lib.rtr_set_option(self._ctx, key.encode(), int(value)))

    def get_option(self, key):
        """Current value of an option; also "reordered" (the resident cloud was sorted by the library) and
        "order_ratio_ppm" (mean 256-point-chunk diagonal / cloud diagonal as uploaded, in 1e-6)."""
        v = C.c_int()
        self._chk(self._lib.rtr_get_option(self._ctx, key.encode(), C.byref(v)))
        return v.value

    def stream_probe(self, P):
        P = self._P(P)
        self._chk(self._lib.rtr_stream_probe(self._ctx, _vp(P)))

    # -- cloud
    def upload_points(self, xyz, rgb):
        """xyz: float32 [n,3|4] (the reference's float4 (x,y,z,1) or tight xyz);
        rgb: uint8 [n,3|4] (uchar4 (c0,c1,c2,255) or tight triples)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        if xyz.ndim != 2 or xyz.shape[1] not in (3, 4) or rgb.ndim != 2 or rgb.shape[1] not in (3, 4) \
                or rgb.shape[0] != xyz.shape[0]:
            raise ValueError("xyz must be [n,3|4] float32 and rgb [n,3|4] uint8 with equal n")
        self._chk(self._lib.rtr_upload_points(self._ctx, _vp(xyz), xyz.shape[1] * 4, _vp(rgb), rgb.shape[1],
                                              xyz.shape[0]))

    def generate_synthetic(self, scene, seed, first, count, total):
        sc = L.SCENES[scene] if isinstance(scene, str) else int(scene)
        self._chk(self._lib.rtr_generate_synthetic(self._ctx, sc, seed, first, count, total))

    def reorder_points(self):
        """One-off Morton sort of the resident cloud (never changes a frame)."""
        self._chk(self._lib.rtr_reorder_points(self._ctx))

    @property
    def num_points(self):
        n = C.c_uint64()
        self._chk(self._lib.rtr_num_points(self._ctx, C.byref(n)))
        return n.value

    def download_points(self, first=0, count=None):
        count = self.num_points - first if count is None else count
        xyzw = np.empty((count, 4), np.float32)
        rgba = np.empty((count, 4), np.uint8)
        self._chk(self._lib.rtr_download_points(self._ctx, _vp(xyzw), _vp(rgba), first, count))
        return xyzw, rgba

    # -- frames
    def set_resolution(self, W, H):
        self._chk(self._lib.rtr_set_resolution(self._ctx, int(W), int(H)))
        self.W, self.H = int(W), int(H)

    @staticmethod
    def _P(P):
        P = np.ascontiguousarray(P, dtype=np.float32).reshape(16)
        return P

    def project(self, P, want_img=True, want_depth=True, filtered=False):
        """-> (img uint8 [H,W,3] | None, depth float32 [H,W] | None)"""
        P = self._P(P)
        img = np.empty((self.H, self.W, 3), np.uint8) if want_img else None
        depth = np.empty((self.H, self.W), np.float32) if want_depth else None
        fn = self._lib.rtr_project_filtered if filtered else self._lib.rtr_project
        self._chk(fn(self._ctx, _vp(P), _vp(img), _vp(depth)))
        return img, depth

    def project_into(self, P, img, depth, filtered=False):
        """The reference's call shape: fills caller-allocated arrays (either may be None), synchronous."""
        P = self._P(P)
        fn = self._lib.rtr_project_filtered if filtered else self._lib.rtr_project
        self._chk(fn(self._ctx, _vp(P), _vp(img), _vp(depth)))

    # -- asynchronous host outputs (rtr.h section 4b)
    def host_output_buffers(self, slot):
        """-> (img uint8 [H,W,3], depth float32 [H,W]): numpy views of the library's pinned buffers of `slot`
        (valid until the next set_resolution; filled by project_async, complete after wait_outputs)."""
        pi, pd = C.c_void_p(), C.c_void_p()
        self._chk(self._lib.rtr_host_output_buffers(self._ctx, int(slot), C.byref(pi), C.byref(pd)))
        img = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint8)), shape=(self.H, self.W, 3))
        depth = np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_float)), shape=(self.H, self.W))
        return img, depth

    def project_async(self, P, slot, filtered=False):
        """Renders a frame and queues the copies of depth + image into the pinned buffers of `slot`; does not wait."""
        P = self._P(P)
        self._chk(self._lib.rtr_project_async(self._ctx, _vp(P), int(slot), 1 if filtered else 0))

    def wait_outputs(self, slot=-1):
        self._chk(self._lib.rtr_wait(self._ctx, int(slot)))

    def render(self, P, with_filter=False):
        P = self._P(P)
        self._chk(self._lib.rtr_render(self._ctx, _vp(P), 1 if with_filter else 0))

    # -- phases (multi-GPU: reduce the device buffers between them)
    def clear(self):
        self._chk(self._lib.rtr_clear(self._ctx))

    def min_depth_pass(self, P):
        P = self._P(P)
        self._chk(self._lib.rtr_min_depth_pass(self._ctx, _vp(P)))

    def accumulate_pass(self, P):
        P = self._P(P)
        self._chk(self._lib.rtr_accumulate_pass(self._ctx, _vp(P)))

    def resolve(self):
        self._chk(self._lib.rtr_resolve(self._ctx))

    def filter(self):
        self._chk(self._lib.rtr_filter(self._ctx))

    def resolve_range(self, first_pixel, count, acc_dev_ptr=None):
        self._chk(self._lib.rtr_resolve_range(self._ctx, C.c_void_p(acc_dev_ptr or 0), first_pixel, count))

    # -- buffers
    _BUF = {L.BUF_DEPTH: (np.uint32, "<u4", lambda w, h: (h, w)),
            L.BUF_ACCUM: (np.uint32, "<u4", lambda w, h: (h, w, 4)),
            L.BUF_IMAGE: (np.uint8, "|u1", lambda w, h: (h, w, 3)),
            L.BUF_TENSOR: (np.uint16, "<f2", lambda w, h: (1, 5, h, w)),
            L.BUF_MASK: (np.uint8, "|u1", lambda w, h: (h, w)),
            L.BUF_MINMAX: (np.uint32, "<u4", lambda w, h: (2,))}

    def device_buffer(self, which, typestr=None):
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        self._chk(self._lib.rtr_device_buffer(self._ctx, which, C.byref(ptr), C.byref(nbytes)))
        _, ts, shp = self._BUF[which]
        return DeviceBuffer(ptr.value, shp(self.W, self.H), typestr or ts)

    def download(self, which):
        dt, _, shp = self._BUF[which]
        out = np.empty(shp(self.W, self.H), dt)
        self._chk(self._lib.rtr_download_buffer(self._ctx, which, _vp(out), out.nbytes))
        return out

    # -- peer-to-peer exchange (rtr.h section 5b)
    def p2p_export(self):
        """-> bytes: this rank's handle block (exchange it with the other ranks, then p2p_open)."""
        buf = C.create_string_buffer(L.P2P_HANDLES_BYTES)
        self._chk(self._lib.rtr_p2p_export(self._ctx, buf))
        return buf.raw

    def p2p_open(self, rank, world, handles):
        """handles: the `world` handle blocks in rank order."""
        assert len(handles) == world and all(len(h) == L.P2P_HANDLES_BYTES for h in handles)
        self._chk(self._lib.rtr_p2p_open(self._ctx, rank, world, C.create_string_buffer(b"".join(handles))))

    def p2p_close(self):
        self._chk(self._lib.rtr_p2p_close(self._ctx))

    def p2p_min_depth(self):
        self._chk(self._lib.rtr_p2p_min_depth(self._ctx))

    def p2p_sum_resolve(self):
        self._chk(self._lib.rtr_p2p_sum_resolve(self._ctx))

    def p2p_render(self, P, with_filter=False):
        P = self._P(P)
        self._chk(self._lib.rtr_p2p_render(self._ctx, _vp(P), 1 if with_filter else 0))

    def p2p_render_owned(self, P, with_filter=False, frame_owner=0):
        """Owner-computes sharded frame: afterwards only rank `frame_owner` holds the global frame."""
        P = self._P(P)
        self._chk(self._lib.rtr_p2p_render_owned(self._ctx, _vp(P), 1 if with_filter else 0, int(frame_owner)))

    def p2p_timeouts(self):
        n = C.c_uint32()
        self._chk(self._lib.rtr_p2p_status(self._ctx, C.byref(n)))
        return n.value

    # -- measurement
    def frame_stats(self):
        """Statistics of the last binned frame (synchronises): work items of the tile kernel, how many of
        them are slices of split tiles, in-frustum entries, entries of the heaviest tile, slice size,
        tile-store error bits of that frame (0 = none), split tiles, 256-point chunks with an in-frustum point
        (each loads 1 KiB of colours)."""
        out = np.zeros(8, np.uint32)
        self._chk(self._lib.rtr_frame_stats(self._ctx, _vp(out)))
        return {"items": int(out[0]), "split_items": int(out[1]), "entries": int(out[2]), "heaviest_tile": int(out[3]),
                "slice": int(out[4]), "errors": int(out[5]), "split_tiles": int(out[6]), "colour_chunks": int(out[7])}

    def timing_enable(self, on=True):
        """True / 1: every phase, 2: only the streaming point kernels, 3: those on every 4th launch, False / 0: off."""
        self._chk(self._lib.rtr_timing_enable(self._ctx, int(on)))

    def timing_reset(self):
        self._chk(self._lib.rtr_timing_reset(self._ctx))

    def timing(self):
        """-> {kernel name: (total_ms, launches)} (synchronises the stream)"""
        out = {}
        for k, name in enumerate(L.KERNEL_NAMES):
            ms, n = C.c_double(), C.c_uint64()
            self._chk(self._lib.rtr_timing_get(self._ctx, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out


class ProjectCloud:
    """Drop-in mirror of the reference class (project_cloud.h:11-19).

    ctor: the reference flattens its grid into float4 / uchar4 arrays
    (project_cloud.cu:191-192, Octreegrid.h:162-180); here the caller passes those
    flattened arrays (or tight xyz / rgb) directly.  `modelFilename` names a TorchScript file
    under $HOME/.render_cache like in the reference (project_cloud.cu:225-246); the U-Net itself
    is not part of this library -- computeFull only hands it the device tensor (zero copy) and
    post-processes its output (project_cloud.cu:471-487).  `set_model` accepts any callable
    instead of a file."""

    def __init__(self, vertices, colors, modelFilename="", device=0, reorder=True):
        """reorder: True keeps the library's default upload policy (the cloud is Morton-sorted once when its
        256-point chunks are not spatially compact -- the grid's 0.25 m blocks are unordered inside); False never
        sorts.  Frames do not depend on the point order."""
        self.modelFilename = modelFilename
        self.model = None
        self._device = device
        self._bound_stream = None
        if modelFilename != "":
            import os
            import torch
            path = os.path.join(os.environ.get("HOME", ""), ".render_cache", modelFilename)
            if not os.path.exists(path):  # the reference prints this and exits (project_cloud.cu:232-236)
                raise FileNotFoundError("Model file does not exist: %s (export a TorchScript model for this "
                                        "camera resolution first)" % path)
            self.model = torch.jit.load(path, map_location="cuda:%d" % device)
        self._p = Projector(device)
        if not reorder:
            self._p.set_option("auto_reorder", 0)
        self._p.upload_points(vertices, colors)

    def set_model(self, model):
        """Use `model` (a callable taking the fp16 {1,5,H,W} cuda tensor) in computeFull."""
        self.model = model

    @classmethod
    def from_grid(cls, grid, modelFilename="", device=0):
        """The reference constructor's argument: a block grid (project_cloud.cu:189-206);
        `grid` is a formats.Grid (CloudReader::loadCloud's result, cloudreader.cpp:180-216)."""
        return cls(grid.vertex_positions(), grid.vertex_colors(), modelFilename, device)

    @property
    def projector(self):
        return self._p

    def _frame(self, calibration, extrinsics, color, depth, filtered):
        if color is None and depth is None:  # project_cloud.cu:270-273
            return -1
        W, H = calibration.getWidth(), calibration.getHeight()
        for name, arr, shape, dt in (("color", color, (H, W, 3), np.uint8), ("depth", depth, (H, W), np.float32)):
            if arr is not None and (arr.dtype != dt or arr.shape != shape or not arr.flags.c_contiguous):
                raise ValueError("%s must be a C-contiguous %s array of shape %s (main.cpp:93-94)" % (name, dt, shape))
        self._p.set_resolution(W, H)  # project_cloud.cu:275-298
        P = compose_projection(calibration.getIntrinsicsMatrix(), extrinsics)  # project_cloud.cu:318
        fn = self._p._lib.rtr_project_filtered if filtered else self._p._lib.rtr_project
        self._p._chk(fn(self._p._ctx, _vp(P), _vp(color), _vp(depth)))
        return 1

    def computeRGBD(self, calibration, extrinsics, color, depth):
        """project_cloud.cu:268-312.  extrinsics = world->camera 4x4 (main.cpp:96)."""
        return self._frame(calibration, extrinsics, color, depth, False)

    def computeFilteredRGBD(self, calibration, extrinsics, color, depth):
        """project_cloud.cu:394-434."""
        return self._frame(calibration, extrinsics, color, depth, True)

    def computeFull(self, calibration, extrinsics, color, depth):
        """project_cloud.cu:437-493: projection + prefilter, then the model on the resident fp16
        tensor; color <- uint8(round(output * 255)) like cv::Mat::convertTo(CV_8UC3, 255.0),
        depth <- the prefiltered depth buffer.  Either output may be None; returns 1."""
        import torch
        if self.model is None:  # the reference warns in the ctor and then crashes in forward()
            raise L.RtrError(L.RTR_ERR_INVALID, "No model: computeFull needs modelFilename or set_model()")
        W, H = calibration.getWidth(), calibration.getHeight()
        for name, arr, shape, dt in (("color", color, (H, W, 3), np.uint8), ("depth", depth, (H, W), np.float32)):
            if arr is not None and (arr.dtype != dt or arr.shape != shape or not arr.flags.c_contiguous):
                raise ValueError("%s must be a C-contiguous %s array of shape %s (main.cpp:93-94)" % (name, dt, shape))
        dev = torch.device("cuda", self._device)
        with torch.cuda.device(dev):
            # kernels and the model share torch's current stream, so the hand-off needs no sync
            stream = torch.cuda.current_stream(dev).cuda_stream
            if self._bound_stream != stream:
                self._p.set_stream(stream)
                self._bound_stream = stream
            self._p.set_resolution(W, H)
            P = compose_projection(calibration.getIntrinsicsMatrix(), extrinsics)
            self._p.render(P, True)
            inp = torch.as_tensor(self._p.device_buffer(L.BUF_TENSOR), device=dev)  # zero copy (from_blob, :471)
            with torch.no_grad():
                out = self.model(inp)
            out = out[0].permute(1, 2, 0).contiguous()  # :475
            if color is not None:
                img = (out.float() * 255.0).round().clamp(0, 255).to(torch.uint8)  # convertTo(CV_8UC3, 255.0), :480
                color[...] = img.cpu().numpy()
            if depth is not None:
                depth[...] = self._p.download(L.BUF_DEPTH).view(np.float32)  # :485
        return 1

    def tensor_device_buffer(self):
        """The planar fp16 {1,5,H,W} device tensor computeFull feeds to the U-Net
        (project_cloud.cu:471); valid after computeFilteredRGBD."""
        return self._p.device_buffer(L.BUF_TENSOR)
