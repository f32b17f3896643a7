"""Multi-GPU sharding of the projector: one process per GPU, points split in contiguous
slices, the two exchange steps of SURVEY.md 8e done with torch.distributed collectives
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Nothing like this exists in the reference (single GPU).  It follows from its atomics:
depth is an atomicMin (render.cu:81) and colour is integer atomicAdd (render.cu:125-128),
both associative, commutative and exact, so any partition of the point array gives
bit-identical frame buffers after an element-wise MIN (depth) / SUM (accumulators).

Per frame:  clear -> local min-depth pass -> all-reduce MIN(depth)
            -> local accumulate pass against the GLOBAL minimum (the 2 cm window of
               render.cu:106 is relative to the global front surface)
            -> SUM of the accumulators -> resolve (-> prefilter, replicated).
Colour exchange forms:
  "allreduce"      all-reduce SUM of the 16 B/px accumulators, every rank resolves all pixels;
  "reduce_scatter" reduce-scatter SUM (each rank receives 1/N of the pixels), slice-local
                   resolve, all-gather of the 3 B/px image: about 0.55 x the bytes on the wire.
"""
import torch
import torch.distributed as dist

from . import _lib as L


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n points owned by `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world


class HipLocal:
    """Adapter: a `Projector` plus zero-copy torch views of its frame buffers.  Depth bit
    patterns of positive floats (and the 0x7F7FFFFF sentinel) are < 2^31, so int32 MIN orders
    them exactly like the reference's u32 atomicMin; int32 SUM wraps exactly like u32 addition."""

    def __init__(self, projector):
        self.p = projector
        self._views = None

    def bind_stream(self):
        """Run the kernels on torch's current stream so RCCL collectives order with them."""
        self.p.set_stream(torch.cuda.current_stream(self.p.device).cuda_stream)

    def _mk(self):
        if self._views is None or self._views[0] != (self.p.W, self.p.H):
            dev = torch.device("cuda", self.p.device)
            d = torch.as_tensor(self.p.device_buffer(L.BUF_DEPTH, "<i4"), device=dev).view(-1)
            a = torch.as_tensor(self.p.device_buffer(L.BUF_ACCUM, "<i4"), device=dev).view(-1)
            i = torch.as_tensor(self.p.device_buffer(L.BUF_IMAGE), device=dev).view(-1)
            self._views = ((self.p.W, self.p.H), d, a, i)
        return self._views

    def depth_tensor(self):
        return self._mk()[1]

    def accum_tensor(self):
        return self._mk()[2]

    def image_tensor(self):
        return self._mk()[3]

    def clear(self):
        self.p.clear()

    def min_depth_pass(self, P):
        self.p.min_depth_pass(P)

    def accumulate_pass(self, P):
        self.p.accumulate_pass(P)

    def resolve(self):
        self.p.resolve()

    def resolve_range(self, first_pixel, count, acc_slice=None):
        self.p.resolve_range(first_pixel, count, None if acc_slice is None else acc_slice.data_ptr())

    def filter(self):
        self.p.filter()

    def render(self, P, with_filter):
        self.p.render(P, with_filter)


class ShardedProjector:
    """Runs the frame sequence over `group`; every rank ends with the full frame.

    `local` is any object with clear / min_depth_pass / accumulate_pass / resolve / filter
    and depth_tensor() / accum_tensor() (int32 torch tensors aliasing its frame buffers):
    `HipLocal` in production, an oracle-backed stand-in in the CPU tests.  The
    "reduce_scatter" colour form additionally needs resolve_range() and image_tensor()."""

    def __init__(self, local, group=None, colour="allreduce", force_exchange=False):
        assert colour in ("allreduce", "reduce_scatter")
        self.force_exchange = force_exchange  # run the collectives even in a 1-rank group (tests)
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.colour = colour
        self._slice = None

    def _colour_reduce_scatter(self):
        lo = self.local
        acc, img = lo.accum_tensor(), lo.image_tensor()
        npix = acc.numel() // 4
        if npix % (4 * self.world) != 0:  # slices must start on a pixel quad
            return False
        per = npix // self.world
        if self._slice is None or self._slice.numel() != per * 4:
            self._slice = torch.empty(per * 4, dtype=acc.dtype, device=acc.device)
        dist.reduce_scatter_tensor(self._slice, acc, op=dist.ReduceOp.SUM, group=self.group)
        lo.resolve_range(self.rank * per, per, self._slice)
        mine = img[self.rank * per * 3:(self.rank + 1) * per * 3].clone()
        dist.all_gather_into_tensor(img, mine, group=self.group)
        return True

    def render(self, P, with_filter=False):
        lo = self.local
        exchange = self.world > 1 or (self.force_exchange and dist.is_initialized())
        if not exchange and hasattr(lo, "render"):
            lo.render(P, with_filter)  # no exchange step: the fused whole-frame call
            return
        lo.clear()
        lo.min_depth_pass(P)
        if exchange:
            dist.all_reduce(lo.depth_tensor(), op=dist.ReduceOp.MIN, group=self.group)
        lo.accumulate_pass(P)
        done = False
        if exchange and self.colour == "reduce_scatter":
            done = self._colour_reduce_scatter()
        if not done:
            if exchange:
                dist.all_reduce(lo.accum_tensor(), op=dist.ReduceOp.SUM, group=self.group)
            lo.resolve()
        if with_filter:
            lo.filter()
