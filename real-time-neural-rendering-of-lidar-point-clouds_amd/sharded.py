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
"""
import torch
import torch.distributed as dist

from . import _lib as L


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n points owned by `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world


class HipLocal:
    """Adapter: a `Projector` plus zero-copy torch views of its depth / accumulator
    buffers.  Depth bit patterns of positive floats (and the 0x7F7FFFFF sentinel) are
    < 2^31, so int32 MIN orders them exactly like the reference's u32 atomicMin; int32
    SUM wraps exactly like u32 addition."""

    def __init__(self, projector):
        self.p = projector
        self._views = None

    def bind_stream(self):
        self.p.set_stream(torch.cuda.current_stream(self.p.device).cuda_stream)

    def _mk(self):
        if self._views is None or self._views[0] != (self.p.W, self.p.H):
            dev = torch.device("cuda", self.p.device)
            d = torch.as_tensor(self.p.device_buffer(L.BUF_DEPTH, "<i4"), device=dev).view(-1)
            a = torch.as_tensor(self.p.device_buffer(L.BUF_ACCUM, "<i4"), device=dev).view(-1)
            self._views = ((self.p.W, self.p.H), d, a)
        return self._views

    def depth_tensor(self):
        return self._mk()[1]

    def accum_tensor(self):
        return self._mk()[2]

    def clear(self):
        self.p.clear()

    def min_depth_pass(self, P):
        self.p.min_depth_pass(P)

    def accumulate_pass(self, P):
        self.p.accumulate_pass(P)

    def resolve(self):
        self.p.resolve()

    def filter(self):
        self.p.filter()

    def render(self, P, with_filter):
        self.p.render(P, with_filter)


class ShardedProjector:
    """Runs the frame sequence over `group`; every rank ends with the full frame.

    `local` is any object with clear / min_depth_pass / accumulate_pass / resolve /
    filter and depth_tensor() / accum_tensor() (int32 torch tensors aliasing its frame
    buffers): `HipLocal` in production, an oracle-backed stand-in in the CPU tests."""

    def __init__(self, local, group=None):
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def render(self, P, with_filter=False):
        lo = self.local
        if self.world == 1 and hasattr(lo, "render"):
            lo.render(P, with_filter)  # no exchange step: the fused whole-frame call
            return
        lo.clear()
        lo.min_depth_pass(P)
        if self.world > 1:
            dist.all_reduce(lo.depth_tensor(), op=dist.ReduceOp.MIN, group=self.group)
        lo.accumulate_pass(P)
        if self.world > 1:
            dist.all_reduce(lo.accum_tensor(), op=dist.ReduceOp.SUM, group=self.group)
        lo.resolve()
        if with_filter:
            lo.filter()
