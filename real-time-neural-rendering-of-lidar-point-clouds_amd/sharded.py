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
Exchange forms:
  exchange="collective"  torch.distributed collectives on the library's own buffers (below);
  exchange="p2p"         the library's hand-written exchange (rtr.h 5b): every rank maps the other
                         ranks' frame buffers through hipIpc and pulls its pixel slice over xGMI
                         (MIN of depth, SUM + resolve of colour), four flag barriers per frame, no
                         host round trip.  The first frame is rendered both ways and compared on
                         every rank; any difference, error or barrier timeout drops back to the
                         collectives for good.  In steady state every `check_every`-th frame ends with
                         a look at the barrier-timeout word (a mapped host word, set by a barrier that
                         gave up on a stalled rank -- the frames of the ranks that waited are then
                         undefined) and an agreement over all ranks: if any rank saw a timeout, all
                         drop to the collectives and that frame is rendered again with them
                         (`p2p_suspect_frames` names the frames since the previous clean check).
  exchange="owned"       the owner-computes form (rtr_p2p_render_owned): no MIN / SUM exchange at all -- after ONE
                         barrier every screen tile is produced by one of the ranks that have points in it, which
                         reads the other occupying ranks' entries out of their tile stores over xGMI and runs the
                         fused per-tile z-buffer once; a second barrier, then the FRAME'S OWNER (frame k -> rank
                         k mod world unless `fixed_owner` is set) collects the tiles and runs the prefilter.  Only
                         that rank ends with the whole frame (`last_owner`).  Verified like "p2p": the first frames
                         are rendered with the collectives too and compared on every rank in turn.
Colour forms of the collective exchange:
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
        self._p2p_res = None  # resolution the peers' buffers are mapped for (shared by every renderer on this context)

    @property
    def p2p_res(self):
        """None unless the library still holds the peers' mappings: a new cloud or resolution on this context closes the
        exchange on the C side (rtr.h, "p2p_open"), and the next sharded frame must run p2p_setup again -- on every
        rank, which is the case when every rank replaced its shard."""
        if self._p2p_res is not None and not self.p.get_option("p2p_open"):
            self._p2p_res = None
        return self._p2p_res

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

    # peer-to-peer exchange (rtr.h 5b)
    def p2p_setup(self, rank, world, group):
        """Exchange the hipIpc handle blocks over `group` and map the peers' buffers."""
        try:
            mine = self.p.p2p_export()
        except Exception as e:  # every rank still takes part in the gather below
            mine = "rank %d: %s" % (rank, e)
        blocks = [None] * world
        dist.all_gather_object(blocks, mine, group=group)
        bad = [b for b in blocks if not isinstance(b, bytes)]
        if bad:
            raise RuntimeError("rtr_p2p_export failed: %s" % "; ".join(map(str, bad)))
        self.p.p2p_open(rank, world, blocks)
        self._p2p_res = (self.p.W, self.p.H)

    def p2p_close(self):
        self._p2p_res = None
        self.p.p2p_close()

    def p2p_min_depth(self):
        self.p.p2p_min_depth()

    def p2p_sum_resolve(self):
        self.p.p2p_sum_resolve()

    def p2p_render(self, P, with_filter):
        self.p.p2p_render(P, with_filter)

    def p2p_render_owned(self, P, with_filter, frame_owner):
        self.p.p2p_render_owned(P, with_filter, frame_owner)

    def p2p_timeouts(self):
        return self.p.p2p_timeouts()


class ShardedProjector:
    """Runs the frame sequence over `group`; every rank ends with the full frame (exchange "owned": only the
    frame's owner, `last_owner`).

    `local` is any object with clear / min_depth_pass / accumulate_pass / resolve / filter
    and depth_tensor() / accum_tensor() (int32 torch tensors aliasing its frame buffers):
    `HipLocal` in production, an oracle-backed stand-in in the CPU tests.  The
    "reduce_scatter" colour form additionally needs resolve_range() and image_tensor()."""

    def __init__(self, local, group=None, colour="allreduce", force_exchange=False, exchange="collective",
                 check_every=16):
        assert colour in ("allreduce", "reduce_scatter") and exchange in ("collective", "p2p", "owned")
        self.fixed_owner = None     # exchange "owned": None = frame k belongs to rank k mod world
        self.last_owner = None      # ... the rank that holds the frame rendered last
        assert check_every >= 1
        self.check_every = check_every  # p2p: frames between two looks at the barrier-timeout word (1 = every frame)
        self.frame_no = 0               # p2p frames rendered since the exchange was verified
        self.p2p_suspect_frames = None  # (first, last) frame_no that may be undefined after a barrier timeout
        self.exchange = exchange
        self.p2p_note = None        # why the p2p exchange was dropped, if it was
        self._p2p_verified = False
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

    # -- peer-to-peer form ---------------------------------------------------------------
    def _all_agree(self, ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.local.depth_tensor().device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item())

    def _drop_p2p(self, why):
        self.exchange, self.p2p_note = "collective", why
        try:
            self.local.p2p_close()
        except Exception:
            pass

    def _p2p_frame(self, P):
        self.local.p2p_render(P, False)  # the same library call the later frames use

    def _render_p2p(self, P, with_filter):
        """-> False if the p2p form is (now) unavailable and the collectives must render the frame."""
        lo = self.local
        res = (lo.p.W, lo.p.H)
        if getattr(lo, "p2p_res", None) != res:  # (re)map the peers' buffers: collective, every rank gets here
            ok = True
            try:
                lo.p2p_setup(self.rank, self.world, self.group)
            except Exception as e:  # e.g. hipIpc refused
                ok, why = False, "setup failed: %s" % e
            if not self._all_agree(ok):
                self._drop_p2p(why if not ok else "setup failed on another rank")
                return False
            self._p2p_verified = False
        owned = self.exchange == "owned"
        if not self._p2p_verified:  # first frame: render with the collectives too and compare
            self._render_collective(P, False)
            ref_d, ref_i = lo.depth_tensor().clone(), lo.image_tensor().clone()
            same = True
            if owned:  # every rank owns the frame once and compares what it collected
                for r in range(self.world):
                    lo.p2p_render_owned(P, False, r)
                    lo.p.synchronize()
                    if r == self.rank:
                        same = bool(torch.equal(lo.depth_tensor(), ref_d) and torch.equal(lo.image_tensor(), ref_i))
                self.last_owner = self.world - 1
            else:
                self._p2p_frame(P)
                same = bool(torch.equal(lo.depth_tensor(), ref_d) and torch.equal(lo.image_tensor(), ref_i))
            same = same and lo.p2p_timeouts() == 0
            if not self._all_agree(same):
                self._drop_p2p("first frame differed from the collectives' (or a barrier timed out)")
                return False
            self._p2p_verified = True
            self.frame_no = 0
            if owned:  # (the caller's frame: with the owner it asked for)
                self.last_owner = self.fixed_owner if self.fixed_owner is not None else 0
                lo.p2p_render_owned(P, with_filter, self.last_owner)
                return True
        else:
            if owned:
                self.last_owner = self.fixed_owner if self.fixed_owner is not None else (self.frame_no + 1) % self.world
                lo.p2p_render_owned(P, with_filter, self.last_owner)
            else:
                lo.p2p_render(P, with_filter)  # the whole sequence in one library call
            self.frame_no += 1
            if self.frame_no % self.check_every == 0:
                # did a barrier give up on a stalled rank since the last check?  The word is host memory; what
                # costs is the agreement (one small all-reduce + host sync), hence only every check_every frames.
                lo.p.synchronize()  # this rank's frames up to here have run (or timed out)
                if not self._all_agree(lo.p2p_timeouts() == 0):
                    self.p2p_suspect_frames = (self.frame_no - self.check_every + 1, self.frame_no)
                    self._drop_p2p("a flag barrier timed out (a rank stalled): frames %d..%d since the last clean check "
                                   "may be undefined on the ranks that waited" % self.p2p_suspect_frames)
                    return False  # the caller renders this frame again with the collectives
            return True
        if with_filter:
            lo.filter()
        return True

    def render(self, P, with_filter=False):
        lo = self.local
        exchange = self.world > 1 or (self.force_exchange and dist.is_initialized())
        if not exchange and hasattr(lo, "render"):
            lo.render(P, with_filter)  # no exchange step: the fused whole-frame call
            return
        if exchange and self.exchange in ("p2p", "owned") and self._render_p2p(P, with_filter):
            return
        self.last_owner = None  # (every rank holds the frame)
        self._render_collective(P, with_filter, exchange)

    def _render_collective(self, P, with_filter, exchange=True):
        lo = self.local
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
