#!/usr/bin/env python3
"""bench.py -- Mpoints/s projected + frames/s of the point-cloud -> framebuffer hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... --
    or plainly: without WORLD_SIZE in the environment bench.py starts the N rank processes itself, as children of a
    process that never touches the GPU, and passes rank 0's JSON line and the launcher's exit code through)

A "step" is one frame of BASELINE.json config C3: a 100 M-point synthetic cloud projected
to 1920x1080 (clear, min-depth pass, accumulate pass, resolve) plus the depth-heuristic
prefilter, one distinct camera pose of the orbit trajectory per frame.  The cloud is
synthesised on-device before the timed region (inputs resident in HBM).

N = 1: `value` is the LiDAR-like scene (room_shell); the same JSON line carries a `uniform_box`
object with the incoherent stress scene, both as handed over (hash order, option auto_reorder = 0)
and under the library's default upload policy (the cloud is Morton-sorted once on upload), each
with its own roofline figure (SURVEY.md 8d asks for both scenes).

N > 1: BASELINE config C4 -- the SAME 100 M-point cloud sharded in contiguous point slices over
the ranks ("scaling": "strong"), depth MIN / accumulator SUM exchanged per frame; the weak-scaling
figure (100 M points per GPU) is measured afterwards and reported under `weak_scaling`.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SEEDS = {"C2": 0xC0FFEE02, "C3": 0xC0FFEE03}
PREFILTER_BYTES_PER_PIXEL = 50.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="room_shell", choices=["uniform_box", "room_shell"])
    ap.add_argument("--no-filter", action="store_true", help="projection only (config C2 style)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="N>1: strong = --points in total, sharded (default: BASELINE C4); weak = --points per GPU")
    ap.add_argument("--no-weak", action="store_true", help="N>1: skip the extra weak-scaling measurement")
    ap.add_argument("--colour", default="reduce_scatter", choices=["allreduce", "reduce_scatter"],
                    help="N>1: how the colour accumulators are merged (see sharded.py)")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2, 3],
                    help="N>1: frames in flight per rank (2 = frame k's RCCL exchange overlaps frame k+1's kernels)")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2, 3],
                    help="N=1: independent frames alternate between this many contexts / HIP streams")
    ap.add_argument("--exchange", default="auto", choices=["auto", "collective", "p2p", "owned"],
                    help="N > 1: 'collective' = torch.distributed (RCCL) on the library's buffers; 'p2p' = the library's "
                         "hand-written MIN / SUM exchange over hipIpc-mapped peer buffers; 'owned' = the owner-computes "
                         "form (every tile produced once, by a rank that has points in it, over the entries of all "
                         "occupying ranks; the frame's owner rotates); 'auto' (default) times the collectives, then the "
                         "two hand-written forms (each verified against the collectives before and after), and reports "
                         "the fastest clean one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + several ranks on ONE GPU is a rehearsal of the N>1 logic")
    ap.add_argument("--force-exchange", action="store_true",
                    help="rehearsal on ONE GPU: run the N>1 code path (streams, RCCL collectives in a 1-rank "
                         "group, 2 frames in flight); the numbers are not a benchmark result")
    ap.add_argument("--time-every", type=int, default=-1, choices=[-1, 0, 1, 2, 4],
                    help="bracket the dominant kernel on every frame (1), every 2nd (2), every 4th frame (4) or never "
                         "(0: no roofline object; to measure what the bracketing itself costs).  Default: every 4th "
                         "frame, every 2nd for runs of <= 12 steps (a 20-step run times 5 launches).  A "
                         "bracketed dispatch costs ~10 us of stream time (completion signal + time stamps): 0.228 ms "
                         "per frame with every frame bracketed, 0.217 with every 4th")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="bracket every phase with HIP events (default: only the dominant streaming kernel)")
    ap.add_argument("--overlap", type=int, default=0, choices=[0, 1],
                    help="N=1: library option \"overlap\" (T1 of frame k+1 on a second stream beside the tail of frame k)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="measurement aid: rtr_set_option(KEY, VALUE) on every context before the cloud is generated "
                         "(e.g. pack=0, point_grid=1024); listed in config.options")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the separately reported legs (uniform_box, chunk culling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="0 = pick so the leg takes about 10-20 s (>= 3)")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size oracle parity gate on pose 0")
    return ap.parse_args()


def host_threads():
    """Threads for the CPU leg: the process's CPU share, capped at 16 (each thread owns a
    private frame buffer; more threads only grow the merge)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline(orc, pkg, args, xyzw, rgba):
    """The oracle's multi-thread projector (kind 'port') on the SAME cloud the GPU renders (the copy the
    parity gate downloaded), same resolution and trajectory, a bounded number of frames."""
    cores = host_threads()
    n, W, H = len(xyzw), args.width, args.height
    mt = orc.MTProjector(W, H, cores)
    mt.project(xyzw, rgba, pkg.orbit_projection(0, W, H))  # warm-up (page faults)
    t0 = time.perf_counter()
    mt.project(xyzw, rgba, pkg.orbit_projection(1, W, H))
    one = time.perf_counter() - t0
    frames = args.cpu_frames or max(3, min(50, int(12.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for k in range(frames):
        mt.project(xyzw, rgba, pkg.orbit_projection(2 + k, W, H))
    dt = time.perf_counter() - t0
    return {"value": n * frames / dt / 1e6, "unit": "Mpoints/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the benchmark's own %d-point %s cloud -> %dx%d (projection only, oracle "
                      "multi-thread projector, %d threads)" % (frames, n, args.scene, W, H, cores),
            "frames_per_s": frames / dt}


def frame_bytes(n_local, W, H, with_filter):
    """Bytes one frame NEEDS in this design (the cloud streamed once + the per-pixel work) and the
    figure SURVEY.md 8d prices the reference's structure at (two passes over xyz)."""
    px = float(W) * H
    required = 12.0 * n_local + 39.0 * px + (PREFILTER_BYTES_PER_PIXEL * px if with_filter else 0.0)
    two_pass = 24.0 * n_local + 39.0 * px
    return required, two_pass


def moved_bytes_model(stream_bpp, n_local, stats):
    """Bytes the point kernel moves per launch, from what it does: the coordinate stream (fp32 SoA 12 B/pt; of the
    lossless packed form what the kernel reads of it, see below), 1 KiB of colours for every 256-point chunk that holds an
    in-frustum point, 8 bytes written per in-frustum entry.  `stats` = frame statistics averaged over frames of the
    timed poses (rtr_frame_stats: entries, colour chunks); None for the atomic form, which streams xyz only."""
    if stats and stream_bpp < 11.9:
        # packed form, two streams per axis (round 4): every chunk's header (32 B per 256 points) and A streams -- the
        # first value of every lane, a quarter of the planes -- are read; the B streams only by the chunks that go on to
        # the long path, counted here by the chunks WITH an in-frustum point (the candidates the per-point tests then
        # reject are not counted: a lower bound, so the roofline fraction it gives errs low)
        planes = stream_bpp - 0.125
        b = n_local * (0.125 + planes / 4.0) + stats["colour_chunks"] * 256.0 * planes * 0.75
    else:
        b = stream_bpp * n_local
    if stats:
        b += 1024.0 * stats["colour_chunks"] + 8.0 * stats["entries"]
    return b


def roofline_of(kern_ms, launches, n_local, traffic, every, stream_bpp=12.0, stats=None, limiter=None, head=None):
    """`achieved` = the bytes the dominant kernel MOVES per launch / its average launch time; `frac` = that over
    the 8 TB/s HBM peak -- a roofline fraction, never above 1.  The bytes are the PMC traffic of the committed
    profile when it was taken on exactly this workload (`bytes_source` "pmc"), else the model above ("model":
    resident stream + colours + entries).  The contract's algorithmic figure (12 B/pt fp32 xyz) over
    the same time is kept as `vs_fp32_stream`: with the lossless packed coordinates (6-9 B/pt) the kernel reads
    fewer bytes than that, so this ratio may exceed 1 -- it is a speed-up over an ideal fp32 stream, not a
    statement about HBM efficiency."""
    t = kern_ms * 1e-3
    model = moved_bytes_model(stream_bpp, n_local, stats)
    moved, source = (float(traffic), "pmc") if traffic else (model, "model")
    achieved = moved / t / 1e9
    return {"bound": "hbm", "kernel": "min_depth (k_project_bin: stream + append)", "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_profile_head": head if traffic else None,
            "bytes_per_launch": moved, "bytes_source": source, "bytes_model": model,
            "avg_launch_ms": kern_ms, "launches_timed": int(launches),
            "resident_stream_bytes_per_point": stream_bpp,
            "frame_stats": stats,
            "algorithmic_bytes_per_launch": 12.0 * n_local,
            "vs_fp32_stream": 12.0 * n_local / t / 1e9 / HBM_PEAK_GBS,
            "limiter": limiter or ("not profiled on this workload; on the headline workload the SQ wait counters "
                                   "(profiles/) say what the kernel waits for"),
            "how": "HIP events on the kernel's stream inside the timed region, " +
                   {1: "every frame", 2: "every 2nd frame", 4: "every 4th frame"}.get(every, "?") +
                   "; in the tile-binned form the two events are the start / stop stamps of the kernel's own "
                   "dispatch (hipExtLaunchKernelGGL): no extra packets on the stream, but a bracketed dispatch still "
                   "costs ~10 us of stream time and reads ~5 % longer than in the rocprof trace"}


def measured_traffic(scene, n_local, W, H, with_filter, pack=1):
    """(HBM bytes per launch of the dominant kernel, what limits it, the git revision the PMC record was taken at) from
    the committed PMC profile -- only when that profile was taken on exactly this workload (else null: a constant is not
    a measurement).  The revision travels into the JSON line (`roofline.traffic_profile_head`) so that a record older
    than the kernel it prices is visible."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        doc = json.load(open(tpath))
        for rec in doc.get("records", []):
            if (rec.get("scene"), rec.get("points"), rec.get("width"), rec.get("height"), rec.get("prefilter"),
                    rec.get("pack", 1)) == (scene, n_local, W, H, with_filter, pack) and rec.get("kernel") == "min_depth":
                return rec.get("bytes_per_launch"), rec.get("limiter"), rec.get("head", doc.get("head"))
    except Exception:
        pass
    return None, None, None


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes as CHILDREN (torch.distributed.run as a
    subprocess; this process has not imported torch, let alone touched the GPU, and never replaces itself), wait for
    them, pass their output through.  Returns the launcher's exit code: non-zero when any rank failed."""
    import socket
    import subprocess
    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1)  # rehearsal: more ranks than GPUs share devices (gloo only)
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_exchange  # take the exchange code path
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    scaling = args.scaling or "strong"  # BASELINE C4: the same cloud split over the ranks

    pkg = entry.load_package()
    W, H = args.width, args.height
    with_filter = not args.no_filter
    poses = [pkg.orbit_projection(k, W, H) for k in range(args.warmup + args.steps)]
    every = args.time_every if args.time_every >= 0 else (2 if args.steps <= 12 else 4)
    depth_k = args.pipeline if multi else args.frames_in_flight

    def all_ranks(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    class Setup:
        """One workload: `depth_k` contexts holding this rank's slice of a `total`-point cloud."""

        def __init__(self, total, scene, auto_reorder=None):
            self.total, self.scene = total, scene
            self.lo, self.hi = pkg.shard_range(total, rank, world)
            self.projs, self.locals_, self.streams = [], [], []
            for j in range(depth_k):
                pj = pkg.Projector(local_rank)
                if auto_reorder is not None:
                    pj.set_option("auto_reorder", auto_reorder)
                for kv in args.set:
                    key, val = kv.split("=")
                    pj.set_option(key, int(val))
                pj.generate_synthetic(scene, SEEDS["C3"], self.lo, self.hi - self.lo, total)
                pj.set_resolution(W, H)
                if args.overlap and not multi:
                    pj.set_option("overlap", 1)
                lj = pkg.sharded.HipLocal(pj)
                st = torch.cuda.Stream(device=local_rank) if (multi or depth_k > 1) else None
                if st is not None:
                    with torch.cuda.stream(st):
                        lj.bind_stream()  # kernels and RCCL collectives ordered on this stream
                self.projs.append(pj), self.locals_.append(lj), self.streams.append(st)

        def renderers(self, colour, exchange="collective"):
            return [pkg.ShardedProjector(lj, colour=colour, force_exchange=args.force_exchange, exchange=exchange)
                    for lj in self.locals_]

        def sync(self):
            for pj in self.projs:
                pj.synchronize()
            torch.cuda.synchronize()
            if multi:
                dist.barrier()

        def render(self, rs, k, P):
            j = k % depth_k
            if self.streams[j] is None:
                rs[j].render(P, with_filter)
            else:
                with torch.cuda.stream(self.streams[j]):
                    rs[j].render(P, with_filter)

        def timed_run(self, rs, steps, warmup):
            """`warmup` untimed frames, then exactly `steps` timed frames between barrier + device sync."""
            for k in range(warmup):
                self.render(rs, k, poses[k])
            self.sync()
            for pj in self.projs:
                pj.timing_enable(1 if args.time_all_kernels else {0: 0, 1: 2, 2: 4, 4: 3}[every])
                pj.timing_reset()
            t0 = time.perf_counter()
            for k in range(steps):
                self.render(rs, k, poses[warmup + k])
            self.sync()
            dt_ = time.perf_counter() - t0
            if multi:  # the slowest rank's clock counts
                tmax = torch.tensor([dt_], dtype=torch.float64, device="cuda")
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt_ = float(tmax.item())
            timing_ = {}
            for pj in self.projs:
                for name, (ms, cnt) in pj.timing().items():
                    a, b2 = timing_.get(name, (0.0, 0))
                    timing_[name] = (a + ms, b2 + cnt)
                pj.timing_enable(False)
            return dt_, timing_

        def close(self):
            for pj in self.projs:
                pj.close()

    total = args.points * (world if scaling == "weak" else 1)
    S = Setup(total, args.scene)
    proj = S.projs[0]

    # N > 1: check the configured colour form against the plain all-reduce form on pose 0 and
    # fall back (on every rank) if any rank sees a difference
    colour = args.colour if multi else "allreduce"
    if multi and colour != "allreduce":
        ok = 1
        try:
            ref_r, new_r = S.renderers("allreduce"), S.renderers(colour)
            S.render(ref_r, 0, poses[0]); S.sync()
            ref_img = S.locals_[0].image_tensor().clone()
            S.render(new_r, 0, poses[0]); S.sync()
            ok = int(torch.equal(ref_img, S.locals_[0].image_tensor()))
        except Exception as exc:  # noqa: BLE001
            print("rank %d: colour form %s failed (%s), falling back" % (rank, colour, exc), file=sys.stderr)
            ok = 0
        if not all_ranks(ok):
            colour = "allreduce"
    renderers = S.renderers(colour)

    # parity gate on pose 0 at FULL size: the resident cloud is copied back and projected by the
    # multi-thread oracle on the host (bounded by host memory: 20 B per point); the same host copy
    # then feeds the CPU baseline
    parity, cpu, rotated_cloud, parity_poses = None, None, None, None
    if rank == 0 and world == 1 and not multi and total <= 250_000_000 and not (args.no_parity and args.no_cpu_baseline):
        orc = entry.load_oracle()
        xyzw, rgba = proj.download_points()
        if not args.no_parity:
            # three poses spread over the TIMED set (first, middle, last), not only pose 0
            parity_poses = sorted({args.warmup, args.warmup + args.steps // 2, args.warmup + args.steps - 1})
            mt = orc.MTProjector(W, H, host_threads())
            parity = True
            for kp in parity_poses:
                img, depth = proj.project(poses[kp], filtered=with_filter)
                ref = mt.project(xyzw, rgba, poses[kp])
                rd, ri = ref["depth_bits"], ref["img"]
                if with_filter:
                    rf = orc.filter(rd, ri)
                    rd, ri = rf["depth"].view(np.uint32), rf["img"]
                    ok_t = np.array_equal(proj.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])
                else:
                    ok_t = True
                parity = parity and bool(np.array_equal(depth.view(np.uint32), rd) and np.array_equal(img, ri) and ok_t)
            del mt
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(orc, pkg, args, xyzw, rgba)
        if not args.no_extra and args.scene == "room_shell" and not args.overlap:
            # the same cloud as a scanner would deliver it: rotated against the axes (30 / 20 degrees about z / x)
            # and with 1 mm of noise -- no coordinate is constant along a wall any more, so the lossless packing
            # keeps ~9.2 B/pt instead of the synthetic scene's 6.5.  Rendered below through the identically
            # rotated camera (the same views).
            c30, s30, c20, s20 = np.cos(np.pi / 6), np.sin(np.pi / 6), np.cos(np.pi / 9), np.sin(np.pi / 9)
            R = np.array([[1, 0, 0], [0, c20, -s20], [0, s20, c20]]) @ np.array([[c30, -s30, 0], [s30, c30, 0], [0, 0, 1]])
            rng = np.random.default_rng(3)
            for lo in range(0, len(xyzw), 10_000_000):
                blk = xyzw[lo:lo + 10_000_000, :3].astype(np.float64) @ R.T
                blk += rng.normal(scale=1e-3, size=blk.shape)
                xyzw[lo:lo + 10_000_000, :3] = blk.astype(np.float32)
            rotated_cloud = (xyzw, rgba, R)
        del xyzw, rgba

    # N > 1 parity gate (small totals only: the whole cloud is regenerated on rank 0's host)
    if multi and not args.no_parity and total <= 50_000_000:
        S.render(renderers, 0, poses[0])
        S.sync()
        if rank == 0:
            orc = entry.load_oracle()
            xyzw, rgba = orc.generate(args.scene, SEEDS["C3"], 0, total, total)
            ref = orc.MTProjector(W, H, host_threads()).project(xyzw, rgba, poses[0])
            del xyzw, rgba
            rd, ri = ref["depth_bits"], ref["img"]
            if with_filter:
                rf = orc.filter(rd, ri)
                rd, ri = rf["depth"].view(np.uint32), rf["img"]
            parity = bool(np.array_equal(proj.download(pkg._lib.BUF_DEPTH), rd) and
                          np.array_equal(proj.download(pkg._lib.BUF_IMAGE), ri))
            parity_poses = [0]
        S.sync()

    # N > 1 at any size: "multi-GPU result identical to the 1-GPU result" (SURVEY 8d).  Rank 0 holds
    # the WHOLE cloud once more in a separate context and renders pose 0 alone; the sharded frame of
    # the same pose must match bit for bit.
    parity_single = None
    if multi and world > 1 and not args.no_parity and total < (1 << 32) and total <= 1_000_000_000:
        S.render(renderers, 0, poses[0])
        S.sync()
        if rank == 0:
            try:
                whole = pkg.Projector(local_rank)
                whole.generate_synthetic(args.scene, SEEDS["C3"], 0, total, total)
                whole.set_resolution(W, H)
                whole.render(poses[0], with_filter)
                parity_single = bool(
                    np.array_equal(whole.download(pkg._lib.BUF_DEPTH), proj.download(pkg._lib.BUF_DEPTH)) and
                    np.array_equal(whole.download(pkg._lib.BUF_IMAGE), proj.download(pkg._lib.BUF_IMAGE)) and
                    (not with_filter or np.array_equal(whole.download(pkg._lib.BUF_TENSOR),
                                                       proj.download(pkg._lib.BUF_TENSOR))))
                whole.close()
            except Exception as exc:  # noqa: BLE001  (e.g. out of memory): reported, not fatal
                parity_single = "not run: %s" % exc
        S.sync()

    def run_exchange_forms(setup, rs, steps, warmup):
        """The timed run, with the p2p form tried after the collectives when asked for (N > 1)."""
        info = None
        if multi and args.exchange in ("p2p", "owned"):
            rs = setup.renderers(colour, args.exchange)
        dt_, timing_ = setup.timed_run(rs, steps, warmup)
        if multi and args.exchange in ("p2p", "owned"):
            info = {"used": args.exchange if all(r.exchange == args.exchange for r in rs)
                    else "collective (%s dropped)" % args.exchange,
                    "p2p_note": [r.p2p_note for r in rs if r.p2p_note]}
        elif multi and args.exchange == "auto":
            # The hand-written peer-to-peer exchange, tried after the collectives have been measured:
            # each renderer checks its first frame against the collectives on every rank; after the
            # timed frames one more frame is compared and the barrier-timeout words are read.  Only a
            # run that is clean on every rank, and faster, replaces the collectives' number.
            info = {"used": "collective", "collective_ms_per_step": dt_ / steps * 1e3}
            try:
                p2p_r = setup.renderers(colour, "p2p")
                dt2, timing2 = setup.timed_run(p2p_r, steps, warmup)
                clean = all(r.exchange == "p2p" for r in p2p_r) and all(pj.p2p_timeouts() == 0 for pj in setup.projs)
                if clean:
                    k_chk = warmup + steps - 1
                    setup.render(p2p_r, 0, poses[k_chk]); setup.sync()
                    d_p, i_p = setup.locals_[0].depth_tensor().clone(), setup.locals_[0].image_tensor().clone()
                    setup.render(rs, 0, poses[k_chk]); setup.sync()
                    clean = bool(torch.equal(d_p, setup.locals_[0].depth_tensor()) and
                                 torch.equal(i_p, setup.locals_[0].image_tensor()))
                note = [r.p2p_note for r in p2p_r if r.p2p_note]
            except Exception as exc:  # noqa: BLE001
                clean, dt2, timing2, note = False, None, None, ["%s" % exc]
            clean = all_ranks(clean)
            info["p2p_clean_on_all_ranks"] = clean
            if dt2 is not None:
                info["p2p_ms_per_step"] = dt2 / steps * 1e3
            if note:
                info["p2p_note"] = note
            if clean and dt2 is not None and dt2 < dt_:  # both are max-over-ranks: every rank decides alike
                dt_, timing_ = dt2, timing2
                info["used"] = "p2p"
            # The owner-computes form: no MIN / SUM exchange, every tile produced once; the frame's owner rotates over
            # the ranks (rank k mod N ends with frame k).  Checked the same way, with rank 0 owning the checked frame.
            try:
                own_r = setup.renderers(colour, "owned")
                dt3, timing3 = setup.timed_run(own_r, steps, warmup)
                clean3 = all(r.exchange == "owned" for r in own_r) and all(pj.p2p_timeouts() == 0 for pj in setup.projs)
                if clean3:
                    k_chk = warmup + steps - 1
                    for r in own_r:
                        r.fixed_owner = 0
                    setup.render(own_r, 0, poses[k_chk]); setup.sync()
                    d_o, i_o = setup.locals_[0].depth_tensor().clone(), setup.locals_[0].image_tensor().clone()
                    setup.render(rs, 0, poses[k_chk]); setup.sync()
                    clean3 = rank != 0 or bool(torch.equal(d_o, setup.locals_[0].depth_tensor()) and
                                               torch.equal(i_o, setup.locals_[0].image_tensor()))
                note3 = [r.p2p_note for r in own_r if r.p2p_note]
            except Exception as exc:  # noqa: BLE001
                clean3, dt3, timing3, note3 = False, None, None, ["%s" % exc]
            clean3 = all_ranks(clean3)
            info["owned_clean_on_all_ranks"] = clean3
            if dt3 is not None:
                info["owned_ms_per_step"] = dt3 / steps * 1e3
            if note3:
                info["owned_note"] = note3
            # (never the headline: in this form only ONE rank ends with each frame -- rotating-owner throughput, not
            # like-for-like with the two forms above, where every rank holds every frame.  --exchange owned asks for it
            # explicitly, and the line then says so in `value_semantics`.)
            info["owned_semantics"] = "rotating-owner throughput: frame k is complete on rank k mod N only"
        return dt_, timing_, info

    def kernel_table(timing_):
        return {k: (ms / max(n, 1)) for k, (ms, n) in timing_.items() if n}

    def sample_stats(pj, pose_list, filt):
        """Frame statistics (in-frustum entries, chunks whose colours were loaded) of the point kernel, averaged over
        a sample of the timed poses: what `roofline.bytes_model` is computed from.  Tile-binned form only."""
        if not pj.get_option("mode") or not pose_list:
            return None
        ent = col = 0
        sample = pose_list[::max(1, len(pose_list) // 25)]
        for P in sample:
            pj.render(P, filt)
            st = pj.frame_stats()
            if st["errors"]:
                sys.exit("tile store error %d in a bench frame" % st["errors"])
            ent, col = ent + st["entries"], col + st["colour_chunks"]
        return {"entries": ent / len(sample), "colour_chunks": col / len(sample), "frames_sampled": len(sample)}

    timed_poses = poses[args.warmup:args.warmup + args.steps]

    dt, timing, exchange_info = run_exchange_forms(S, renderers, args.steps, args.warmup)
    stream_bpp = S.projs[0].get_option("packed_millibytes_per_point") / 1000.0  # 12.0 unless the cloud is packed
    n_local = S.hi - S.lo
    main_stats = sample_stats(S.projs[0], timed_poses, with_filter)  # (the local slice's own frames)
    if multi:
        S.sync()

    # Reported separately, N = 1: the layout north_star names -- fp32 SoA xyz (option pack = 0: the point kernel
    # streams 12 B/pt) -- over the same poses, with its own roofline object
    fp32_soa = None
    if not multi and not args.no_extra and stream_bpp < 11.9:
        proj.set_option("pack", 0)
        dtf, tf = S.timed_run(renderers, args.steps, args.warmup)
        kf = kernel_table(tf)
        stf = sample_stats(proj, timed_poses, with_filter)
        trf, limf, headf = measured_traffic(args.scene, n_local, W, H, with_filter, pack=0)
        fp32_soa = {"what": "option pack = 0: the point kernel streams the fp32 SoA coordinates (12 B/pt, the layout "
                            "BASELINE.json's north_star names) instead of their lossless packed form; same poses, "
                            "bit-identical frames",
                    "value": total * args.steps / dtf / 1e6, "unit": "Mpoints/s", "ms_per_step": dtf / args.steps * 1e3,
                    "steps": args.steps,
                    "roofline": roofline_of(kf["min_depth"], tf["min_depth"][1], n_local, trf, every, 12.0, stf, limf, headf)
                                if kf.get("min_depth") else None}
        proj.set_option("pack", 1)

    # Reported separately, N = 1: the reference's CALL SHAPE -- computeFilteredRGBD copies depth (W*H*4) and colour
    # (W*H*3) into caller-allocated host arrays every frame (project_cloud.cu:302-309,424-431) -- over the same poses.
    # Never `value` (inputs and outputs of `value` stay in HBM); this is the PCIe-inclusive rate.
    host_out = None
    if not multi and not args.no_extra:
        img_h, depth_h = np.empty((H, W, 3), np.uint8), np.empty((H, W), np.float32)
        img_h[...] = 0
        depth_h[...] = 0  # (pages touched before the timed region)
        for k in range(args.warmup):
            proj.project_into(poses[k], img_h, depth_h, with_filter)
        t1 = time.perf_counter()
        for k in range(args.steps):
            proj.project_into(poses[args.warmup + k], img_h, depth_h, with_filter)
        dth = time.perf_counter() - t1
        host_out = {"what": "rtr_project%s into caller-allocated (pageable) host arrays, synchronous per frame: the "
                            "reference's call shape, %d bytes over PCIe per frame" %
                            ("_filtered" if with_filter else "", img_h.nbytes + depth_h.nbytes),
                    "value": total * args.steps / dth / 1e6, "unit": "Mpoints/s", "ms_per_step": dth / args.steps * 1e3,
                    "steps": args.steps, "bytes_per_frame": img_h.nbytes + depth_h.nbytes}
        if hasattr(proj, "project_async"):
            # the same frames through the asynchronous pair: frame k's copies overlap frame k + 1's kernels
            outs = [proj.host_output_buffers(j) for j in range(2)]
            for k in range(args.warmup):
                proj.project_async(poses[k], k & 1, with_filter)
            proj.wait_outputs()
            t1 = time.perf_counter()
            for k in range(args.steps):
                proj.project_async(poses[args.warmup + k], k & 1, with_filter)
            proj.wait_outputs()
            dta = time.perf_counter() - t1
            ok = None
            if not args.no_parity:  # the last frame, against the synchronous call
                proj.project_into(poses[args.warmup + args.steps - 1], img_h, depth_h, with_filter)
                io, do = outs[(args.steps - 1) & 1]
                ok = bool(np.array_equal(io, img_h) and np.array_equal(do.view(np.uint32), depth_h.view(np.uint32)))
            host_out["async"] = {"what": "rtr_project_async into the library's pinned output buffers (two in rotation), "
                                         "one rtr_wait at the end: frame k's device-to-host copies run beside frame "
                                         "k + 1's kernels", "ms_per_step": dta / args.steps * 1e3,
                                 "value": total * args.steps / dta / 1e6, "unit": "Mpoints/s",
                                 "equals_sync_call": ok}

    # BASELINE config C2 (N = 1): ~1e7 points -> 1920x1080, z-buffer 1x1 splat only (ScanNet++ is not in the
    # container: the synthetic room_shell stand-in with C2's seed), with its own parity check
    c2 = None
    if not multi and not args.no_extra and (args.points, W, H) == (100_000_000, 1920, 1080):
        n2 = 10_000_000
        pc = pkg.Projector(local_rank)
        pc.generate_synthetic("room_shell", SEEDS["C2"], 0, n2, n2)
        pc.set_resolution(W, H)
        for k in range(args.warmup):
            pc.render(poses[k], False)
        pc.synchronize()
        pc.timing_enable({0: 0, 1: 2, 2: 4, 4: 3}[every])
        pc.timing_reset()
        t1 = time.perf_counter()
        for k in range(args.steps):
            pc.render(poses[args.warmup + k], False)
        pc.synchronize()
        dt2 = time.perf_counter() - t1
        t2 = pc.timing()
        pc.timing_enable(False)
        k2 = kernel_table(t2)
        ok2 = None
        if not args.no_parity:
            orc = entry.load_oracle()
            x2, c2c = pc.download_points()
            i2, d2 = pc.project(poses[0])
            r2 = orc.MTProjector(W, H, host_threads()).project(x2, c2c, poses[0])
            ok2 = bool(np.array_equal(d2.view(np.uint32), r2["depth_bits"]) and np.array_equal(i2, r2["img"]))
            del x2, c2c, r2
        bpp2 = pc.get_option("packed_millibytes_per_point") / 1000.0
        c2 = {"what": "BASELINE C2: %d-point room_shell (seed 0xC0FFEE02) -> %dx%d, projection only (no prefilter)" % (n2, W, H),
              "value": n2 * args.steps / dt2 / 1e6, "unit": "Mpoints/s", "ms_per_step": dt2 / args.steps * 1e3,
              "frames_per_s": args.steps / dt2, "steps": args.steps, "parity_vs_oracle": ok2,
              "roofline": roofline_of(k2["min_depth"], t2["min_depth"][1], n2, None, every, bpp2,
                                      sample_stats(pc, timed_poses, False)) if k2.get("min_depth") else None}
        pc.close()

    # BASELINE config C1 (N = 1): a 100k-point synthetic .ply -> 640x480 through the NAIVE single-thread host loop
    # (plumbing): the cloud is written as a binary-LE .ply (float x, y, z + uchar red, green, blue:
    # the layout cloudreader.cpp:140-170 reads), read back, put through the loader's 0.25 m grid and flattened the way
    # the reference hands it to ProjectCloud (Octreegrid.h:162-180); the oracle's single-thread projector is timed on
    # it, and the GPU renders the same arrays (C1 itself names no GPU: its frame is the parity check of the plumbing).
    c1 = None
    if not multi and not args.no_extra and not args.no_cpu_baseline:
        import tempfile
        orc = entry.load_oracle()
        F = pkg.formats
        n1, W1, H1 = 100_000, 640, 480
        x1, col1 = orc.generate("room_shell", 0xC0FFEE01, 0, n1, n1)
        with tempfile.TemporaryDirectory() as td:
            ply = os.path.join(td, "c1.ply")
            F.write_ply(ply, x1[:, :3], col1[:, :3])
            g1 = F.compute_grid(*F.read_ply(ply))
        v1, k1 = g1.vertex_positions(), g1.vertex_colors()
        poses1 = [pkg.orbit_projection(k, W1, H1) for k in range(64)]
        orc.project(v1, k1, poses1[0], W1, H1)
        t1 = time.perf_counter()
        frames1 = 0
        while frames1 < 50 and (frames1 < 3 or time.perf_counter() - t1 < 2.0):
            ref1 = orc.project(v1, k1, poses1[frames1], W1, H1)
            frames1 += 1
        dt1 = time.perf_counter() - t1
        p1 = pkg.Projector(local_rank)
        p1.set_resolution(W1, H1)
        p1.upload_points(v1, k1)
        i1, d1 = p1.project(poses1[frames1 - 1])
        for k in range(5):
            p1.render(poses1[k], False)
        p1.synchronize()
        t1 = time.perf_counter()
        for k in range(50):
            p1.render(poses1[k], False)
        p1.synchronize()
        dg1 = time.perf_counter() - t1
        p1.close()
        c1 = {"what": "BASELINE C1: %d-point room_shell (seed 0xC0FFEE01) written to a binary .ply, read back, gridded in "
                      "0.25 m blocks and flattened like the reference's loader -> %dx%d; the oracle's NAIVE single-thread "
                      "host loop timed, the GPU frame of the same arrays as the check of the plumbing" % (n1, W1, H1),
              "cpu_naive": {"value": n1 * frames1 / dt1 / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "port",
                            "ms_per_frame": dt1 / frames1 * 1e3, "frames": frames1},
              "gpu": {"value": n1 * 50 / dg1 / 1e6, "unit": "Mpoints/s", "ms_per_step": dg1 / 50 * 1e3, "steps": 50},
              "blocks": int(len(g1.keys)),
              "parity_vs_oracle": bool(np.array_equal(d1.view(np.uint32), ref1["depth_bits"]) and
                                       np.array_equal(i1, ref1["img"]))}
        del x1, col1, v1, k1, g1

    # Reported separately (never part of `value`), N = 1 only.
    extra_cull, ubox, pipelined, rotated = None, None, None, None
    if not multi and not args.no_extra and not args.overlap:
        # (0) the same frames with library option "overlap": the point stream of frame k+1 is queued on a second
        #     HIP stream and starts while the tile kernel / prefilter of frame k still run
        proj.set_option("overlap", 1)
        m = args.steps  # (the same poses as the headline run: T1 depends on the pose)
        dtp, tp = S.timed_run(renderers, m, args.warmup)
        proj.set_option("overlap", 0)
        pipelined = {"what": "option overlap = 1: T1 of frame k+1 on a second stream beside the tail of frame k (two tile "
                             "stores); same frames, bit-identical output.  Not the headline: beside the tail the point "
                             "kernel's own launches stretch, so its roofline figure would no longer describe the kernel",
                     "value": total * m / dtp / 1e6, "unit": "Mpoints/s", "ms_per_step": dtp / m * 1e3, "steps": m,
                     "min_depth_avg_launch_ms": kernel_table(tp).get("min_depth")}
    if not multi and not args.no_extra:
        # (1) the same frames with exact per-chunk frustum culling on a Morton-sorted cloud: an algorithmic
        #     byte reduction, not a roofline claim
        if not proj.get_option("reordered"):
            proj.reorder_points()
        proj.set_option("cull", 1)
        for k in range(min(args.warmup, 5)):
            proj.render(poses[k], with_filter)
        proj.synchronize()
        m = args.steps  # (the same poses as the headline run: T1 depends on the pose)
        t1 = time.perf_counter()
        for k in range(m):
            proj.render(poses[args.warmup + k], with_filter)
        proj.synchronize()
        dte = time.perf_counter() - t1
        proj.set_option("cull", 0)
        extra_cull = {"what": "Morton-reordered cloud + exact 256-point-chunk frustum culling (option cull=1); same "
                              "frames, bit-identical output; an algorithmic byte reduction, not a roofline claim",
                      "value": total * m / dte / 1e6, "unit": "Mpoints/s", "ms_per_step": dte / m * 1e3, "steps": m}
        # (1b) the scene as a scanner would deliver it (rotated against the axes + 1 mm noise), same views
        if rotated_cloud is not None:
            xr, cr, R = rotated_cloud
            rotated_cloud = None
            T = np.eye(4)
            T[:3, :3] = R.T  # P' X' = P R^T (R X) = P X
            poses_r = [np.ascontiguousarray((np.asarray(P, np.float64).reshape(4, 4) @ T).astype(np.float32).reshape(16))
                       for P in poses]
            pr = pkg.Projector(local_rank)
            pr.set_resolution(W, H)
            pr.upload_points(xr, cr)
            for k in range(args.warmup):
                pr.render(poses_r[k], with_filter)
            pr.synchronize()
            m = args.steps
            pr.timing_enable({0: 0, 1: 2, 2: 4, 4: 3}[every])
            pr.timing_reset()
            t1 = time.perf_counter()
            for k in range(m):
                pr.render(poses_r[args.warmup + k], with_filter)
            pr.synchronize()
            dtr = time.perf_counter() - t1
            tr = pr.timing()
            pr.timing_enable(False)
            kr = kernel_table(tr)
            img_r, depth_r = pr.project(poses_r[0])
            ref_r = entry.load_oracle().MTProjector(W, H, host_threads()).project(xr, cr, poses_r[0])
            rotated = {"what": "the same 1e8-point cloud rotated 30 / 20 degrees about z / x with 1 mm of noise, rendered "
                               "through the identically rotated camera (same views): no coordinate is constant along a "
                               "wall any more, which is what the synthetic scene's 6.5 B/pt owe a third of their saving to",
                       "value": total * m / dtr / 1e6, "unit": "Mpoints/s", "ms_per_step": dtr / m * 1e3, "steps": m,
                       "packed_bytes_per_point": pr.get_option("packed_millibytes_per_point") / 1000.0,
                       "parity_vs_oracle": bool(np.array_equal(depth_r.view(np.uint32), ref_r["depth_bits"]) and
                                                np.array_equal(img_r, ref_r["img"])),
                       "roofline": roofline_of(kr["min_depth"], tr["min_depth"][1], total, None, every,
                                               pr.get_option("packed_millibytes_per_point") / 1000.0,
                                               sample_stats(pr, poses_r[args.warmup:args.warmup + m], with_filter))
                                   if kr.get("min_depth") else None}
            pr.close()
            del xr, cr, ref_r
        # (2) the incoherent stress scene of SURVEY.md 8d, as handed over and under the default upload policy
        if args.scene == "room_shell":
            S.close()
            S = None
            ubox = {"what": "uniform_box: 100 M points in hash order (consecutive points are unrelated): every wave takes "
                            "the exact path and claims stream positions per point"}
            m = args.steps  # (the same poses as the headline run: T1 depends on the pose)
            for key, policy in (("as_uploaded", 0), ("default_upload_policy", None)):
                U = Setup(args.points, "uniform_box", auto_reorder=policy)
                t_up = None
                if policy is None:  # what the one-off sort costs, measured on a second generation
                    t0 = time.perf_counter()
                    U.projs[0].generate_synthetic("uniform_box", SEEDS["C3"], 0, args.points, args.points)
                    U.projs[0].synchronize()
                    t_up = time.perf_counter() - t0
                dtu, tu = U.timed_run(U.renderers("allreduce"), m, args.warmup)
                ku = kernel_table(tu)
                stu = sample_stats(U.projs[0], timed_poses, with_filter)
                tru, limu, headu = measured_traffic("uniform_box" if policy == 0 else "uniform_box_sorted", args.points, W, H,
                                                    with_filter)
                req, two = frame_bytes(args.points, W, H, with_filter)
                ubox[key] = {"value": args.points * m / dtu / 1e6, "unit": "Mpoints/s", "ms_per_step": dtu / m * 1e3,
                             "steps": m, "reordered_by_library": bool(U.projs[0].get_option("reordered")),
                             "order_ratio": U.projs[0].get_option("order_ratio_ppm") / 1e6,
                             "roofline": roofline_of(ku["min_depth"], tu["min_depth"][1], args.points, tru, every,
                                                     U.projs[0].get_option("packed_millibytes_per_point") / 1000.0, stu,
                                                     limu, headu),
                             "frame_required_bytes_frac": req / (dtu / m) / 1e9 / HBM_PEAK_GBS}
                if t_up is not None:
                    ubox[key]["generate_plus_sort_s"] = t_up
                U.close()

    # N > 1: the weak-scaling figure (--points per GPU), a shorter run of the same code path
    weak = None
    if multi and world > 1 and scaling == "strong" and not args.no_weak:
        try:
            S.close()
            S = None
            Wk = Setup(args.points * world, args.scene)
            m = min(args.steps, 30)
            dtw, _, infw = run_exchange_forms(Wk, Wk.renderers(colour), m, min(args.warmup, 5))
            weak = {"value": args.points * world * m / dtw / 1e6, "unit": "Mpoints/s", "ms_per_step": dtw / m * 1e3,
                    "steps": m, "points_total": args.points * world, "points_per_gpu": args.points,
                    "exchange": (infw or {}).get("used")}
            Wk.close()
        except Exception as exc:  # noqa: BLE001
            weak = "not run: %s" % exc

    if rank == 0:
        kern = kernel_table(timing)
        dom = max(("min_depth", "accumulate"), key=lambda k: kern.get(k, 0.0))
        required, two_pass = frame_bytes(n_local, W, H, with_filter)
        step_s = dt / args.steps
        main_traffic = measured_traffic(args.scene, n_local, W, H, with_filter, pack=1 if stream_bpp < 11.9 else 0)
        out = {
            "metric": "Mpoints/sec projected + frames/sec at %dx%d" % (W, H),
            "value": total * args.steps / dt / 1e6,
            "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "frames_per_s": args.steps / dt,
            "higher_is_better": True,
            "scaling": scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %d-point synthetic %s cloud -> %dx%d, 1x1 splat z-buffer%s%s"
                                   % (("C3" if (args.points, W, H, world) == (100_000_000, 1920, 1080, 1) and with_filter else
                                       ("C4" if (args.points, W, H) == (100_000_000, 1920, 1080) and scaling == "strong"
                                        and world > 1 else "custom")),
                                      total, args.scene, W, H, " + depth-heuristic prefilter" if with_filter else "",
                                      (" sharded %d ways" % world) if world > 1 else ""),
                       "points_total": total, "points_per_gpu": n_local, "scene": args.scene,
                       "resolution": [W, H], "prefilter": with_filter,
                       **({"options": args.set} if args.set else {}),
                       "parallelism": (("point-shard x%d, owner-computes tiles over hipIpc-mapped tile stores (no MIN / SUM "
                                        "exchange; the frame's owner rotates), %d frames in flight" % (world, depth_k))
                                       if (exchange_info or {}).get("used") == "owned" else
                                       ("point-shard x%d, hand-written peer-to-peer exchange over hipIpc-mapped buffers: "
                                        "MIN(depth), SUM(accum) + slice resolve, %d frames in flight" % (world, depth_k))
                                       if (exchange_info or {}).get("used") == "p2p" else
                                       ("point-shard x%d, %s all-reduce MIN(depth) + %s SUM(accum), %d frames in "
                                        "flight" % (world, "RCCL" if args.backend == "nccl" else "gloo", colour,
                                                    depth_k))) if multi else
                       ("single GPU" + (", %d frames in flight" % depth_k if depth_k > 1 else ""))},
            "roofline": roofline_of(kern[dom], timing[dom][1], n_local, main_traffic[0], every, stream_bpp,
                                    main_stats if dom == "min_depth" else None, main_traffic[1], main_traffic[2])
                        if kern.get(dom) else None,
            # the frame against the bytes THIS design has to move (cloud streamed once: 12 B/pt, 39 B/px of
            # clear / resolve work, ~50 B/px of prefilter) ...
            "frame_required_bytes": required,
            "frame_required_bytes_frac": required / step_s / 1e9 / HBM_PEAK_GBS,
            "frame_required_bytes_note": "SURVEY.md 8d's ALGORITHMIC bytes (12 B per point + 39 B per pixel, + the prefilter's) over "
                                         "the frame's time and 8 TB/s: a comparison with an ideal fp32 stream, not a roofline "
                                         "fraction -- above 1 when the frame reads fewer bytes than that (lossless packed "
                                         "coordinates of which the lane test reads a quarter); roofline.frac is the measured one",
            # ... the same with the coordinate stream at its RESIDENT size (packed clouds: what HBM really delivers)
            "frame_resident_bytes": required - (12.0 - stream_bpp) * n_local,
            "frame_resident_bytes_frac": (required - (12.0 - stream_bpp) * n_local) / step_s / 1e9 / HBM_PEAK_GBS,
            # ... and against SURVEY.md 8d's figure for the reference's two-pass structure (24 B/pt + 39 B/px;
            # its acceptance line is >= 0.70).  Not a roofline fraction: a one-pass frame can exceed 1.
            "vs_two_pass_bytes": two_pass / step_s / 1e9 / HBM_PEAK_GBS,
            "kernel_ms": kern,
            "parity_vs_oracle": parity,
            "parity_poses": parity_poses,
            "parity_vs_single_gpu": parity_single,
            "c1": c1,
            "fp32_soa": fp32_soa,
            "c2": c2,
            "host_outputs": host_out,
            "rotated_noisy_scene": rotated,
            "uniform_box": ubox,
            "pipelined": pipelined,
            "with_chunk_culling": extra_cull,
        }
        if world > 1:
            out["weak_scaling"] = weak
            out["multi_gpu_note"] = ("the driver's N = 1, 2, 4, 8 runs of this script are the only scaling curve: the "
                                     "builder has no multi-GPU box (peer-to-peer kernels rehearsed with several "
                                     "processes on ONE GPU)")
        if exchange_info is not None:
            out["exchange"] = exchange_info
            if exchange_info.get("used") == "owned":
                out["value_semantics"] = "rotating-owner throughput: frame k is complete on rank k mod N only"
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if S is not None:
        S.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
