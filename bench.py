#!/usr/bin/env python3
"""bench.py -- Mpoints/s projected + frames/s of the point-cloud -> framebuffer hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one frame of BASELINE.json config C3: a 100 M-point synthetic cloud projected
to 1920x1080 (clear, min-depth pass, accumulate pass, resolve) plus the depth-heuristic
prefilter, one distinct camera pose of the orbit trajectory per frame.  The cloud is
synthesised on-device before the timed region (inputs resident in HBM).  With N > 1 the
cloud is sharded in contiguous point slices over the ranks and the depth / accumulator
buffers are MIN / SUM all-reduced over RCCL: by default every rank keeps 100 M points
(N x 100 M in total, "scaling": "weak" -- per-frame pixel work and the exchanged buffers
do not shrink with N, so a fixed 100 M-point cloud split 8 ways, `--scaling strong` =
BASELINE config C4, is bound by them, not by the projector).

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
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="N>1: weak = --points per GPU (default; C5-style growth), strong = --points total (BASELINE C4)")
    ap.add_argument("--colour", default="reduce_scatter", choices=["allreduce", "reduce_scatter"],
                    help="N>1: how the colour accumulators are merged (see sharded.py)")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2, 3],
                    help="N>1: frames in flight per rank (2 = frame k's RCCL exchange overlaps frame k+1's kernels)")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2, 3],
                    help="N=1: independent frames alternate between this many contexts / HIP streams, so the "
                         "latency-bound tail of frame k (tile sort, tile z-buffer, prefilter) overlaps the "
                         "bandwidth-bound stream of frame k+1")
    ap.add_argument("--exchange", default="auto", choices=["auto", "collective", "p2p"],
                    help="N > 1: 'collective' = torch.distributed (RCCL) on the library's buffers; 'p2p' = the library's "
                         "hand-written exchange over hipIpc-mapped peer buffers; 'auto' (default) times the collectives, "
                         "then the p2p form (verified against the collectives before and after), and reports the faster")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + several ranks on ONE GPU is a rehearsal of the N>1 logic")
    ap.add_argument("--force-exchange", action="store_true",
                    help="rehearsal on ONE GPU: run the N>1 code path (streams, RCCL collectives in a 1-rank "
                         "group, 2 frames in flight); the numbers are not a benchmark result")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="bracket every phase with HIP events (default: only the dominant streaming kernel, two "
                         "event records per frame; the full per-kernel table is in profiles/)")
    ap.add_argument("--no-extra", action="store_true", help="skip the separately reported chunk-culling measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-points", type=int, default=10_000_000)
    ap.add_argument("--cpu-frames", type=int, default=0, help="0 = pick so the leg takes about 10-20 s")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size oracle parity gate on pose 0")
    return ap.parse_args()


def host_threads():
    """Threads for the CPU leg: the process's CPU share, capped at 16 (each thread owns a
    private 1080p frame buffer; more threads only grow the merge)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline(orc, pkg, args):
    """The oracle's multi-thread projector (kind 'port') on a bounded sample of the same
    workload: a cpu-points instance of the same scene, same resolution and trajectory."""
    cores = host_threads()
    n, W, H = args.cpu_points, args.width, args.height
    xyzw, rgba = orc.generate(args.scene, SEEDS["C3"], 0, n, n)
    mt = orc.MTProjector(W, H, cores)
    mt.project(xyzw, rgba, pkg.orbit_projection(0, W, H))  # warm-up (page faults)
    t0 = time.perf_counter()
    mt.project(xyzw, rgba, pkg.orbit_projection(1, W, H))
    one = time.perf_counter() - t0
    frames = args.cpu_frames or max(2, min(50, int(12.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for k in range(frames):
        mt.project(xyzw, rgba, pkg.orbit_projection(2 + k, W, H))
    dt = time.perf_counter() - t0
    return {"value": n * frames / dt / 1e6, "unit": "Mpoints/s", "cores": cores, "kind": "port",
            "sample": "%d frames of a %d-point %s cloud -> %dx%d (projection only, oracle multi-thread "
                      "projector, %d threads)" % (frames, n, args.scene, W, H, cores),
            "frames_per_s": frames / dt}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
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

    pkg = entry.load_package()
    W, H = args.width, args.height
    total = args.points * (world if args.scaling == "weak" else 1)
    lo, hi = pkg.shard_range(total, rank, world)
    with_filter = not args.no_filter
    poses = [pkg.orbit_projection(k, W, H) for k in range(args.warmup + args.steps)]

    # one context per frame in flight: with N > 1 two frames alternate between two contexts on
    # two HIP streams, so the RCCL exchange of frame k overlaps the kernels of frame k + 1
    depth_k = args.pipeline if multi else args.frames_in_flight
    projs, locals_, streams = [], [], []
    for j in range(depth_k):
        pj = pkg.Projector(local_rank)
        pj.generate_synthetic(args.scene, SEEDS["C3"], lo, hi - lo, total)
        pj.set_resolution(W, H)
        lj = pkg.sharded.HipLocal(pj)
        st = torch.cuda.Stream(device=local_rank) if (multi or depth_k > 1) else None
        if st is not None:
            with torch.cuda.stream(st):
                lj.bind_stream()  # kernels and RCCL collectives ordered on this stream
        projs.append(pj), locals_.append(lj), streams.append(st)
    proj = projs[0]

    def make_renderers(colour, exchange="collective"):
        return [pkg.ShardedProjector(lj, colour=colour, force_exchange=args.force_exchange, exchange=exchange)
                for lj in locals_]

    def sync():
        for pj in projs:
            pj.synchronize()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()

    def render(renderers, k, P):
        j = k % depth_k
        if streams[j] is None:
            renderers[j].render(P, with_filter)
        else:
            with torch.cuda.stream(streams[j]):
                renderers[j].render(P, with_filter)

    # N > 1: check the configured colour form against the plain all-reduce form on pose 0 and
    # fall back (on every rank) if any rank sees a difference
    colour = args.colour if multi else "allreduce"
    if multi and colour != "allreduce":
        ok = 1
        try:
            ref_r, new_r = make_renderers("allreduce"), make_renderers(colour)
            render(ref_r, 0, poses[0]); sync()
            ref_img = locals_[0].image_tensor().clone()
            render(new_r, 0, poses[0]); sync()
            ok = int(torch.equal(ref_img, locals_[0].image_tensor()))
        except Exception as exc:  # noqa: BLE001
            print("rank %d: colour form %s failed (%s), falling back" % (rank, colour, exc), file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            colour = "allreduce"
    renderers = make_renderers(colour)

    # parity gate on pose 0 at FULL size: the resident cloud is copied back and projected by the
    # multi-thread oracle on the host (bounded by host memory: 20 B per point)
    parity = None
    if rank == 0 and not args.no_parity and world == 1 and not multi and total <= 250_000_000:
        orc = entry.load_oracle()
        xyzw, rgba = proj.download_points()
        img, depth = proj.project(poses[0], filtered=with_filter)
        ref = orc.MTProjector(W, H, host_threads()).project(xyzw, rgba, poses[0])
        del xyzw, rgba
        rd, ri = ref["depth_bits"], ref["img"]
        if with_filter:
            rf = orc.filter(rd, ri)
            rd, ri = rf["depth"].view(np.uint32), rf["img"]
            ok_t = np.array_equal(proj.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])
        else:
            ok_t = True
        parity = bool(np.array_equal(depth.view(np.uint32), rd) and np.array_equal(img, ri) and ok_t)

    # N > 1 parity gate (small totals only: the whole cloud is regenerated on rank 0's host)
    if multi and not args.no_parity and total <= 50_000_000:
        render(renderers, 0, poses[0])
        sync()
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
        sync()

    # N > 1 at any size: "multi-GPU result identical to the 1-GPU result" (SURVEY 8d).  Rank 0 holds
    # the WHOLE cloud once more in a separate context (16 B/pt + lists; 8e8 points are ~45 GB of
    # 288) and renders pose 0 alone; the sharded frame of the same pose must match bit for bit.
    parity_single = None
    if multi and world > 1 and not args.no_parity and total < (1 << 32) and total <= 1_000_000_000:
        render(renderers, 0, poses[0])
        sync()
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
        sync()

    def timed_run(rs):
        """W warm-up frames, then exactly K timed frames between barrier + device sync."""
        for k in range(args.warmup):
            render(rs, k, poses[k])
        sync()
        for pj in projs:
            pj.timing_enable(1 if args.time_all_kernels else 3)  # 3: the dominant kernel, every 4th frame
            pj.timing_reset()
        t0 = time.perf_counter()
        for k in range(args.steps):
            render(rs, k, poses[args.warmup + k])
        sync()
        dt_ = time.perf_counter() - t0
        if multi:  # the slowest rank's clock counts
            tmax = torch.tensor([dt_], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        timing_ = {}
        for pj in projs:
            for name, (ms, cnt) in pj.timing().items():
                a, b2 = timing_.get(name, (0.0, 0))
                timing_[name] = (a + ms, b2 + cnt)
            pj.timing_enable(False)
        return dt_, timing_

    def all_ranks(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    exchange_info = None
    if multi and args.exchange == "p2p":
        renderers = make_renderers(colour, "p2p")
    dt, timing = timed_run(renderers)
    if multi and args.exchange == "p2p":
        exchange_info = {"used": "p2p" if all(r.exchange == "p2p" for r in renderers) else "collective (p2p dropped)",
                         "p2p_note": [r.p2p_note for r in renderers if r.p2p_note]}
    elif multi and args.exchange == "auto":
        # The hand-written peer-to-peer exchange, tried after the collectives have been measured:
        # each renderer checks its first frame against the collectives on every rank; after the
        # timed frames one more frame is compared and the barrier-timeout words are read.  Only a
        # run that is clean on every rank, and faster, replaces the collectives' number.
        exchange_info = {"used": "collective", "collective_ms_per_step": dt / args.steps * 1e3}
        try:
            p2p_r = make_renderers(colour, "p2p")
            dt2, timing2 = timed_run(p2p_r)
            clean = all(r.exchange == "p2p" for r in p2p_r) and all(pj.p2p_timeouts() == 0 for pj in projs)
            if clean:
                k_chk = args.warmup + args.steps - 1
                render(p2p_r, 0, poses[k_chk]); sync()
                d_p, i_p = locals_[0].depth_tensor().clone(), locals_[0].image_tensor().clone()
                render(renderers, 0, poses[k_chk]); sync()
                clean = bool(torch.equal(d_p, locals_[0].depth_tensor()) and torch.equal(i_p, locals_[0].image_tensor()))
            note = [r.p2p_note for r in p2p_r if r.p2p_note]
        except Exception as exc:  # noqa: BLE001
            clean, dt2, timing2, note = False, None, None, ["%s" % exc]
        clean = all_ranks(clean)
        exchange_info["p2p_clean_on_all_ranks"] = clean
        if dt2 is not None:
            exchange_info["p2p_ms_per_step"] = dt2 / args.steps * 1e3
        if note:
            exchange_info["p2p_note"] = note
        if clean and dt2 is not None and dt2 < dt:  # both are max-over-ranks: every rank decides alike
            dt, timing = dt2, timing2
            exchange_info["used"] = "p2p"

    # Reported separately (never part of `value`): the same frames with the one-off Morton
    # reorder + exact per-chunk frustum culling ("cull"), an algorithmic byte reduction.
    extra = None
    if not multi and not args.no_extra:
        proj.reorder_points()
        proj.set_option("cull", 1)
        for k in range(min(args.warmup, 5)):
            proj.render(poses[k], with_filter)
        proj.synchronize()
        m = min(args.steps, 50)
        t1 = time.perf_counter()
        for k in range(m):
            proj.render(poses[args.warmup + k], with_filter)
        proj.synchronize()
        dte = time.perf_counter() - t1
        proj.set_option("cull", 0)
        extra = {"what": "Morton-reordered cloud + exact 256-point-chunk frustum culling (option cull=1); same frames, "
                         "bit-identical output; an algorithmic byte reduction, not a roofline claim",
                 "value": total * m / dte / 1e6, "unit": "Mpoints/s", "ms_per_step": dte / m * 1e3, "steps": m}

    if rank == 0:
        n_local = hi - lo
        kern = {k: (ms / max(n, 1)) for k, (ms, n) in timing.items() if n}
        dom = max(("min_depth", "accumulate"), key=lambda k: kern.get(k, 0.0))
        dom_ms = kern[dom]
        achieved = 12.0 * n_local / (dom_ms * 1e-3) / 1e9  # GB/s: 12 B/pt streamed per pass
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.scene, {}).get(dom)
            except Exception:
                traffic = None
        frame_bytes = 24.0 * n_local + 39.0 * W * H
        out = {
            "metric": "Mpoints/sec projected + frames/sec at %dx%d" % (W, H),
            "value": total * args.steps / dt / 1e6,
            "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "frames_per_s": args.steps / dt,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %d-point synthetic %s cloud -> %dx%d, 1x1 splat z-buffer%s"
                                   % (("C3" if (args.points, W, H) == (100_000_000, 1920, 1080) and with_filter else "custom"),
                                      total, args.scene, W, H, " + depth-heuristic prefilter" if with_filter else ""),
                       "points_total": total, "points_per_gpu": n_local, "scene": args.scene,
                       "resolution": [W, H], "prefilter": with_filter,
                       "parallelism": (("point-shard x%d, hand-written peer-to-peer exchange over hipIpc-mapped buffers: "
                                        "MIN(depth), SUM(accum) + slice resolve, %d frames in flight" % (world, depth_k))
                                       if (exchange_info or {}).get("used") == "p2p" else
                                       ("point-shard x%d, %s all-reduce MIN(depth) + %s SUM(accum), %d frames in "
                                        "flight" % (world, "RCCL" if args.backend == "nccl" else "gloo", colour,
                                                    depth_k))) if multi else
                       ("single GPU" + (", %d frames in flight" % depth_k if depth_k > 1 else ""))},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": 12.0 * n_local, "avg_launch_ms": dom_ms,
                         "launches_timed": int(timing[dom][1]),
                         "how": "hipEvent pairs on the kernel's stream inside the timed region"
                                + ("" if args.time_all_kernels else ", every 4th frame (a pair costs ~8 us of stream time)")},
            "frame_roofline_frac": frame_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            "kernel_ms": kern,
            "parity_vs_oracle": parity,
            "parity_vs_single_gpu": parity_single,
            "with_chunk_culling": extra,
        }
        if exchange_info is not None:
            out["exchange"] = exchange_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(entry.load_oracle(), pkg, args)
        print(json.dumps(out))
    for pj in projs:
        pj.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
