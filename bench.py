#!/usr/bin/env python3
"""bench.py -- Mpoints/s projected + frames/s of the point-cloud -> framebuffer hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one frame of BASELINE.json config C3: a 100 M-point synthetic cloud projected
to 1920x1080 (clear, min-depth pass, accumulate pass, resolve) plus the depth-heuristic
prefilter, one distinct camera pose of the orbit trajectory per frame.  The cloud is
synthesised on-device before the timed region (inputs resident in HBM).  With N > 1 the
same 100 M points are sharded in contiguous slices over the ranks (config C4: strong
scaling) and the depth / accumulator buffers are MIN / SUM all-reduced over RCCL.

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
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = --points total (BASELINE C4), weak = --points per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-points", type=int, default=10_000_000)
    ap.add_argument("--cpu-frames", type=int, default=0, help="0 = pick so the leg takes about 10-20 s")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle parity gate on frame 0")
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
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = entry.load_package()
    W, H = args.width, args.height
    total = args.points * (world if args.scaling == "weak" else 1)
    lo, hi = pkg.shard_range(total, rank, world)
    proj = pkg.Projector(local_rank)
    proj.generate_synthetic(args.scene, SEEDS["C3"], lo, hi - lo, total)
    proj.set_resolution(W, H)
    with_filter = not args.no_filter

    local = pkg.sharded.HipLocal(proj)
    if world > 1:
        local.bind_stream()  # kernels and RCCL collectives ordered on torch's current stream
    sharded = pkg.ShardedProjector(local)
    poses = [pkg.orbit_projection(k, W, H) for k in range(args.warmup + args.steps)]

    def sync():
        proj.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # parity gate on pose 0 at full size: single-GPU / sharded result vs the oracle run
    # on the host (bounded: only when the cloud is small enough to regenerate on the CPU)
    parity = None
    if rank == 0 and not args.no_parity and world == 1 and total <= 20_000_000:
        orc = entry.load_oracle()
        xyzw, rgba = orc.generate(args.scene, SEEDS["C3"], 0, total, total)
        img, depth = proj.project(poses[0])
        ref = orc.MTProjector(W, H, host_threads()).project(xyzw, rgba, poses[0])
        parity = bool(np.array_equal(depth.view(np.uint32), ref["depth_bits"]) and np.array_equal(img, ref["img"]))
        del xyzw, rgba

    for k in range(args.warmup):
        sharded.render(poses[k], with_filter)
    sync()
    proj.timing_enable(True)
    proj.timing_reset()
    t0 = time.perf_counter()
    for k in range(args.steps):
        sharded.render(poses[args.warmup + k], with_filter)
    sync()
    dt = time.perf_counter() - t0
    timing = proj.timing()
    proj.timing_enable(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

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
            "config": {"workload": "C3: %d-point synthetic %s cloud -> %dx%d, 1x1 splat z-buffer%s"
                                   % (total, args.scene, W, H, " + depth-heuristic prefilter" if with_filter else ""),
                       "points_total": total, "points_per_gpu": n_local, "scene": args.scene,
                       "resolution": [W, H], "prefilter": with_filter,
                       "parallelism": "point-shard x%d, RCCL all-reduce MIN(depth)+SUM(accum)" % world
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": 12.0 * n_local, "avg_launch_ms": dom_ms},
            "frame_roofline_frac": frame_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            "kernel_ms": kern,
            "parity_vs_oracle": parity,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(entry.load_oracle(), pkg, args)
        print(json.dumps(out))
    proj.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
