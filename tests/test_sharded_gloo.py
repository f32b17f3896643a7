"""CPU: the N > 1 frame sequence over torch.distributed/gloo, world_size 2
and 3: point-slice sharding + all-reduce MIN(depth) + all-reduce SUM(accumulators) must be
bit-identical to the single-shard frame (render.cu:81,125-128: min and integer add commute)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scene, n, W, H, with_filter, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as entry
    from oracle_local import OracleLocal
    pkg, orc = entry.load_package(), entry.load_oracle()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        xyzw, rgba = orc.generate(scene, 0xC0FFEE01, 0, n, n)
        lo, hi = pkg.shard_range(n, rank, world)
        local = OracleLocal(orc, xyzw[lo:hi], rgba[lo:hi], W, H)
        sp = pkg.ShardedProjector(local)
        ok = True
        for k in (0, 250):
            P = pkg.orbit_projection(k, W, H)
            sp.render(P, with_filter)
            ref = orc.project(xyzw, rgba, P, W, H)
            ok &= np.array_equal(local.depth.reshape(H, W), ref["depth_bits"])
            ok &= np.array_equal(local.acc.reshape(H, W, 4), ref["acc"])
            if with_filter:
                rf = orc.filter(ref["depth_bits"], ref["img"])
                ok &= np.array_equal(local.filtered["tensor"], rf["tensor"])
                ok &= np.array_equal(local.filtered["img"], rf["img"])
                ok &= np.array_equal(local.filtered["depth"].view(np.uint32), rf["depth"].view(np.uint32))
            else:
                ok &= np.array_equal(local.img, ref["img"])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,scene,with_filter", [(2, "room_shell", False), (2, "uniform_box", True),
                                                      (3, "room_shell", True)])
def test_sharded_frame_equals_single_shard(world, scene, with_filter):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scene, 60_000, 320, 240, with_filter, q))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert res == {r: True for r in range(world)}


def test_shard_range_partitions(pkg):
    for n in (0, 1, 7, 100_000_003):
        for world in (1, 2, 3, 8):
            parts = [pkg.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= 1
