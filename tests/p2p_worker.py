"""Worker of test_gpu_p2p.py (one rank of a torch.distributed.run job, all ranks on GPU 0):
renders a point-sharded cloud with the library's peer-to-peer exchange (hipIpc mappings of the
other ranks' frame buffers, flag barriers) and checks every rank's frames against the oracle."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    W, H, n, frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    scene, mode = sys.argv[5], int(sys.argv[6])
    stall_rank = int(sys.argv[7]) if len(sys.argv) > 7 else -1  # this rank sleeps past the barrier timeout once
    form = sys.argv[8] if len(sys.argv) > 8 else "p2p"          # "owned": rtr_p2p_render_owned, the frame owner rotates
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg, orc = entry.load_package(), entry.load_oracle()
    proj = pkg.Projector(0)
    lo, hi = pkg.shard_range(n, rank, world)
    proj.set_option("mode", mode)   # 0: the atomic form, no bins: every tile counts as occupied
    proj.generate_synthetic(scene, 11, lo, hi - lo, n)
    proj.set_resolution(W, H)
    local = pkg.sharded.HipLocal(proj)
    local.bind_stream()
    if stall_rank >= 0:
        proj.set_option("p2p_timeout_ms", 150)
    sp = pkg.ShardedProjector(local, colour="reduce_scatter", exchange="p2p", check_every=1 if stall_rank >= 0 else 16)
    xyzw, rgba = orc.generate(scene, 11, 0, n, n)
    ok, notes = True, []
    if form == "owned":
        # owner-computes form: every tile is produced once, by a rank that has points in it (reading the other
        # occupying ranks' entries out of their tile stores); only the frame's owner ends with the whole frame
        local.p2p_setup(rank, world, None)
        for k in range(frames):
            P = pkg.orbit_projection(7 * k, W, H)
            filt = (k % 2 == 1) and W % 16 == 0
            owner = (k + 1) % world
            proj.p2p_render_owned(P, filt, owner)
            if rank == owner:
                ref = orc.project(xyzw, rgba, P, W, H)
                rd, ri, rt = ref["depth_bits"], ref["img"], None
                if filt:
                    rf = orc.filter(rd, ri)
                    rd, ri, rt = rf["depth"].view(np.uint32), rf["img"], rf["tensor"]
                same = np.array_equal(proj.download(pkg._lib.BUF_DEPTH), rd) and np.array_equal(proj.download(pkg._lib.BUF_IMAGE), ri)
                if rt is not None:
                    same = same and np.array_equal(proj.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rt)
                if not same:
                    ok = False
                    notes.append("frame %d differs on its owner, rank %d" % (k, rank))
            else:
                proj.synchronize()
        out = {"rank": rank, "ok": ok, "exchange": "owned", "p2p_note": None, "timeouts": proj.p2p_timeouts(),
               "notes": notes, "suspect": None, "errors": proj.frame_stats()["errors"]}
        gathered = [None] * world
        dist.all_gather_object(gathered, out)
        if rank == 0:
            print(json.dumps(gathered), flush=True)
        dist.barrier()
        proj.close()
        dist.destroy_process_group()
        return
    for k in range(frames):
        P = pkg.orbit_projection(7 * k, W, H)
        filt = (k % 2 == 1) and W % 16 == 0
        if k == 3 and rank == stall_rank:
            import time
            proj.synchronize()
            time.sleep(1.0)  # the other ranks' barriers give up after 150 ms
        sp.render(P, filt)
        ref = orc.project(xyzw, rgba, P, W, H)
        rd, ri = ref["depth_bits"], ref["img"]
        if filt:
            rf = orc.filter(rd, ri)
            rd, ri = rf["depth"].view(np.uint32), rf["img"]
        same = np.array_equal(proj.download(pkg._lib.BUF_DEPTH), rd) and np.array_equal(proj.download(pkg._lib.BUF_IMAGE), ri)
        if not same:
            ok = False
            notes.append("frame %d differs on rank %d" % (k, rank))
    out = {"rank": rank, "ok": ok, "exchange": sp.exchange, "p2p_note": sp.p2p_note, "timeouts": proj.p2p_timeouts(),
           "notes": notes, "suspect": sp.p2p_suspect_frames}
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        print(json.dumps(gathered), flush=True)
    dist.barrier()
    proj.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
