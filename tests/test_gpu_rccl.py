"""-m gpu: the exchange steps through REAL RCCL on one GPU (a 1-rank nccl group): checks
that RCCL accepts the zero-copy int32 views of the library's buffers, MIN / SUM ops, the
reduce-scatter + in-place all-gather colour form, and that the frame stays bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("colour", ["allreduce", "reduce_scatter"])
def test_exchange_path_through_rccl(pkg, orc, nccl_group, colour):
    import torch
    n, W, H = 150_000, 1920, 1080
    xyzw, rgba = orc.generate("room_shell", 33, 0, n, n)
    p = pkg.Projector(0)
    try:
        p.upload_points(xyzw, rgba)
        p.set_resolution(W, H)
        loc = pkg.sharded.HipLocal(p)
        loc.bind_stream()
        sp = pkg.ShardedProjector(loc, colour=colour, force_exchange=True)
        for k in (5, 400):
            P = pkg.orbit_projection(k, W, H)
            sp.render(P, with_filter=True)
            torch.cuda.synchronize()
            ref = orc.project(xyzw, rgba, P, W, H)
            rf = orc.filter(ref["depth_bits"], ref["img"])
            assert np.array_equal(p.download(pkg._lib.BUF_IMAGE), rf["img"])
            assert np.array_equal(p.download(pkg._lib.BUF_DEPTH), rf["depth"].view(np.uint32))
            assert np.array_equal(p.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])
    finally:
        p.close()
