"""-m gpu: bench.py's real N > 1 code path with TWO ranks on one GPU (torch.distributed.run,
gloo backend moving CUDA tensors): point-slice sharding, MIN / SUM exchange, reduce-scatter
colour form with its self-check, two frames in flight -- and the frame of pose 0 compared with
the oracle inside bench.py ("parity_vs_oracle").  RCCL itself is covered by test_gpu_rccl.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("extra", [[], ["--scaling", "weak", "--colour", "allreduce", "--pipeline", "1"]],
                         ids=["strong-rs-pipelined", "weak-allreduce"])
def test_two_rank_bench_matches_oracle(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend",
           "gloo", "--points", "3000000", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["parity_vs_oracle"] is True
    assert out["parity_vs_single_gpu"] is True   # rank 0 renders the whole cloud alone and compares
    # default = BASELINE C4's regime: the SAME cloud sharded over the ranks, the weak figure as an extra key
    assert out["scaling"] == ("weak" if "weak" in extra else "strong")
    assert out["config"]["points_total"] == (6_000_000 if "weak" in extra else 3_000_000)
    if "weak" not in extra:
        assert out["weak_scaling"]["points_total"] == 6_000_000 and out["weak_scaling"]["value"] > 0
    assert "multi-GPU box" in out["multi_gpu_note"]
    assert out["roofline"]["bound"] == "hbm" and out["value"] > 0
    # --exchange auto: the collectives are timed first, then the hand-written peer-to-peer exchange,
    # which must have matched them before and after its timed frames on both ranks
    assert out["exchange"]["p2p_clean_on_all_ranks"] is True, out["exchange"]
    # ... and the owner-computes form (every tile produced once, the frame's owner rotating over the ranks)
    assert out["exchange"]["owned_clean_on_all_ranks"] is True, out["exchange"]
    assert out["exchange"]["used"] in ("owned", "p2p", "collective")


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment: bench.py starts the two rank
    processes itself (children of a process that never touches the GPU), prints rank 0's one JSON line and exits 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--points", "3000000",
           "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-weak", "--exchange", "collective"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["parity_vs_oracle"] is True


def test_self_launch_reports_a_failing_rank():
    """... and a rank that fails makes the whole command fail (an unknown scene name is refused by every rank)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--points", "1000",
           "--set", "no_such_option=1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-weak"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode != 0
