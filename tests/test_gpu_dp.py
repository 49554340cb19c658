"""Data-parallel parity on the GPU (SURVEY.md section 8 row e): two ranks, each running the HIP
training step on half of a 4-image batch with gradients averaged by the flat-bucket reducer
and the centre all-reduced, must follow the SAME trajectory as one process on the whole batch
(the mean of the per-rank losses is the global loss; averaged gradients are the global
gradients).  fp32, 1e-4 relative, three steps.  See tests/dp_worker.py for why the two ranks
share one device over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_two_ranks_follow_the_single_process_trajectory(tmp_path):
    from tests import dp_worker
    from tests.golden_spec import TINY
    from tests.helpers_model import build_pair
    out = str(tmp_path / "dp.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           os.path.join(ROOT, "tests", "dp_worker.py"), out]
    # DINO_DP_STRICT: the reducer raises if a bucket was left for the end-of-backward flush
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DINO_DP_STRICT="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    dp = np.load(out, allow_pickle=False)
    assert float(dp["rank_divergence"][0]) == 0.0

    student, teacher = build_pair(TINY, torch.float32)
    crops = [c.to("cuda") for c in dp_worker.dp_crops()]
    ref = dp_worker.run(student, teacher, crops, dp_worker.NSTEPS, torch.float32,
                        dp_worker.GLOBAL_BATCH)
    for it, r in enumerate(ref):
        loss_dp = float(dp["losses"][:, it].mean())
        assert abs(loss_dp - r["loss"]) < 1e-4 * abs(r["loss"]), (it, loss_dp, r["loss"])
        assert _rel(dp[f"center{it}"], r["center"]) < 1e-4, it
        assert _rel(dp[f"student{it}"], r["student"]) < 1e-4, it
        assert _rel(dp[f"teacher{it}"], r["teacher"]) < 1e-4, it
    # the two ranks really saw different data
    assert abs(dp["losses"][0, 0] - dp["losses"][1, 0]) > 1e-6


def test_rccl_branches_with_one_rank(tmp_path):
    """The code an N-GPU run executes over RCCL -- asynchronous bucket all-reduces with ReduceOp.AVG
    launched from the backward schedule, the stream-level wait before the optimiser, the asynchronous
    centre all-reduce joined at the next step's loss -- on the one GPU this box has: one rank, backend
    "nccl", DINO_DP_FORCE=1 so that the multi-rank paths are taken.  With a single rank every
    collective is the identity, so the run must reproduce the plain single-process trajectory."""
    from tests import dp_worker
    from tests.golden_spec import TINY
    from tests.helpers_model import build_pair
    out = str(tmp_path / "dp1.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           os.path.join(ROOT, "tests", "dp_worker.py"), out]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DINO_DP_STRICT="1", DINO_DP_FORCE="1",
               DP_BACKEND="nccl")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    dp = np.load(out, allow_pickle=False)
    student, teacher = build_pair(TINY, torch.float32)
    crops = [c.to("cuda") for c in dp_worker.dp_crops()]
    ref = dp_worker.run(student, teacher, crops, dp_worker.NSTEPS, torch.float32, dp_worker.GLOBAL_BATCH)
    for it, r_ in enumerate(ref):
        assert abs(float(dp["losses"][0, it]) - r_["loss"]) < 1e-5 * abs(r_["loss"]), it
        assert _rel(dp[f"center{it}"], r_["center"]) < 1e-5, it
        assert _rel(dp[f"student{it}"], r_["student"]) < 1e-4, it
        assert _rel(dp[f"teacher{it}"], r_["teacher"]) < 1e-4, it
