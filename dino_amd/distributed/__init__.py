"""Process-group helpers with the reference's names (dino/distributed/__init__.py:15-77,
251-303) over RCCL (torch.distributed backend "nccl" on ROCm), one process per GPU, plus the
flat-bucket data-parallel gradient reducer (see .reducer)."""
import datetime
import os
import socket

import torch
import torch.distributed as dist

_LOCAL_RANK = -1
_LOCAL_WORLD_SIZE = -1

_ENV_VARS = ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")


def is_enabled() -> bool:
    return dist.is_available() and dist.is_initialized()


def is_enabled_and_multiple_gpus() -> bool:
    """Distributed AND more than one rank (the reference tests torch.cuda.device_count() > 1;
    with one process per GPU the rank count is the meaningful quantity, and it also lets the
    gloo CPU tests exercise the multi-rank path)."""
    if not is_enabled():
        return False
    # DINO_DP_FORCE=1: take the multi-rank code paths (bucketed all-reduce, centre all-reduce) even
    # with one rank -- lets a 1-GPU box exercise the RCCL branches (tests/test_gpu_dp.py)
    return dist.get_world_size() > 1 or os.environ.get("DINO_DP_FORCE", "0") == "1"


def get_global_size() -> int:
    return dist.get_world_size() if is_enabled() else 1


def get_global_rank() -> int:
    return dist.get_rank() if is_enabled() else 0


def get_local_rank() -> int:
    if not is_enabled():
        return 0
    assert 0 <= _LOCAL_RANK < _LOCAL_WORLD_SIZE
    return _LOCAL_RANK


def get_local_size() -> int:
    if not is_enabled():
        return 1
    assert 0 <= _LOCAL_RANK < _LOCAL_WORLD_SIZE
    return _LOCAL_WORLD_SIZE


def is_main_process() -> bool:
    return get_global_rank() == 0


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def enable(*, set_cuda_current_device: bool = True, overwrite: bool = False,
           allow_nccl_timeout: bool = False, backend: str = None):
    """Join the env:// rendez-vous that torchrun prepared (or a 1-rank local one) and create the
    process group: RCCL ("nccl") when a GPU is present, gloo otherwise (CPU tests)."""
    global _LOCAL_RANK, _LOCAL_WORLD_SIZE
    if _LOCAL_RANK >= 0 or _LOCAL_WORLD_SIZE >= 0:
        raise RuntimeError("Distributed mode has already been enabled")
    present = [v for v in _ENV_VARS if v in os.environ]
    if present and len(present) != len(_ENV_VARS):
        raise RuntimeError(f"Partially set environment: {', '.join(present)}")
    if not present:
        env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0",
               "WORLD_SIZE": "1", "LOCAL_RANK": "0", "LOCAL_WORLD_SIZE": "1"}
        os.environ.update(env)
    local_rank = int(os.environ["LOCAL_RANK"])
    use_gpu = torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if use_gpu else "gloo"
    kw = {}
    if backend == "nccl":
        if set_cuda_current_device:
            torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device(f"cuda:{local_rank}")
        if allow_nccl_timeout:
            os.environ.setdefault("NCCL_ASYNC_ERROR_HANDLING", "1")
    dist.init_process_group(backend=backend, timeout=datetime.timedelta(days=14), **kw)
    dist.barrier()
    _LOCAL_RANK = local_rank
    _LOCAL_WORLD_SIZE = int(os.environ["LOCAL_WORLD_SIZE"])
    # (The GEMMs' work queue -- ops.gemm_work_queue / DINO_NT_QUEUE=1 -- is NOT switched on here: it makes a GEMM that
    # starts while RCCL's kernels hold CUs 1.03 x instead of 1.5 x slower, but costs 0.7 ms per step by itself; on the
    # one-GPU stand-in (bench.py --occupy) it pays from ~8 ms of held CUs per step on, more than the ~2 ms that
    # 176 MB of gradients should take over xGMI.  DESIGN.md section 5.)

def disable():
    global _LOCAL_RANK, _LOCAL_WORLD_SIZE
    if is_enabled():
        dist.destroy_process_group()
    _LOCAL_RANK = _LOCAL_WORLD_SIZE = -1


def all_reduce(t: torch.Tensor, op=dist.ReduceOp.SUM, group=None):
    """Blocking all-reduce.  RCCL orders itself after the producing stream; gloo given a device
    tensor stages it through pinned host copies on a side stream, and that wait on a deep
    backlog of compute was seen to dead-lock when ranks share a GPU (the 1-GPU rehearsal), so
    that path drains the device first."""
    if t.is_cuda and dist.get_backend(group) != "nccl":
        torch.cuda.synchronize(t.device)
    dist.all_reduce(t, op=op, group=group)
    return t


def gather_tensor(t: torch.Tensor, dst_rank: int = 0):
    gather_list = []
    if dist.get_rank() == dst_rank:
        gather_list = [torch.empty_like(t) for _ in range(get_global_size())]
    dist.gather(t, gather_list, dst=dst_rank)
    return gather_list


from .reducer import FlatGradReducer, DataParallel  # noqa: E402,F401
