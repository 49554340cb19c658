"""One rank of the 2-rank data-parallel parity run (started by tests/test_gpu_dp.py through
torch.distributed.run).  Both ranks share cuda:0 -- RCCL refuses two ranks on one device, so the
process group is gloo here; everything above the collective backend (DataParallel's flat-bucket
reducer hooked into the backward schedule, the centre all-reduce in DINOLoss, FusedAdamW on the
averaged gradients) is the code an N-GPU RCCL run executes.

Each rank trains on ITS half of a 4-image batch; rank 0 writes per-step losses of both ranks,
the centre and the student/teacher parameters to argv[1]."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NSTEPS = 3
GLOBAL_BATCH = 4


def dp_crops():
    from oracle import detfill
    from tests.golden_spec import TINY
    return detfill.det_crops(GLOBAL_BATCH, TINY["sizes"], tag="dp:img")


def run(student, teacher, crops, nsteps, dtype, world_batch):
    import dino_amd.distributed as D
    from dino_amd.components.loss import DINOLoss
    from dino_amd.optim import FusedAdamW
    from dino_amd.utils import get_params_groups
    from dino_amd.utils.train_utils import train_step
    from tests.golden_spec import TINY
    from tests.helpers_model import schedules
    lr, wd, mom, _ = schedules(world_batch, nsteps)
    loss_fn = DINOLoss(TINY["head"]["out_dim"], len(crops), 0.04, 0.04, 0, nsteps).to("cuda")
    opt = FusedAdamW(get_params_groups(student), student, teacher)
    out = []
    for it in range(nsteps):
        loss, _ = train_step(student, teacher, loss_fn, opt, crops, it, lr, wd, mom, 3.0,
                             TINY["freeze_iter"], sync_stats=True)
        inner = student.module if hasattr(student, "module") else student
        # Only the main process looks at the loss module between steps, as train_one_epoch does when it
        # checkpoints (dino/utils/train_utils.py:176-188: dino_loss.state_dict() on rank 0 only): joining the
        # centre update must not involve a collective the other ranks never enter.
        main = D.is_main_process()
        center = loss_fn.state_dict()["center"].detach().cpu().numpy().copy() if main else None
        out.append(dict(loss=loss.item(), center=center,
                        student=torch.cat([p.detach().reshape(-1) for p in inner.parameters()]).cpu().numpy(),
                        teacher=torch.cat([p.detach().reshape(-1) for p in teacher.parameters()]).cpu().numpy()))
    return out


def main():
    import dino_amd
    import dino_amd.distributed as D
    from tests.golden_spec import TINY
    from tests.helpers_model import build_pair
    dtype = torch.float32
    backend = os.environ.get("DP_BACKEND", "gloo")     # "nccl": the RCCL branches, one rank per device
    if backend == "nccl":
        D.enable(backend="nccl")
    else:
        D.enable(backend="gloo", set_cuda_current_device=False)
    rank, world = D.get_global_rank(), D.get_global_size()
    if backend != "nccl":
        torch.cuda.set_device(0)
    dino_amd.set_default_dtype(dtype)
    student, teacher = build_pair(TINY, dtype)
    if rank != 0:       # DataParallel must bring every rank to rank 0's weights
        with torch.no_grad():
            for p in student.parameters():
                p.add_(0.01)
    student = D.DataParallel(student, bucket_cap_mb=1)   # > 1 bucket even for the tiny model
    assert len(student.reducer.buckets) >= 2
    teacher.load_state_dict(student.module.state_dict())
    per = GLOBAL_BATCH // world
    crops = [c[rank * per:(rank + 1) * per].contiguous().to("cuda") for c in dp_crops()]
    res = run(student, teacher, crops, NSTEPS, dtype, GLOBAL_BATCH)
    cdev = "cuda" if backend == "nccl" else "cpu"      # (RCCL has no CPU tensors)
    losses = torch.tensor([r["loss"] for r in res], dtype=torch.float64, device=cdev)
    gathered = [torch.zeros_like(losses) for _ in range(world)]
    torch.distributed.all_gather(gathered, losses)
    # every rank must hold the same parameters after every step
    last = torch.from_numpy(res[-1]["student"]).double().to(cdev)
    ref = last.clone()
    torch.distributed.broadcast(ref, src=0)
    same = torch.tensor([float((ref - last).abs().max())], dtype=torch.float64, device=cdev)
    torch.distributed.all_reduce(same, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        blob = {"losses": torch.stack(gathered).cpu().numpy(), "rank_divergence": same.cpu().numpy()}
        for it, r in enumerate(res):
            blob[f"center{it}"] = r["center"]
            blob[f"student{it}"] = r["student"]
            blob[f"teacher{it}"] = r["teacher"]
        np.savez(sys.argv[1], **blob)
    torch.cuda.synchronize()
    torch.distributed.barrier()      # nobody tears the group down while a peer is still in a collective
    D.disable()


if __name__ == "__main__":
    main()
