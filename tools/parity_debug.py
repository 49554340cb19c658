"""Per-tensor gradient parity of the HIP step against the CPU oracle on one golden_spec case
(step 0, fp32): prints the worst tensors.  Usage: python tools/parity_debug.py [full|multi|tiny]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import golden_spec as G  # noqa: E402
from tests.helpers_model import build_pair, oracle_trainer  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "multi"
spec, crops = {"full": (G.FULL, G.full_crops), "multi": (G.MULTI, G.multi_crops),
               "tiny": (G.TINY, G.tiny_crops)}[which]
crops = crops()
torch.set_num_threads(16)
if os.environ.get("PD_NO_ORACLE") == "1":
    g = np.load(os.path.join("tests", "golden", "vits16_b8_steps.npz"))
    ref = None
else:
    ora = oracle_trainer(spec, 3)
    trace = {}
    out = ora.step(crops, 0, trace=trace)
    ref = trace["grads_raw"]

from dino_amd.components.loss import DINOLoss  # noqa: E402
ONLY = os.environ.get("PD_ONLY_STEP") == "1"
student, teacher = build_pair(spec, torch.float32)
dc = [c.cuda() for c in crops]
loss_fn = DINOLoss(spec["head"]["out_dim"], len(crops), 0.04, 0.04, 0, 3).cuda()
if ONLY:
    names = [n for n, _ in student.named_parameters()]
    refn = g["gnorms0"] if ref is None else np.array([ref[n].double().norm().item() for n in names])
    from dino_amd.utils import get_params_groups
    from dino_amd.optim import FusedAdamW
    from dino_amd.utils.train_utils import train_step
    from tests.helpers_model import schedules
    opt = FusedAdamW(get_params_groups(student), student, teacher)
    lr, wd, mom, _ = schedules(spec["batch"], 3)
    loss, work = train_step(student, teacher, loss_fn, opt, dc, 0, lr, wd, mom, 3.0, spec["freeze_iter"])
    kn2 = np.array(work.norms.tolist())
    tn = np.array([0.0 if p.grad is None else p.grad.double().norm().item() for _, p in student.named_parameters()])
    r = np.abs(kn2 / refn - 1)
    print("ONLY train_step flow: loss", loss.item(), "norms vs oracle max rel", r.max())
    for i in np.argsort(-r)[:6]:
        print("   ", names[i], kn2[i], refn[i], r[i], "torch norm of p.grad now:", tn[i])
    sys.exit(0)
with torch.no_grad():
    t_out = teacher(dc[:2])
tr = {}
s_out = student(dc, trace=tr)
s_out.retain_grad()
loss = loss_fn(s_out, t_out, 0)
loss.backward()


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item(), ((a - b).norm() / (b.norm() + 1e-30)).item()


print("loss", loss.item(), out["loss"].item())
print("student_out", rel(s_out, trace["student_out"]))
print("teacher_out", rel(t_out, trace["teacher_out"]))
print("dstudent", rel(s_out.grad, trace["student_out"].grad))
for k in ("features", "bottleneck"):
    if k in tr and k in trace:
        print(k, rel(tr[k], trace[k]))
rows = []
for n, p in student.named_parameters():
    mx, l2 = rel(p.grad, ref[n])
    nr = abs(p.grad.double().norm().item() / ref[n].double().norm().item() - 1)
    rows.append((nr, mx, l2, n))
rows.sort(reverse=True)
for r in rows[:25]:
    print("norm-rel %.2e  max-rel %.2e  l2-rel %.2e  %s" % r)

# ---- the kernels' per-tensor norms against torch's on the same grads, then the train_step flow
from dino_amd.utils import grad_stats_async, get_params_groups  # noqa: E402
from dino_amd.optim import FusedAdamW  # noqa: E402
from dino_amd.utils.train_utils import train_step  # noqa: E402
from tests.helpers_model import schedules  # noqa: E402
names = [n for n, _ in student.named_parameters()]
refn = np.array([ref[n].double().norm().item() for n in names])
w = grad_stats_async(student, 3.0, "student", materialize=False)
kn = np.array(w.norms.tolist())
tn = np.array([p.grad.double().norm().item() for _, p in student.named_parameters()])
print("kernel norms vs torch norms of the same grads: max rel", np.abs(kn / tn - 1).max())
print("kernel norms vs oracle: max rel", np.abs(kn / refn - 1).max())
del student, teacher
student, teacher = build_pair(spec, torch.float32)
loss_fn = DINOLoss(spec["head"]["out_dim"], len(crops), 0.04, 0.04, 0, 3).cuda()
opt = FusedAdamW(get_params_groups(student), student, teacher)
lr, wd, mom, _ = schedules(spec["batch"], 3)
loss, work = train_step(student, teacher, loss_fn, opt, dc, 0, lr, wd, mom, 3.0, spec["freeze_iter"])
kn2 = np.array(work.norms.tolist())
r = np.abs(kn2 / refn - 1)
print("train_step flow: loss", loss.item(), "norms vs oracle max rel", r.max())
for i in np.argsort(-r)[:6]:
    print("   ", names[i], kn2[i], refn[i], r[i])
