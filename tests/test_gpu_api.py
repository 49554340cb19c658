"""API-level behaviour a user of the reference relies on, on the GPU (SURVEY.md section 8f row 4
and the edge cases of 8b): stand-alone backbone inference, checkpoint round trip through
state_dict()/torch.save (reference train_utils.py:174-188, utils/utils.py:139-181),
train_one_epoch over a loader, tensor / list inputs, batch of one, stand-alone update_center."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detfill  # noqa: E402
from oracle import dino_oracle as O  # noqa: E402
from tests.golden_spec import TINY  # noqa: E402
from tests.helpers_model import build_pair, oracle_cfgs, oracle_trainer, schedules  # noqa: E402

DEV = "cuda"


def _rel(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _crops(batch, tag):
    return detfill.det_crops(batch, TINY["sizes"], tag=tag)


def _trainer(student, teacher, nsteps, ncrops):
    from dino_amd.components.loss import DINOLoss
    from dino_amd.optim import FusedAdamW
    from dino_amd.utils import get_params_groups
    loss_fn = DINOLoss(TINY["head"]["out_dim"], ncrops, 0.04, 0.04, 0, nsteps).to(DEV)
    opt = FusedAdamW(get_params_groups(student), student, teacher)
    return loss_fn, opt


def test_backbone_inference_matches_oracle():
    """Stand-alone VisionTransformer in eval mode (the Tuner's feature extraction,
    dino/eval/features.py:82-178): CLS features and get_intermediate_layers."""
    from functools import partial
    import torch.nn as nn
    from dino_amd.models import vision_transformer as vits
    v = TINY["vit"]
    model = vits.VisionTransformer(img_size=v["img_size"], patch_size=v["patch_size"],
                                   embed_dim=v["embed_dim"], depth=v["depth"], num_heads=v["num_heads"],
                                   mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    model.compute_dtype = torch.float32
    sd = detfill.fill_state_dict_(dict(model.named_parameters()), tag="infer")
    ref_sd = {k: t.detach().clone() for k, t in sd.items()}
    model = model.to(DEV).eval()
    vit_cfg, _ = oracle_cfgs(TINY)
    for size in (56, 24):                      # native grid and the interpolated one
        x = detfill.det_normal((3, 3, size, size), f"infer:img{size}")
        with torch.no_grad():
            got = model(x.to(DEV))
        trace = {}
        want = O.vit_forward(x, ref_sd, vit_cfg, pre="", trace=trace)
        assert got.shape == want.shape == (3, v["embed_dim"])
        assert _rel(got, want) < 1e-4, size
        outs = model.get_intermediate_layers(x.to(DEV), n=2)
        assert len(outs) == 2
        for j, o in enumerate(outs):
            t = trace[f"block{v['depth'] - 2 + j}"]
            ref = O.layer_norm(t, ref_sd["norm.weight"], ref_sd["norm.bias"], vit_cfg.ln_eps)
            assert _rel(o, ref) < 1e-4, (size, j)
    # the module holds no autograd state after no-grad passes and can still train
    assert not any(p.grad is not None for p in model.parameters())


def test_checkpoint_round_trip_resumes_the_trajectory(tmp_path):
    """Save student / teacher / optimizer / loss state after 2 steps (the reference's checkpoint
    dict), load it into freshly built objects, and the third step must equal the uninterrupted
    run's third step."""
    from dino_amd.utils.train_utils import train_step
    nsteps = 3
    crops = [c.to(DEV) for c in _crops(TINY["batch"], "ckpt:img")]
    lr, wd, mom, _ = schedules(TINY["batch"], nsteps)

    def fresh():
        s, t = build_pair(TINY, torch.float32)
        lf, op = _trainer(s, t, nsteps, len(crops))
        return s, t, lf, op

    s, t, lf, op = fresh()
    for it in range(2):
        train_step(s, t, lf, op, crops, it, lr, wd, mom, 3.0, TINY["freeze_iter"])
    path = str(tmp_path / "ckpt.pt")
    torch.save({"iteration": 2, "student": s.state_dict(), "teacher": t.state_dict(),
                "optimizer": op.state_dict(), "dino_loss": lf.state_dict()}, path)
    loss_a, _ = train_step(s, t, lf, op, crops, 2, lr, wd, mom, 3.0, TINY["freeze_iter"])
    ref_student = [p.detach().clone() for p in s.parameters()]
    ref_teacher = [p.detach().clone() for p in t.parameters()]

    s2, t2, lf2, op2 = fresh()
    from dino_amd.utils import resume_from_checkpoint
    ck = torch.load(path, map_location=DEV, weights_only=True)
    it0 = resume_from_checkpoint(path, verbose=False, student=s2, teacher=t2, optimizer=op2, dino_loss=lf2)
    assert it0 == 2
    loss_b, _ = train_step(s2, t2, lf2, op2, crops, it0, lr, wd, mom, 3.0, TINY["freeze_iter"])
    assert abs(loss_a.item() - loss_b.item()) < 1e-5 * abs(loss_a.item())
    for a, b in zip(ref_student, s2.parameters()):
        assert _rel(b, a) < 1e-5
    for a, b in zip(ref_teacher, t2.parameters()):
        assert _rel(b, a) < 1e-5
    # names and order are the reference's (158-entry layout for ViT-S; here the tiny model)
    assert list(ck["student"].keys()) == [n for n, _ in s.named_parameters()]


def test_train_one_epoch_over_a_loader(tmp_path):
    """train_one_epoch with the reference's argument list over a 3-batch loader: mean loss equals
    the oracle's, checkpoints appear on the save_every grid, schedules are applied."""
    from dino_amd.utils.train_utils import train_one_epoch
    niter = 3
    batches = [_crops(TINY["batch"], f"epoch:img{i}") for i in range(niter)]
    loader = [(b, torch.zeros(TINY["batch"], dtype=torch.long)) for b in batches]
    s, t = build_pair(TINY, torch.float32)
    lf, op = _trainer(s, t, niter, len(batches[0]))
    lr, wd, mom, _ = schedules(TINY["batch"], niter)
    stats = train_one_epoch(s, t, t, lf, loader, op, lr, wd, mom, 0, 1, None, 3.0, TINY["freeze_iter"],
                            2, str(tmp_path), 0)
    tr = oracle_trainer(TINY, niter)
    ref = [float(tr.step(batches[i], i)["loss"]) for i in range(niter)]
    assert abs(stats["loss"] - sum(ref) / niter) < 1e-4 * abs(sum(ref) / niter)
    assert math.isclose(stats["lr"], lr[niter - 1]) and math.isclose(stats["wd"], wd[niter - 1])
    assert sorted(os.listdir(tmp_path)) == ["iter_000000.pt", "iter_000002.pt"]
    ck = torch.load(os.path.join(tmp_path, "iter_000002.pt"), map_location="cpu", weights_only=True)
    assert set(ck) == {"iteration", "student", "teacher", "optimizer", "dino_loss"}


def test_tensor_input_batch_of_one_and_uneven_crop_lists():
    """MultiCropWrapper accepts a bare tensor (models/__init__.py:26-27), a batch of one, and any
    run structure of resolutions (e.g. global, local, global)."""
    s, _ = build_pair(TINY, torch.float32)
    sd = {k: p.detach().cpu().clone() for k, p in s.named_parameters()}
    vit_cfg, head_cfg = oracle_cfgs(TINY)
    x = detfill.det_normal((1, 3, 56, 56), "edge:one")
    with torch.no_grad():
        got = s(x.to(DEV))
    want = O.multicrop_forward([x], sd, vit_cfg, head_cfg)
    assert got.shape == (1, TINY["head"]["out_dim"]) and _rel(got, want) < 1e-4
    crops = [detfill.det_normal((2, 3, sz, sz), f"edge:{i}") for i, sz in enumerate((56, 24, 24, 56, 24))]
    with torch.no_grad():
        got = s([c.to(DEV) for c in crops])
    want = O.multicrop_forward(crops, sd, vit_cfg, head_cfg)
    assert got.shape == (10, TINY["head"]["out_dim"]) and _rel(got, want) < 1e-4
    with pytest.raises(Exception):
        s(torch.zeros(1, 3, 56, 56))            # CPU tensor: no fallback


def test_update_center_stand_alone():
    from dino_amd.components.loss import DINOLoss
    K = 512
    lf = DINOLoss(K, 4, 0.04, 0.04, 0, 10).to(DEV)
    t_out = detfill.det_normal((6, K), "center:t")
    center = torch.zeros(1, K)
    for _ in range(2):
        lf.update_center(t_out.to(DEV))
        center = O.update_center(center, t_out, 0.9)
    assert _rel(lf.center.reshape(1, K), center) < 1e-5


@pytest.mark.parametrize("norm_last_layer,clip", [(True, 3.0), (False, 0.0)])
def test_head_options_match_oracle(norm_last_layer, clip):
    """norm_last_layer=True (weight_g frozen at its value and absent from the optimiser groups,
    vision_transformer.py:706-708) and clip_grad=0 (no clipping, default.yaml:29): three fp32 steps
    against the oracle."""
    from dino_amd.utils.train_utils import train_step
    nsteps = 3
    crops = _crops(TINY["batch"], "opt:img")
    s, t = build_pair(TINY, torch.float32, norm_last_layer=norm_last_layer)
    lf, op = _trainer(s, t, nsteps, len(crops))
    lr, wd, mom, tt = schedules(TINY["batch"], nsteps)
    vit_cfg, head_cfg = oracle_cfgs(TINY)
    sd = detfill.fill_state_dict_(O.new_state_dict(vit_cfg, head_cfg), tag=TINY["tag"])
    if norm_last_layer:
        # (the deterministic fill gives weight_g values near 1 in both; what is tested is that it
        # stays frozen and out of the optimiser while weight_v trains)
        assert not dict(s.named_parameters())["head.last_layer.weight_g"].requires_grad
    hp = O.StepHP(ncrops=len(crops), clip_grad=clip, freeze_last_layer_iter=TINY["freeze_iter"],
                  norm_last_layer=norm_last_layer)
    tr = O.OracleTrainer(vit_cfg, head_cfg, hp, sd, lr, wd, mom, tt)
    dcrops = [c.to(DEV) for c in crops]
    for it in range(nsteps):
        loss, _ = train_step(s, t, lf, op, dcrops, it, lr, wd, mom, clip, TINY["freeze_iter"])
        ref = float(tr.step(crops, it)["loss"])
        assert abs(loss.item() - ref) < 1e-4 * abs(ref), (it, loss.item(), ref)
    for k, p in s.named_parameters():
        assert _rel(p, tr.student[k]) < 1e-4, k
    for k, p in t.named_parameters():
        assert _rel(p, tr.teacher[k]) < 1e-4, k


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 1e-2)])
def test_loss_module_with_temperature_warmup(dtype, tol):
    """DINOLoss as a module over several iterations of a warming teacher temperature
    (loss.py:28-36): forward value, gradient w.r.t. the student logits and the centre after each
    call, against the oracle."""
    from dino_amd.components.loss import DINOLoss
    K, B, ncrops, iters = 1024, 3, 4, 5
    lf = DINOLoss(K, ncrops, 0.02, 0.07, 0.6, iters).to(DEV)
    sched = O.teacher_temp_schedule(0.02, 0.07, 0.6, iters)
    assert np.allclose(lf.teacher_temp_schedule, sched)
    center = torch.zeros(1, K)
    for it in range(iters):
        s = detfill.det_normal((ncrops * B, K), f"lw:s{it}")
        t = detfill.det_normal((2 * B, K), f"lw:t{it}")
        s_ref = s.to(dtype).float().clone().requires_grad_(True)
        t_ref = t.to(dtype).float()
        ref = O.dino_loss(s_ref, t_ref, center, ncrops, float(sched[it]))
        ref.backward()
        center = O.update_center(center, t_ref, 0.9)
        sd = s.detach().to(DEV).to(dtype).requires_grad_(True)
        loss = lf(sd, t.to(DEV).to(dtype), it)
        loss.backward()
        assert abs(loss.item() - ref.item()) < tol * abs(ref.item()), (it, loss.item(), ref.item())
        assert _rel(sd.grad, s_ref.grad) < max(tol, 2e-4), it
        assert _rel(lf.center, center) < 1e-4, it


def test_unsynced_steps_use_each_steps_own_hyperparameters():
    """the optimiser's per-tensor table (lr, wd, bias corrections, active flag)
    goes to the device through pinned memory while the host runs ahead of a GPU-bound step.  20 steps
    with NO host synchronisation in between: the table the kernel of step i read (snapshotted on the
    stream right behind it) must be step i's own, and the weights must match a run that synchronises
    after every step."""
    from dino_amd.components.loss import DINOLoss
    from dino_amd.optim import FusedAdamW
    from dino_amd.utils import get_params_groups
    from dino_amd.utils.train_utils import train_step
    from tests.golden_spec import FULL
    nsteps, B = 20, 96
    spec = dict(FULL, batch=B, sizes=[224, 224, 96, 96], freeze_iter=6)
    crops = [torch.randn(B, 3, s, s, device=DEV, generator=torch.Generator(DEV).manual_seed(i))
             for i, s in enumerate(spec["sizes"])]
    lr = np.linspace(0.0, 2e-3, nsteps)            # changes a lot from step to step
    wd = np.linspace(0.04, 0.4, nsteps)
    mom = np.full(nsteps, 0.99)
    finals = {}
    for mode in ("free", "synced", "synced2"):
        student, teacher = build_pair(spec, torch.bfloat16)
        loss_fn = DINOLoss(spec["head"]["out_dim"], len(crops), 0.04, 0.04, 0, nsteps).to(DEV)
        opt = FusedAdamW(get_params_groups(student), student, teacher)
        log = None
        for it in range(nsteps):
            train_step(student, teacher, loss_fn, opt, crops, it, lr, wd, mom, 3.0, spec["freeze_iter"])
            if log is None:
                log = torch.empty((nsteps,) + tuple(opt._hp_dev.shape), device=DEV)
            log[it].copy_(opt._hp_dev)             # same stream: what step it's kernel just read
            if mode != "free":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        log = log.cpu().numpy()
        names = student.store().layout.names
        reg = {id(p) for p in opt.param_groups[0]["params"]}
        for it in range(nsteps):
            for r, k in enumerate(names):
                frozen = "last_layer" in k and it < spec["freeze_iter"]
                t = it + 1 - (spec["freeze_iter"] if "last_layer" in k else 0)
                row = log[it, r]
                if frozen:
                    assert row[4] == 0.0 and row[0] == 0.0, (mode, it, k, row)
                    continue
                want = [lr[it], wd[it] if id(student.store().params[k]) in reg else 0.0,
                        1 - 0.9 ** t, 1 - 0.999 ** t, 1.0]
                np.testing.assert_allclose(row[:5], np.asarray(want, np.float32), rtol=1e-6, atol=0,
                                           err_msg=f"{mode} step {it} {k}")
        finals[mode] = [p.detach().float().clone() for p in teacher.parameters()]
        del student, teacher, opt, loss_fn
    # Adam turns the last-bit differences of the weight-gradient atomics into lr-sized differences on
    # near-zero gradients, so two identical runs do not end bit-equal: the un-synchronised run must be
    # no further from a synchronised one than two synchronised runs are from each other
    def dist(x, y):
        num = sum(((a - b).double() ** 2).sum().item() for a, b in zip(finals[x], finals[y]))
        den = sum((b.double() ** 2).sum().item() for b in finals[y])
        return math.sqrt(num / den)
    noise = dist("synced2", "synced")
    assert dist("free", "synced") <= 3 * noise + 1e-6, (dist("free", "synced"), noise)


def test_forward_without_backward_does_not_stall_training():
    """ADVICE r1 (engine/autograd.py): a grad-enabled student forward that is never back-propagated
    (evaluation without no_grad, an exception before backward) must not block later steps; if its
    output is still alive at optimiser time the step fails loudly instead of silently skipping."""
    from dino_amd.utils.train_utils import train_step
    nsteps = 3
    crops = [c.to(DEV) for c in _crops(TINY["batch"], "orphan:img")]
    lr, wd, mom, _ = schedules(TINY["batch"], nsteps)
    ref_student, ref_teacher = build_pair(TINY, torch.float32)
    loss_fn, opt = _trainer(ref_student, ref_teacher, nsteps, len(crops))
    ref = [train_step(ref_student, ref_teacher, loss_fn, opt, crops, it, lr, wd, mom, 3.0, 0)[0].item()
           for it in range(nsteps)]
    student, teacher = build_pair(TINY, torch.float32)
    loss_fn, opt = _trainer(student, teacher, nsteps, len(crops))
    got = []
    for it in range(nsteps):
        student(crops)                     # recorded forward whose output is dropped at once
        out = student(crops[:2])           # ... and one that is overwritten by the next statement
        out = None
        got.append(train_step(student, teacher, loss_fn, opt, crops, it, lr, wd, mom, 3.0, 0)[0].item())
        assert all(p.grad is not None for p in student.parameters())
    assert np.allclose(got, ref, rtol=1e-5), (got, ref)
    held = student(crops)                  # still referenced when the optimiser runs
    with pytest.raises(RuntimeError, match="never back-propagated"):
        train_step(student, teacher, loss_fn, opt, crops, 0, lr, wd, mom, 3.0, 0)
    del held


def test_get_last_selfattention_matches_reference_golden(golden_dir):
    """VisionTransformer.get_last_selfattention (reference :385-392): [B, heads, N, N] probabilities of
    the last block against the REFERENCE's own output (tests/golden/ops.npz), fp32 1e-4 and bf16 2e-2."""
    g = np.load(os.path.join(golden_dir, "ops.npz"), allow_pickle=False)
    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 2e-2)):
        student, _ = build_pair(TINY, dtype)
        bb = student.backbone
        bb.compute_dtype = dtype
        sd = {k[len("backbone."):]: v.detach().clone() for k, v in student.state_dict().items()
              if k.startswith("backbone.")}
        alone = type(bb)(img_size=TINY["vit"]["img_size"], patch_size=TINY["vit"]["patch_size"],
                         embed_dim=TINY["vit"]["embed_dim"], depth=TINY["vit"]["depth"],
                         num_heads=TINY["vit"]["num_heads"], mlp_ratio=4, qkv_bias=True,
                         norm_layer=__import__("functools").partial(torch.nn.LayerNorm, eps=1e-6))
        alone.compute_dtype = dtype
        alone.load_state_dict(sd)
        alone = alone.to(DEV).eval()
        for size in (56, 24):
            x = detfill.det_normal((2, 3, size, size), f"lastattn:img{size}").to(DEV)
            got = alone.get_last_selfattention(x)
            want = g[f"lastattn/{size}"]
            assert tuple(got.shape) == want.shape
            assert _rel(got.float(), want) < tol, (dtype, size, _rel(got.float(), want))
            rows = got.float().sum(-1)
            assert torch.allclose(rows, torch.ones_like(rows), atol=2e-2 if dtype == torch.bfloat16 else 1e-5)


def test_multicrop_wrapper_with_features_contract():
    """MultiCropWrapperWithFeatures (reference models/__init__.py:49-103): forward(x) -> logits,
    forward(x, return_features=True) -> (logits, CLS features); same rows as MultiCropWrapper."""
    from dino_amd.components import DINOLoss  # noqa: F401  (the reference's import form)
    from dino_amd.models import MultiCropWrapper, MultiCropWrapperWithFeatures
    student, _ = build_pair(TINY, torch.float32)
    wf = MultiCropWrapperWithFeatures(student.backbone, student.head, compute_dtype=torch.float32).to(DEV)
    assert isinstance(wf, MultiCropWrapper) and MultiCropWrapperWithFeatures is not MultiCropWrapper
    crops = [c.to(DEV) for c in _crops(TINY["batch"], "wf:img")]
    with torch.no_grad():
        plain = student(crops)
        out = wf(crops)
        out2, feats = wf(crops, return_features=True)
    assert torch.equal(out, out2) and _rel(out, plain) < 1e-6
    assert feats.shape == (len(crops) * TINY["batch"], TINY["vit"]["embed_dim"])
    wf.backbone.compute_dtype = torch.float32
    with torch.no_grad():
        cls_g = wf.backbone(torch.cat(crops[:2]))
    assert _rel(feats[:2 * TINY["batch"]], cls_g) < 1e-5


def test_checkpoint_layout_matches_the_reference_manifest(golden_dir, tmp_path):
    """A checkpoint written by dino_amd (the dict of reference train_utils.py:176-188) has the
    REFERENCE's keys, order, shapes and dtypes for student / teacher / dino_loss / optimizer -- the
    manifest in tests/golden/ops.npz was taken from the reference's own state_dict()s (ViT-S/16,
    K=65536, torch.optim.AdamW after one step), so the files interchange."""
    import json
    from dino_amd.utils.train_utils import train_step
    from tests.golden_spec import FULL
    man = json.loads(str(np.load(os.path.join(golden_dir, "ops.npz"), allow_pickle=False)["ckpt/manifest_json"]))
    spec = dict(FULL, batch=2, sizes=[224, 224, 96, 96])
    student, teacher = build_pair(spec, torch.bfloat16)
    crops = [torch.randn(2, 3, s, s, device=DEV) for s in spec["sizes"]]
    from dino_amd.components import DINOLoss
    from dino_amd.optim import FusedAdamW
    from dino_amd.utils import get_params_groups
    loss_fn = DINOLoss(65536, len(crops), 0.04, 0.04, 0, 4).to(DEV)
    opt = FusedAdamW(get_params_groups(student), student, teacher)
    lr, wd, mom, _ = schedules(2, 4)
    train_step(student, teacher, loss_fn, opt, crops, 0, lr, wd, mom, 3.0, 0)
    path = tmp_path / "iter_000000.pt"
    torch.save({"iteration": 0, "student": student.state_dict(), "teacher": teacher.state_dict(),
                "optimizer": opt.state_dict(), "dino_loss": loss_fn.state_dict()}, path)
    ck = torch.load(path, map_location="cpu", weights_only=True)

    def info(sd):
        return [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]
    assert info(ck["student"]) == man["student"]
    assert info(ck["teacher"]) == man["student"]
    assert info(ck["dino_loss"]) == man["dino_loss"]
    osd = ck["optimizer"]
    assert [{"keys": sorted(g.keys()), "params": g["params"]} for g in osd["param_groups"]] == man["optimizer_groups"]
    got_state = {str(i): {k: [list(v.shape), str(v.dtype)] for k, v in st.items()} for i, st in osd["state"].items()}
    assert got_state == man["optimizer_state"]
    assert all(float(st["step"]) == 1.0 for st in osd["state"].values())
    # and a stock torch.optim.AdamW over the same parameters accepts the file
    stock = torch.optim.AdamW(get_params_groups(student))
    stock.load_state_dict(osd)


def test_optimizer_rebind_keeps_the_step_counts():
    """FusedAdamW keeps AdamW's per-parameter step count in a row table; when the student's store is re-materialised
    (student.to(), the data-parallel re-bind) the optimiser binds again and must carry the counts over -- restarting
    the bias corrections at t = 1 on carried-over moments would spike the effective step size."""
    from dino_amd.utils.train_utils import train_step
    nsteps = 4
    crops = [c.to(DEV) for c in _crops(TINY["batch"], "rebind:img")]
    lr, wd, mom, _ = schedules(TINY["batch"], nsteps)
    s, t = build_pair(TINY, torch.float32)
    lf, op = _trainer(s, t, nsteps, len(crops))
    for it in range(3):
        train_step(s, t, lf, op, crops, it, lr, wd, mom, 3.0, TINY["freeze_iter"])
    before = op._steps.copy()          # 3 everywhere, 2 for the last layer (its gradient is cancelled at it = 0)
    assert before.max() == 3.0 and before.min() >= 2.0
    s.__dict__.pop("_dino_store")      # the next use builds a new store, as after student.to(...)
    st_new = op._bind()
    assert st_new is s.store() and np.array_equal(op._steps, before), op._steps
    train_step(s, t, lf, op, crops, 3, lr, wd, mom, 3.0, TINY["freeze_iter"])
    got = sorted(float(st["step"]) for st in op.state_dict()["state"].values())
    assert got == sorted(before + 1.0), got


def test_teacher_column_sums_only_ride_on_train_mode_passes():
    """The no-grad head pass of a module in train mode (the teacher of a training step) leaves the column sums of its
    logits on the tensor for DINOLoss.update_center; an eval-mode pass (feature extraction, inspection) returns plain
    logits and pays for neither the [K] buffer nor the column-sum epilogue.  A tensor written into after the head
    is not trusted."""
    from dino_amd.components.loss import DINOLoss
    s, t = build_pair(TINY, torch.float32)
    crops = [c.to(DEV) for c in _crops(TINY["batch"], "colsum:img")]
    with torch.no_grad():
        out_train = t(crops[:2])
        t.eval()
        out_eval = t(crops[:2])
        t.train()
    assert hasattr(out_train, "_dino_colsum") and not hasattr(out_eval, "_dino_colsum")
    assert _rel(out_eval, out_train) < 1e-6
    sums, version = out_train._dino_colsum
    assert version == out_train._version and _rel(sums, out_train.double().sum(0)) < 1e-5
    # modified in place after the head: DINOLoss recomputes the sums from the tensor it is given
    lf = DINOLoss(TINY["head"]["out_dim"], len(crops), 0.04, 0.04, 0, 2).to(DEV)
    with torch.no_grad():
        out_train.mul_(0.5)
    lf(s(crops), out_train, 0)
    B2 = out_train.shape[0]
    want = out_train.double().sum(0, keepdim=True) / B2 * (1 - lf.center_momentum)
    assert _rel(lf.center, want) < 1e-5
