#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation on CPU.

Runs only in the authoring container (needs /root/reference on PYTHONPATH; the
reference never travels to the GPU box).  The committed .npz files hold data only:
inputs are regenerated from oracle/detfill.py (seed-free), outputs are what the
reference modules produced.  Usage:  python tools/make_golden.py
"""
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DINO_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# dino.utils imports wandb / omegaconf at module top (dino/utils/utils.py:4,13); neither is
# installed and neither is used by the functions we call -> inert in-process stand-ins.
if "wandb" not in sys.modules:
    sys.modules["wandb"] = types.ModuleType("wandb")
if "omegaconf" not in sys.modules:
    import yaml
    oc = types.ModuleType("omegaconf")
    oc.DictConfig = dict

    class _OC:
        load = staticmethod(lambda f: yaml.safe_load(open(f)))
        create = staticmethod(lambda x: x)
    oc.OmegaConf = _OC
    sys.modules["omegaconf"] = oc

import dino.models.vision_transformer as rvits  # noqa: E402
from dino.models import MultiCropWrapper as RefWrapper  # noqa: E402
from dino.components.loss import DINOLoss as RefLoss  # noqa: E402
from dino.utils.utils import (cosine_scheduler as ref_cosine, get_params_groups as ref_groups,  # noqa: E402
                              compute_gradient_stats as ref_gstats,
                              cancel_gradients_last_layer as ref_cancel)

from oracle import detfill  # noqa: E402
from tests.golden_spec import (CURVE, TINY, FULL, MULTI, OTHER_FULL, other_crops, digest, tiny_crops, full_crops, multi_crops,  # noqa: E402
                               SCHED_CASES, POS_CASES, LOSS_CASE)

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def build_ref(spec, drop_path_rate=0.0):
    v = spec["vit"]
    bb = rvits.VisionTransformer(img_size=v["img_size"], patch_size=v["patch_size"],
                                 embed_dim=v["embed_dim"], depth=v["depth"],
                                 num_heads=v["num_heads"], mlp_ratio=4, qkv_bias=True,
                                 norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                 drop_path_rate=drop_path_rate)
    h = spec["head"]
    hd = rvits.DINOHead(v["embed_dim"], h["out_dim"], hidden_dim=h["hidden_dim"],
                        bottleneck_dim=h["bottleneck_dim"], norm_last_layer=False)
    return RefWrapper(bb, hd)


def run_steps(spec, crops, nsteps, full_tensors, act_digests=False):
    """Reference modules driven in the order of dino/utils/train_utils.py:52-162."""
    student, teacher = build_ref(spec), build_ref(spec)
    detfill.fill_state_dict_(dict(student.named_parameters()), tag=spec["tag"])
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    ncrops = len(crops)
    B = crops[0].shape[0]
    loss_fn = RefLoss(spec["head"]["out_dim"], ncrops, *spec.get("ttemp", (0.04, 0.04, 0)), nsteps)
    opt = torch.optim.AdamW(ref_groups(student))
    lr = ref_cosine(spec.get("lr", 5e-4 * B / 256.0), 1e-6, 1, nsteps, warmup_epochs=0)
    wd = ref_cosine(0.04, 0.4, 1, nsteps)
    mom = ref_cosine(0.996, 1, 1, nsteps)
    freeze_iter = spec.get("freeze_iter", 0)
    out = {"param_names": np.array([n for n, _ in student.named_parameters()])}
    acts = {}
    if full_tensors or act_digests:
        def hook(name):
            def f(_m, _i, o):
                acts.setdefault(name, []).append(o.detach().clone())
            return f
        for i, blk in enumerate(student.backbone.blocks):
            blk.register_forward_hook(hook(f"block{i}"))
    for it in range(nsteps):
        acts.clear()
        for i, g in enumerate(opt.param_groups):
            g["lr"] = lr[it]
            if i == 0:
                g["weight_decay"] = wd[it]
        t_out = teacher(crops[:2])
        s_out = student(crops)
        s_out.retain_grad()
        loss = loss_fn(s_out, t_out, it)
        opt.zero_grad()
        loss.backward()
        raw = {n: p.grad.detach().clone() for n, p in student.named_parameters()}
        stats = ref_gstats(student, clip=3.0, model_name="student")
        ref_cancel(it, student, freeze_iter)
        opt.step()
        with torch.no_grad():
            m = mom[it]
            for pq, pk in zip(student.parameters(), teacher.parameters()):
                pk.data.mul_(m).add_((1 - m) * pq.detach().data)
        out[f"loss{it}"] = np.float64(loss.item())
        out[f"center{it}"] = loss_fn.center.detach().numpy().copy() if full_tensors \
            else digest(loss_fn.center)
        out[f"gstats{it}"] = np.array([stats["student_grad_norm"], stats["student_grad_max"],
                                       stats["student_clip_fraction"]], dtype=np.float64)
        out[f"gnorms{it}"] = np.array([raw[n].norm(2).item() for n in raw], dtype=np.float64)
        # the same norms accumulated in float64: torch's fp32 CPU norm of a 16.7 M-element tensor
        # (weight_v) is itself off by ~2e-3, which would hide a gradient error of that size
        out[f"gnorms64_{it}"] = np.array([raw[n].double().norm(2).item() for n in raw], dtype=np.float64)
        out[f"student_digest{it}"] = np.stack([digest(p) for p in student.parameters()])
        out[f"teacher_digest{it}"] = np.stack([digest(p) for p in teacher.parameters()])
        if full_tensors and it == 0:
            out["student_out0"] = s_out.detach().numpy().copy()
            out["teacher_out0"] = t_out.detach().numpy().copy()
            out["dstudent_out0"] = s_out.grad.detach().numpy().copy()
            for name, lst in acts.items():  # one entry per backbone pass (global, local)
                for pi, a in enumerate(lst):
                    out[f"act0_{name}_pass{pi}"] = a.numpy().copy()
            for n, g in raw.items():
                out[f"grad0/{n}"] = g.numpy().copy()
        if act_digests and it == 0:
            # fingerprints instead of tensors: per block and backbone pass, logits, d loss / d logits,
            # every parameter gradient
            out["student_out_digest0"] = digest(s_out)
            out["teacher_out_digest0"] = digest(t_out)
            out["dstudent_out_digest0"] = digest(s_out.grad)
            for name, lst in acts.items():
                for pi, a in enumerate(lst):
                    out[f"actd0_{name}_pass{pi}"] = digest(a)
            out["grad_digest0"] = np.stack([digest(raw[n]) for n in raw])
        print(f"[{spec['tag']}] it {it} loss {loss.item():.8f} gnorm {stats['student_grad_norm']:.6e}",
              flush=True)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:] or ["tiny", "ops", "full", "multi"]
    for name in OTHER_FULL:
        if name in only:
            # (5) BASELINE configs #3-#5 at full depth, multi-crop, K=65536 -- digests only, 2 steps
            np.savez_compressed(os.path.join(OUT, f"{name}_multicrop_steps.npz"),
                                **run_steps(OTHER_FULL[name], other_crops(name), 2, full_tensors=False,
                                            act_digests=True))
    if "multi" in only:
        # (4) BASELINE config #2's shape (ViT-S/16 full depth, 2x224 + 10x96, K=65536) at B=2
        np.savez_compressed(os.path.join(OUT, "vits16_multicrop_steps.npz"),
                            **run_steps(MULTI, multi_crops(), 3, full_tensors=False, act_digests=True))
    if "full" in only:
        # (3) full-size config #1: ViT-S/16, 2x224, B=8, K=65536 -- scalars / digests only
        np.savez_compressed(os.path.join(OUT, "vits16_b8_steps.npz"),
                            **run_steps(FULL, full_crops(), 3, full_tensors=False))
    if "curve" in only:
        # (6) a loss curve: the tiny model for 24 steps (scalars and digests per step)
        np.savez_compressed(os.path.join(OUT, "tiny_curve.npz"),
                            **run_steps(CURVE, tiny_crops(), CURVE["nsteps"], full_tensors=False))
    if "tiny" in only:
        # (1) tiny model, full tensors
        np.savez_compressed(os.path.join(OUT, "tiny_steps.npz"),
                            **run_steps(TINY, tiny_crops(), 3, full_tensors=True))
    if "ops" not in only:
        return
    # (2) op-level fixtures
    ops = {}
    for name, (b, f, n, nit, we, sv) in SCHED_CASES.items():
        ops[f"cos/{name}"] = ref_cosine(b, f, n, nit, warmup_epochs=we, start_warmup_value=sv)
    l = RefLoss(16, 4, 0.02, 0.07, 0.3, 50)
    ops["ttemp/warm"] = np.asarray(l.teacher_temp_schedule)
    for name, (img, P, D, crop) in POS_CASES.items():
        vt = rvits.VisionTransformer(img_size=img, patch_size=P, embed_dim=D, depth=1, num_heads=1)
        with torch.no_grad():
            vt.pos_embed.copy_(detfill.det_normal(vt.pos_embed.shape, f"pos:{name}"))
            npatch = (crop // P) ** 2
            x = torch.zeros(1, npatch + 1, D)
            ops[f"pos/{name}"] = vt.interpolate_pos_encoding(x, crop, crop).numpy().copy()
    K, B, nc = LOSS_CASE
    s = detfill.det_normal((nc * B, K), "loss:s", std=2.0).requires_grad_(True)
    t = detfill.det_normal((2 * B, K), "loss:t", std=2.0)
    lf = RefLoss(K, nc, 0.04, 0.04, 0, 10)
    lf.center.copy_(detfill.det_normal((1, K), "loss:c", std=0.5))
    val = lf(s, t, 0)
    val.backward()
    ops["loss/value"] = np.float64(val.item())
    ops["loss/grad_digest"] = digest(s.grad)
    ops["loss/center_digest"] = digest(lf.center)
    # get_last_selfattention of the reference backbone (vision_transformer.py:385-392) on the tiny
    # model: native 7x7 grid (56 px) and the interpolated 3x3 grid (24 px)
    tiny_ref = build_ref(TINY)
    detfill.fill_state_dict_(dict(tiny_ref.named_parameters()), tag=TINY["tag"])
    with torch.no_grad():
        for size in (56, 24):
            x = detfill.det_normal((2, 3, size, size), f"lastattn:img{size}")
            ops[f"lastattn/{size}"] = tiny_ref.backbone.get_last_selfattention(x).numpy().copy()
    wrapped = build_ref(FULL)
    reg, noreg = ref_groups(wrapped)
    ids = {id(p): n for n, p in wrapped.named_parameters()}
    ops["groups/reg"] = np.array([ids[id(p)] for p in reg["params"]])
    ops["groups/noreg"] = np.array([ids[id(p)] for p in noreg["params"]])
    ops["groups/shapes"] = np.array([str(tuple(p.shape)) for _, p in wrapped.named_parameters()])
    # key / shape / dtype manifest of the checkpoint entries the reference writes
    # (dino/utils/train_utils.py:176-188): student.state_dict(), DINOLoss.state_dict() and
    # torch.optim.AdamW.state_dict() after one step, ViT-S/16 + 65536-prototype head
    import json
    loss_ref = RefLoss(FULL["head"]["out_dim"], 12, 0.04, 0.04, 0, 10)
    opt_ref = torch.optim.AdamW(ref_groups(wrapped))
    for p_ in wrapped.parameters():
        p_.grad = torch.zeros_like(p_)
    opt_ref.step()

    def tinfo(t):
        return [list(t.shape), str(t.dtype)]
    osd = opt_ref.state_dict()
    manifest = {
        "student": [[k, *tinfo(v)] for k, v in wrapped.state_dict().items()],
        "dino_loss": [[k, *tinfo(v)] for k, v in loss_ref.state_dict().items()],
        "optimizer_state": {str(i): {k: tinfo(v) for k, v in st.items()} for i, st in osd["state"].items()},
        "optimizer_groups": [{"keys": sorted(g.keys()), "params": g["params"]} for g in osd["param_groups"]],
    }
    ops["ckpt/manifest_json"] = np.array(json.dumps(manifest))
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **ops)


if __name__ == "__main__":
    main()
