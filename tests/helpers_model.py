"""Builders shared by the GPU model tests and smoke(): the dino_amd student/teacher pair for a
golden_spec case with deterministic weights, and the matching oracle trainer."""
import torch

from oracle import detfill
from oracle import dino_oracle as O


def build_pair(spec, dtype, device="cuda", drop_path_rate=0.0, norm_last_layer=False):
    from dino_amd.models import MultiCropWrapper, vision_transformer as vits
    from functools import partial
    import torch.nn as nn
    v, h = spec["vit"], spec["head"]

    def make(dpr):
        bb = vits.VisionTransformer(img_size=v["img_size"], patch_size=v["patch_size"],
                                    embed_dim=v["embed_dim"], depth=v["depth"],
                                    num_heads=v["num_heads"], mlp_ratio=4, qkv_bias=True,
                                    norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_path_rate=dpr)
        hd = vits.DINOHead(v["embed_dim"], h["out_dim"], hidden_dim=h["hidden_dim"],
                           bottleneck_dim=h["bottleneck_dim"], norm_last_layer=norm_last_layer)
        return MultiCropWrapper(bb, hd, compute_dtype=dtype)
    student, teacher = make(drop_path_rate), make(0.0)
    detfill.fill_state_dict_(dict(student.named_parameters()), tag=spec["tag"])
    student, teacher = student.to(device), teacher.to(device)
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    return student, teacher


def oracle_cfgs(spec):
    v, h = spec["vit"], spec["head"]
    return O.VitCfg(**v), O.HeadCfg(in_dim=v["embed_dim"], **h)


def oracle_trainer(spec, nsteps):
    vit, head = oracle_cfgs(spec)
    sd = detfill.fill_state_dict_(O.new_state_dict(vit, head), tag=spec["tag"])
    B = spec["batch"]
    hp = O.StepHP(ncrops=len(spec["sizes"]), clip_grad=3.0, freeze_last_layer_iter=spec["freeze_iter"])
    lr, wd, mom, tt = schedules(B, nsteps, spec.get("lr"), spec.get("ttemp", (0.04, 0.04, 0)))
    return O.OracleTrainer(vit, head, hp, sd, lr, wd, mom, tt)


def schedules(B, nsteps, lr=None, ttemp=(0.04, 0.04, 0)):
    return (O.cosine_scheduler(5e-4 * B / 256.0 if lr is None else lr, 1e-6, 1, nsteps),
            O.cosine_scheduler(0.04, 0.4, 1, nsteps),
            O.cosine_scheduler(0.996, 1, 1, nsteps),
            O.teacher_temp_schedule(*ttemp, nsteps))
