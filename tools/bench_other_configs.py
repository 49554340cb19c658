#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations at full depth on one GPU (bench.py measures
configs[1] only): ViT-B/16 (configs[2]), ViT-S/8 (configs[3]), ViT-L/14 (configs[4]); per-GPU batches as
SURVEY.md 8(d) proposes.  Usage: python tools/bench_other_configs.py [vitb16|vits8|vitl14] [batch]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dino_amd
from dino_amd.components.loss import DINOLoss
from dino_amd.models import MultiCropWrapper, vision_transformer as vits
from dino_amd.optim import FusedAdamW
from dino_amd.utils import cosine_scheduler, fix_random_seeds, get_params_groups
from dino_amd.utils.train_utils import train_step

CFG = {"vitb16": (vits.vit_base, dict(img_size=256, patch_size=16), 768, 128, 474.11),
       "vits8": (vits.vit_small, dict(img_size=256, patch_size=8), 384, 64, 557.18),
       "vitl14": (vits.vit_large, dict(img_size=256, patch_size=14), 1024, 32, 1973.74)}
name = sys.argv[1] if len(sys.argv) > 1 else "vitb16"
ctor, arch, D, B, gflop = CFG[name]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
steps, warm, K = 6, 2, 65536
dino_amd.set_default_dtype(torch.bfloat16)
fix_random_seeds(0)
student = MultiCropWrapper(ctor(drop_path_rate=0.1, **arch), vits.DINOHead(D, K, norm_last_layer=False)).cuda()
teacher = MultiCropWrapper(ctor(**arch), vits.DINOHead(D, K)).cuda()
teacher.load_state_dict(student.state_dict())
for p in teacher.parameters():
    p.requires_grad = False
total = steps + warm + 2
loss_fn = DINOLoss(K, 12, 0.04, 0.04, 0, total).cuda()
opt = FusedAdamW(get_params_groups(student), student, teacher)
lr = cosine_scheduler(5e-4 * B / 256.0, 1e-6, 1, total)
wd = cosine_scheduler(0.04, 0.4, 1, total)
mom = cosine_scheduler(0.996, 1, 1, total)
g = torch.Generator().manual_seed(1234)
crops = [torch.randn(B, 3, 224, 224, generator=g).cuda() for _ in range(2)] + \
        [torch.randn(B, 3, 96, 96, generator=g).cuda() for _ in range(10)]
for it in range(warm):
    loss, _ = train_step(student, teacher, loss_fn, opt, crops, it, lr, wd, mom, 3.0, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(warm, warm + steps):
    loss, _ = train_step(student, teacher, loss_fn, opt, crops, it, lr, wd, mom, 3.0, 0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"config": name, "batch": B, "ms_per_step": round(dt * 1e3, 2), "images_per_s": round(B / dt, 1),
                  "model_tflops_per_s": round(B / dt * gflop / 1e3, 1), "loss": round(loss.item(), 4),
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
if os.environ.get("PRINT_FOLD"):
    import dino_amd.engine.vit as E
    print("LayerNorm fold by (device, tokens, width, hidden):", E._FOLD_OK, flush=True)
