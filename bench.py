#!/usr/bin/env python3
"""Headline benchmark: train images/sec of the DINO step (BASELINE.json metric) on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: ViT-S/16, 2x224 + 10x96 multi-crop, out_dim 65536, bf16,
batch 256 per GPU, drop_path 0.1, clip 3.0, AdamW, EMA teacher; synthetic N(0,1) crops resident
in HBM before the timed region, random-init weights.  Weak scaling: per-GPU batch fixed.
Rank 0 prints ONE JSON line (plus a `roofline` object for the dominant kernel, measured with
HIP events on the launch stream, and a `cpu_baseline` object: the CPU oracle timed on this
box's host cores on a bounded sample of the same workload, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ARCH = dict(img_size=256, patch_size=16)
OUT_DIM = 65536
NLOCAL = 10
GFLOP_PER_IMAGE = 123.80  # SURVEY.md section 8(d), ViT-S/16 2x224+10x96, K=65536


def nt_traffic_from_profiles():
    """HBM bytes per NT GEMM launch (launch-weighted) from the newest committed counter summary
    profiles/r*_hbm_traffic_per_kernel_*.csv (tools/hbm_traffic.py: separate rocprofv3 --pmc FETCH_SIZE
    and WRITE_SIZE passes over this bench, FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md
    prescribes).  bench.py cannot run counter passes itself; None if no summary travelled with the tree."""
    import csv
    import glob
    import re

    def key(f):
        m = re.search(r"r(\d+)_hbm_traffic_per_kernel(?:_v(\d+))?", os.path.basename(f))
        return (int(m.group(1)), int(m.group(2) or 0)) if m else (-1, -1)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_per_kernel*.csv")), key=key)
    if not files:
        return None, None
    f = files[-1]
    mb = n = 0.0
    with open(f) as fh:
        rd = csv.reader(fh)
        header = next(rd)
        col = [i for i, h in enumerate(header) if h.startswith("avg HBM MB per launch")]
        if not col:                       # older summaries are in MiB and lack the column
            return None, None
        for row in rd:
            if row and "gemm_nt_bf16" in row[0]:
                mb += float(row[col[0]]) * float(row[1])
                n += float(row[1])
    return (mb / n * 1e6 if n else None), os.path.relpath(f, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (images)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-h2d-leg", action="store_true",
                    help="skip the second timed region (N=1 only) that feeds every step's crops from pinned "
                         "host memory -- SURVEY section 8(d)'s step with the H2D copy inside")
    ap.add_argument("--no-whole-block-leg", action="store_true",
                    help="skip the short extra leg (N=1 only) that times the step with the whole last transformer "
                         "block computed (reported beside `value` as with_whole_last_block)")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=6)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 dry run on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL "
                         "(checks the multi-rank code path; the number it prints means nothing)")
    ap.add_argument("--feed", default="resident", choices=["resident", "f32", "u8", "aug"],
                    help="resident (the contract's number): crops already in HBM.  f32 / u8: every step's "
                         "crops come from pinned host memory through dino_amd.data.CropPrefetcher "
                         "(u8: uint8 crops, ToTensor+Normalize on the device) -- the PCIe-inclusive "
                         "rate quoted in DESIGN.md section 7")
    ap.add_argument("--occupy", default=None, metavar="N,US[,EVERY]",
                    help="one-GPU stand-in for overlapped gradient all-reduces: every EVERY-th (default 3) block of "
                         "the backward pass a kernel on a side stream holds N CUs for US microseconds "
                         "(dino_diag_occupy_cus), as RCCL's kernels would while a bucket is reduced.  With "
                         "DINO_NT_QUEUE=0/1 it shows what the work queue of the GEMMs buys (DESIGN.md section 5)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU-only check of the --gpus N launcher: every rank joins a gloo group, rank 0 prints a "
                         "JSON line (tests/test_host_logic.py)")
    return ap.parse_args()


def make_crops(B, device, rank):
    g = torch.Generator().manual_seed(1234 + rank)
    crops = [torch.randn(B, 3, 224, 224, generator=g) for _ in range(2)]
    crops += [torch.randn(B, 3, 96, 96, generator=g) for _ in range(NLOCAL)]
    return [c.to(device) for c in crops]


def cpu_baseline(batch, steps):
    """The CPU oracle (fp32 restatement pinned to the reference, oracle/dino_oracle.py) on the
    same multi-crop workload at a bounded batch; first step is warm-up."""
    from oracle import dino_oracle as O
    from oracle import detfill
    vit = O.VitCfg(img_size=256, patch_size=16, embed_dim=384, depth=12, num_heads=6)
    head = O.HeadCfg(in_dim=384, out_dim=OUT_DIM)
    sd = detfill.fill_state_dict_(O.new_state_dict(vit, head), tag="bench")
    hp = O.StepHP(ncrops=2 + NLOCAL, clip_grad=3.0, freeze_last_layer_iter=0)
    lr, wd, mom, tt = O.make_schedules(steps, batch)
    tr = O.OracleTrainer(vit, head, hp, sd, lr, wd, mom, tt)
    g = torch.Generator().manual_seed(7)
    crops = [torch.randn(batch, 3, 224, 224, generator=g) for _ in range(2)]
    crops += [torch.randn(batch, 3, 96, 96, generator=g) for _ in range(NLOCAL)]
    tr.step(crops, 0)
    t0 = time.perf_counter()
    for it in range(1, steps):
        tr.step(crops, it)
    dt = time.perf_counter() - t0
    n = steps - 1
    return {"value": round(batch * n / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n} steps of the same ViT-S/16 2x224+10x96 K=65536 step at batch {batch}, "
                      f"fp32 torch-CPU oracle ({dt:.1f} s)"}


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv=None, stream=None):
    """Run this script as args.gpus ranks under torch.distributed.run (child processes, 127.0.0.1
    rendez-vous) and relay rank 0's JSON line to stdout.  Returns the launcher's exit code."""
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    out = stream or sys.stdout
    got = False
    for ln in p.stdout:
        if ln.lstrip().startswith("{"):
            out.write(ln)
            out.flush()
            got = True
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc == 0 and not got:
        sys.stderr.write("bench.py: the ranks exited 0 but rank 0 printed no JSON line\n")
        return 1
    return rc


class OpTimer:
    """Brackets kernel launches with HIP events recorded on the launch stream, on every `every`-th
    step of the timed region (`active`; the event records cost 1-2 % of a step).  Classes:
      nt        every bf16 NT GEMM launch (the dominant kernel class); FLOPs = 2*M*N*K
      loss      dino_loss_fwd_bwd (teacher statistics + student pass + centre column sums)
      adamw     dino_adamw_ema
      ln_fwd / ln_bwd   dino_layernorm_fwd / _bwd over whole token matrices
      attn_fwd / attn_bwd
    For the HBM-bound classes `work` is SURVEY section 8(d)'s ALGORITHMIC bytes, not counter bytes."""

    def __init__(self, ops, dtype, every=10):
        self.ops, self.dtype, self.every, self.active = ops, dtype, every, True
        self.rec = {}
        self._orig = {}

    def _bracket(self, name, work, fn, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        c = self.rec.setdefault(name, {"events": [], "work": 0.0, "n": 0})
        c["events"].append((e0, e1))
        c["work"] += work
        c["n"] += 1
        return r

    def __enter__(self):
        ops, o = self.ops, self._orig
        for k in ("gemm", "loss_fwd_bwd", "adamw_ema", "layernorm_fwd", "layernorm_bwd", "layernorm_bwd_xn", "attn_fwd",
                  "attn_bwd"):
            o[k] = getattr(ops, k)

        def gemm(A, B, out, **kw):
            if not self.active or kw.get("trans") or A.dtype != self.dtype:
                return o["gemm"](A, B, out, **kw)
            M, N, K = kw.get("M") or A.shape[0], kw.get("N") or B.shape[0], kw.get("K") or A.shape[1]
            return self._bracket("nt", 2.0 * M * N * K, o["gemm"], A, B, out, **kw)

        def loss_fwd_bwd(student, teacher, center, *a, **kw):
            if not self.active:
                return o["loss_fwd_bwd"](student, teacher, center, *a, **kw)
            es = student.element_size()
            K = student.shape[-1]
            work = (2 * student.shape[0] + teacher.shape[0]) * K * es + 2 * K * 4
            return self._bracket("loss", work, o["loss_fwd_bwd"], student, teacher, center, *a, **kw)

        def adamw_ema(param, *a, **kw):
            if not self.active:
                return o["adamw_ema"](param, *a, **kw)
            # fp32 p, g, m, v, teacher read (20 B) + p, m, v, teacher written (16 B); the bf16 shadow
            # refresh (4 B more) is this implementation's own traffic and not counted
            return self._bracket("adamw", 36.0 * param.numel(), o["adamw_ema"], param, *a, **kw)

        def small(x):   # the last block's [crops, D] matrices (CLS rows only): not "whole token matrices"
            return x.dim() == 2 and x.shape[0] < 16384

        def layernorm_fwd(x, *a, **kw):
            if not self.active or kw.get("rows") is not None or small(x):
                return o["layernorm_fwd"](x, *a, **kw)
            return self._bracket("ln_fwd", 2.0 * x.numel() * x.element_size(), o["layernorm_fwd"], x, *a, **kw)

        def layernorm_bwd(dy, x, *a, **kw):
            if not self.active or kw.get("rows") is not None or small(x):
                return o["layernorm_bwd"](dy, x, *a, **kw)
            # dy, x read + dx written (3) + the residual-stream gradient added in the same pass (1)
            nstream = 4.0 if kw.get("dres") is not None else 3.0
            return self._bracket("ln_bwd", nstream * x.numel() * x.element_size(), o["layernorm_bwd"], dy, x,
                                 *a, **kw)

        def layernorm_bwd_xn(dy, x, *a, **kw):
            if not self.active:
                return o["layernorm_bwd_xn"](dy, x, *a, **kw)
            # the form the folded-LayerNorm step uses: dy, x (+ the residual-stream gradient) read, dx and the
            # normalised rows written
            nstream = 5.0 if kw.get("dres") is not None else 4.0
            return self._bracket("ln_bwd", nstream * x.numel() * x.element_size(), o["layernorm_bwd_xn"], dy, x,
                                 *a, **kw)

        def attn_fwd(qkv, Bc, N, H, *a, **kw):
            if not self.active:
                return o["attn_fwd"](qkv, Bc, N, H, *a, **kw)
            return self._bracket(f"attn_fwd_N{N}", 4.0 * N * N * 64 * Bc * H, o["attn_fwd"], qkv, Bc, N, H, *a, **kw)

        def attn_bwd(qkv, out, dout, lse, Bc, N, H, *a, **kw):
            if not self.active:
                return o["attn_bwd"](qkv, out, dout, lse, Bc, N, H, *a, **kw)
            return self._bracket(f"attn_bwd_N{N}", 8.0 * N * N * 64 * Bc * H, o["attn_bwd"], qkv, out, dout, lse,
                                 Bc, N, H, *a, **kw)
        for k, f in dict(gemm=gemm, loss_fwd_bwd=loss_fwd_bwd, adamw_ema=adamw_ema, layernorm_fwd=layernorm_fwd,
                         layernorm_bwd=layernorm_bwd, layernorm_bwd_xn=layernorm_bwd_xn, attn_fwd=attn_fwd,
                         attn_bwd=attn_bwd).items():
            setattr(ops, k, f)
        return self

    def __exit__(self, *a):
        for k, f in self._orig.items():
            setattr(self.ops, k, f)

    def _ms(self, name):
        c = self.rec.get(name)
        if not c or not c["n"]:
            return None, None
        return sum(a.elapsed_time(b) for a, b in c["events"]), c

    def result(self):
        """(roofline object for the NT GEMM class, hbm_kernels, mfma_kernels)"""
        torch.cuda.synchronize()
        ms, c = self._ms("nt")
        if ms is None or ms <= 0:
            return None, None, None
        tflops = c["work"] / (ms * 1e-3) / 1e12
        # what an event pair reads around a ~2 us kernel in a busy stream: the part of
        # avg_launch_us that is launch gap, not kernel (rocprofv3's kernel duration excludes it).
        # Reported, not subtracted.
        x = torch.zeros(1, device="cuda")
        pairs = []
        for _ in range(200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            x.zero_()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        floor = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
        traffic, src = nt_traffic_from_profiles()
        roof = {"bound": "mfma", "kernel": "gemm_nt_bf16_v3 + gemm_nt_bf16_v2 (all bf16 NT GEMMs: fwd + dgrad)",
                "achieved": round(tflops, 2), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(tflops / 2500.0, 4), "traffic": traffic,
                "traffic_source": None if src is None else
                f"{src}: HBM bytes per launch, launch-weighted over the same kernels, from separate rocprofv3 --pmc "
                "FETCH_SIZE (x2, gfx950) and WRITE_SIZE passes over this bench (tools/hbm_traffic.py); "
                "not re-collected by this run",
                "launches": c["n"], "sampled": f"every {self.every}th step of the timed region (those steps on one stream)",
                "avg_launch_us": round(1e3 * ms / c["n"], 2),
                "event_pair_floor_us": round(1e3 * floor, 2),
                "avg_launch_gflop": round(c["work"] / c["n"] / 1e9, 3)}
        hbm, mfma = {}, {}
        for name in sorted(self.rec):
            if name == "nt":
                continue
            ms, c = self._ms(name)
            if ms is None or ms <= 0:
                continue
            us = 1e3 * ms / c["n"]
            per = c["work"] / c["n"]
            if name.startswith("attn"):
                tf = per / (us * 1e-6) / 1e12
                mfma[name] = {"avg_us": round(us, 1), "gflop": round(per / 1e9, 2), "tflops": round(tf, 1),
                              "frac_of_2500": round(tf / 2500.0, 4), "calls": c["n"]}
            else:
                tbs = per / (us * 1e-6) / 1e12
                hbm[name] = {"avg_us": round(us, 1), "algorithmic_MB": round(per / 1e6, 1),
                             "TBps": round(tbs, 3), "frac_of_8TBps": round(tbs / 8.0, 4), "calls": c["n"]}
        return roof, hbm, mfma


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU ourselves, the way the
        # reference's main.py:28-44 does.  Nothing in this process has touched the GPU yet (the ranks are
        # CHILD processes; a process that has initialised HIP must never be replaced by exec).
        sys.exit(launch_ranks(args))
    if world != args.gpus:
        sys.exit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    if args.launcher_selftest:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launcher_selftest": True, "n_gpus": world, "rank_sum": t.item()}), flush=True)
        dist.destroy_process_group()
        return
    import dino_amd
    import dino_amd.distributed as distributed
    from dino_amd import ops
    from dino_amd.components.loss import DINOLoss
    from dino_amd.models import MultiCropWrapper, vision_transformer as vits
    from dino_amd.optim import FusedAdamW
    from dino_amd.utils import cosine_scheduler, fix_random_seeds, get_params_groups
    from dino_amd.utils.train_utils import train_step

    if args.rehearse_on_one_gpu:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(120, exit=True)
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    # DINO_DP_FORCE=1 under torch.distributed.run with ONE rank: the whole data-parallel machinery
    # (RCCL process group, DataParallel's bucketed all-reduces, asynchronous centre all-reduce) on a
    # 1-GPU box -- every collective is the identity, the code path is the N-GPU one
    dp = world > 1 or (os.environ.get("DINO_DP_FORCE", "0") == "1" and "RANK" in os.environ)
    if dp:
        if args.rehearse_on_one_gpu:
            distributed.enable(backend="gloo", set_cuda_current_device=False)
        else:
            distributed.enable(overwrite=True)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dino_amd.set_default_dtype(dtype)
    fix_random_seeds(0)  # every rank builds the same student (reference: utils/config.py:79)
    B = args.batch
    student = MultiCropWrapper(vits.vit_small(drop_path_rate=0.1, **ARCH),
                               vits.DINOHead(384, OUT_DIM, norm_last_layer=False)).to(device)
    teacher = MultiCropWrapper(vits.vit_small(**ARCH), vits.DINOHead(384, OUT_DIM)).to(device)
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    if dp:
        student = distributed.DataParallel(student)
    total = args.steps + args.warmup + 4   # (schedule length; the extra legs clamp their index to its last entry)
    loss_fn = DINOLoss(OUT_DIM, 2 + NLOCAL, 0.04, 0.04, 0, total).to(device)
    opt = FusedAdamW(get_params_groups(student), student, teacher)
    lr = cosine_scheduler(5e-4 * B * world / 256.0, 1e-6, 1, total, warmup_epochs=0)
    wd = cosine_scheduler(0.04, 0.4, 1, total)
    mom = cosine_scheduler(0.996, 1, 1, total)
    freeze = int(0.01 * total)
    if args.feed == "resident":
        crops = make_crops(B, device, rank)
        feed = None
    elif args.feed == "aug":
        # the whole input side of the reference's step on the device: raw uint8 256 x 256 images from pinned
        # host memory (50 MB per batch), the multi-crop augmentation of dino/data/augmentations.py:75-157 as
        # HIP kernels on a side stream, one batch ahead
        from dino_amd.data import DeviceAugmentationDINO, DeviceAugmentingLoader
        g = torch.Generator().manual_seed(1234 + rank)
        raw = [torch.randint(0, 256, (B, 256, 256, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(2)]
        aug = DeviceAugmentationDINO(224, 96, (0.32, 1.0), (0.05, 0.32), NLOCAL, solarization=True, seed=rank)
        feed = iter(DeviceAugmentingLoader(((raw[i % 2], None) for i in range(total)), aug, device))
    else:
        from dino_amd.data import CropPrefetcher, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
        g = torch.Generator().manual_seed(1234 + rank)

        def host_crop(s):
            if args.feed == "u8":
                return torch.randint(0, 256, (B, s, s, 3), dtype=torch.uint8, generator=g).pin_memory()
            return torch.randn(B, 3, s, s, generator=g).pin_memory()
        host = [[host_crop(224) for _ in range(2)] + [host_crop(96) for _ in range(NLOCAL)] for _ in range(2)]
        feed = iter(CropPrefetcher(((host[i % 2], None) for i in range(total)), device,
                                   normalize=(IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)
                                   if args.feed == "u8" else None))

    occupy = None
    if args.occupy:
        parts = [int(v) for v in args.occupy.split(",")]
        occupy = {"cus": parts[0], "us": parts[1], "every_blocks": parts[2] if len(parts) > 2 else 3, "launches": 0}
        st_ = (student.module if dp else student).store()
        side_ = torch.cuda.Stream()
        chained, seen = st_.ready_cb, [0]

        def ready(names):                       # called by the backward pass each time a block's gradients are final
            seen[0] += 1
            if seen[0] % occupy["every_blocks"] == 0:
                side_.wait_stream(torch.cuda.current_stream())   # (an all-reduce starts behind the bucket's producers)
                ops.occupy_cus(occupy["cus"], occupy["us"], stream=side_)
                occupy["launches"] += 1
            if chained is not None:
                chained(names)
        st_.ready_cb = ready

    def step(it):
        if args.rehearse_on_one_gpu:
            print(f"[rank {rank}] step {it}", file=sys.stderr, flush=True)
        batch = crops if feed is None else next(feed)[0]
        loss, _ = train_step(student, teacher, loss_fn, opt, batch, it, lr, wd, mom, 3.0, freeze)
        return loss

    def fence():
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        loss = step(it)
        it += 1
    fence()
    if dp:                                  # event-time the stream-level waits on collectives from here on
        student.reducer.timing = True
        loss_fn.timing = True
        student.reducer.pop_wait_ms(), loss_fn.pop_wait_ms()
    timer = None
    if rank == 0 and not args.no_roofline and dtype == torch.bfloat16:
        timer = OpTimer(ops, dtype)
        timer.__enter__()
    wait0 = opt.host_wait_s
    t0 = time.perf_counter()
    import dino_amd.engine.vit as ev
    import dino_amd.utils.train_utils as tu
    two_streams, wg_stream = tu._TEACHER_STREAM, ev._WG_STREAM
    for k in range(args.steps):
        if timer is not None:
            timer.active = k % timer.every == 0
            # the bracketed steps run on ONE stream: with the teacher forward / the grouped weight
            # gradients on their side streams two GEMMs share the chip and an event pair (like
            # rocprofv3's kernel duration) would time the pair, not the kernel
            tu._TEACHER_STREAM = two_streams and not timer.active
            ev._WG_STREAM = wg_stream and not timer.active
        loss = step(it)
        it += 1
    tu._TEACHER_STREAM, ev._WG_STREAM = two_streams, wg_stream
    host_dt = time.perf_counter() - t0        # the host has issued every launch of the timed region
    host_wait = opt.host_wait_s - wait0  # of which: throttled behind the device (3-deep table ring)
    fence()
    dt = time.perf_counter() - t0
    if timer is not None:
        timer.__exit__()
    last_loss = loss.item()
    dp_wait = None
    if dp:
        # exposed communication: time the compute stream sat in work.wait() -- gradient buckets that backward
        # did not hide (FlatGradReducer.finish) and the centre's all-reduce at its join (DINOLoss._flush_center)
        w = torch.tensor([student.reducer.pop_wait_ms(), loss_fn.pop_wait_ms()], dtype=torch.float64, device=device)
        if world > 1:
            distributed.all_reduce(w, op=torch.distributed.ReduceOp.MAX)
        dp_wait = (w / args.steps).tolist()
    h2d = None
    if world == 1 and feed is None and not args.no_h2d_leg:
        # SURVEY section 8(d)'s step has the host -> device copy of the batch inside.  Same step count
        # again with every step's 12 fp32 crops (591 MB at B=256) taken from pinned host memory through
        # dino_amd.data.CropPrefetcher (copy of batch i+2 on its own stream under step i+1).  Reported
        # beside `value`, which by the bench contract is the HBM-resident rate.
        from dino_amd.data import CropPrefetcher
        g = torch.Generator().manual_seed(99)
        host = [[torch.randn(B, 3, sz, sz, generator=g).pin_memory() for sz in [224] * 2 + [96] * NLOCAL]
                for _ in range(2)]
        nwarm = min(args.warmup, 5)
        n2 = args.steps + nwarm
        feed2 = iter(CropPrefetcher(((host[i % 2], None) for i in range(n2)), device))
        start = it
        for _ in range(nwarm):
            train_step(student, teacher, loss_fn, opt, next(feed2)[0], min(it, total - 1), lr, wd, mom, 3.0, freeze)
            it += 1
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            train_step(student, teacher, loss_fn, opt, next(feed2)[0], min(it, total - 1), lr, wd, mom, 3.0, freeze)
            it += 1
        fence()
        dt2 = time.perf_counter() - t1
        h2d = {"value": round(B * args.steps / dt2, 2), "unit": "images/s", "steps": args.steps,
               "ms_per_step": round(1e3 * dt2 / args.steps, 3),
               "feed": "fp32 crops from pinned host memory every step (CropPrefetcher, copy stream)"}
        del host, feed2
    whole = None
    if world == 1 and feed is None and not args.no_whole_block_leg and ev._CLS_LAST:
        # For the record beside `value`: the same step with the WHOLE last transformer block computed, as the
        # reference does (it then keeps x[:, 0] only, vision_transformer.py:382-383; the engine's default runs that
        # block on its CLS rows -- same loss, same gradients, tests/test_gpu_model.py).  A short leg: 3 + 10 steps.
        ev._CLS_LAST = False
        try:
            for _ in range(3):
                train_step(student, teacher, loss_fn, opt, crops, min(it, total - 1), lr, wd, mom, 3.0, freeze)
                it += 1
            fence()
            t2 = time.perf_counter()
            for _ in range(10):
                train_step(student, teacher, loss_fn, opt, crops, min(it, total - 1), lr, wd, mom, 3.0, freeze)
                it += 1
            fence()
            dt3 = time.perf_counter() - t2
        finally:
            ev._CLS_LAST = True
        whole = {"value": round(B * 10 / dt3, 2), "unit": "images/s", "steps": 10,
                 "ms_per_step": round(1e3 * dt3 / 10, 3)}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    if rank == 0:
        ips = B * world * args.steps / dt
        line = {
            "metric": "train images/sec/node, ViT-S/16 2x224+10x96 DINO step",
            "value": round(ips, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic" + (" (REHEARSAL: ranks share one GPU)"
                                                        if args.rehearse_on_one_gpu else "")
            + ("" if feed is None else
               " (raw uint8 images from pinned host memory, multi-crop augmentation on the device every step)"
               if args.feed == "aug" else f" (crops from pinned host memory every step, {args.feed})"),
            "config": {"workload": "ViT-S/16 DINO multi-crop 2x224+10x96, out_dim 65536, "
                                   f"batch {B}/GPU, drop_path 0.1, clip 3.0, AdamW+EMA teacher",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "model_tflops_per_s": round(ips * GFLOP_PER_IMAGE / 1e3, 1)},
            "final_loss": round(last_loss, 5),
            "peak_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
            "prototype_layer_fused_with_loss": bool(getattr(student.module if dp else student, "fused_head_loss",
                                                            os.environ.get("DINO_PROTO_FUSED", "0") == "1")),
            # host time to issue a step's launches: the GPU is the bottleneck while this stays below ms_per_step
            "host_issue_ms_per_step": round(1e3 * (host_dt - host_wait) / args.steps, 3),
            "host_throttled_ms_per_step": round(1e3 * host_wait / args.steps, 3),
            "gemm_work_queue": ops.gemm_work_queue(),
            # the last transformer block runs on its CLS rows (all that VisionTransformer.forward returns, :382-383):
            # same loss and gradients, ~5 % less executed work than GFLOP_PER_IMAGE (the reference's count) says
            "last_block_cls_rows_only": os.environ.get("DINO_CLS_LAST", "1") != "0",
            "with_whole_last_block": whole,
        }
        if occupy is not None:
            line["occupied_cus"] = {k: occupy[k] for k in ("cus", "us", "every_blocks")}
            line["occupied_cus"]["launches_per_step"] = round(occupy["launches"] / max(1, it), 2)
        if dp:
            line["rccl_ranks"] = {"world_size": torch.distributed.get_world_size(),
                                  "backend": torch.distributed.get_backend(),
                                  "grad_buckets": len(student.reducer.buckets),
                                  "grad_bucket_MB": [round(b / 1e6, 1) for b in student.reducer.bucket_bytes()],
                                  "DINO_CU_MARGIN": os.environ.get("DINO_CU_MARGIN", "0"),
                                  "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS")}
            # max over ranks, per step: [gradient buckets, centre]; 0 = fully hidden behind backward / the step
            line["dp_wait_ms_per_step"] = round(sum(dp_wait), 4)
            line["dp_wait_ms_per_step_parts"] = {"grad_buckets": round(dp_wait[0], 4), "center": round(dp_wait[1], 4)}
        if timer is not None:
            roof, hbm, mfma = timer.result()
            line["roofline"] = roof
            # event-timed (launch gap included), ALGORITHMIC bytes of SURVEY section 8(d) / 8 TB/s
            line["hbm_kernels"] = hbm
            line["attention_kernels"] = mfma
        if h2d is not None:
            line["with_h2d_in_step"] = h2d
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            line["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if dp:
        distributed.disable()


if __name__ == "__main__":
    main()
