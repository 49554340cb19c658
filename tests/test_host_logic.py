"""CPU tests of the host-side mirror of the reference interface (no GPU, no kernels):
schedules, parameter groups / names / order, config schema, crop grouping, flat layout,
pos-embed resize taps, and that modules refuse to compute off-GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import dino_oracle as O
from tests.golden_spec import FULL, SCHED_CASES, POS_CASES


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_schedules_match_reference(golden_dir):
    from dino_amd.utils import cosine_scheduler
    from dino_amd.components.loss import DINOLoss
    ops = _load(golden_dir, "ops.npz")
    for name, (b, f, n, nit, we, sv) in SCHED_CASES.items():
        got = cosine_scheduler(b, f, n, nit, warmup_epochs=we, start_warmup_value=sv)
        np.testing.assert_array_equal(got, ops[f"cos/{name}"])
    loss = DINOLoss(16, 4, 0.02, 0.07, 0.3, 50)
    np.testing.assert_array_equal(loss.teacher_temp_schedule, ops["ttemp/warm"])
    assert tuple(loss.center.shape) == (1, 16) and list(loss.state_dict()) == ["center"]


def test_param_names_order_groups_match_reference(golden_dir):
    from dino_amd.models import MultiCropWrapper, vision_transformer as vits
    from dino_amd.utils import get_params_groups
    g = _load(golden_dir, "vits16_b8_steps.npz")
    ops = _load(golden_dir, "ops.npz")
    m = MultiCropWrapper(vits.vit_small(patch_size=16), vits.DINOHead(384, 65536, norm_last_layer=False))
    names = [n for n, _ in m.named_parameters()]
    assert names == [str(s) for s in g["param_names"]]
    assert list(m.state_dict()) == names  # no buffers, same keys as the reference checkpoint
    assert [str(tuple(p.shape)) for _, p in m.named_parameters()] == [str(s) for s in ops["groups/shapes"]]
    ids = {id(p): n for n, p in m.named_parameters()}
    reg, noreg = get_params_groups(m)
    assert [ids[id(p)] for p in reg["params"]] == [str(s) for s in ops["groups/reg"]]
    assert [ids[id(p)] for p in noreg["params"]] == [str(s) for s in ops["groups/noreg"]]
    assert noreg["weight_decay"] == 0.0
    # norm_last_layer freezes weight_g (vision_transformer.py:707-708) and drops it from the groups
    m2 = MultiCropWrapper(vits.vit_small(patch_size=16), vits.DINOHead(384, 1024))
    assert not m2.head.last_layer.weight_g.requires_grad
    assert all(ids2 is not m2.head.last_layer.weight_g for grp in get_params_groups(m2) for ids2 in grp["params"])


def test_arch_constructors():
    from dino_amd.models import vision_transformer as vits
    for fn, D, L, H in ((vits.vit_tiny, 192, 12, 3), (vits.vit_small, 384, 12, 6),
                        (vits.vit_base, 768, 12, 12)):
        m = fn(img_size=256, patch_size=16)
        assert m.embed_dim == D and len(m.blocks) == L and m.num_heads == H
        assert tuple(m.pos_embed.shape) == (1, 197, D)
    m = vits.vit_small(img_size=256, patch_size=8, drop_path_rate=0.1)
    assert tuple(m.pos_embed.shape) == (1, 785, 384)
    assert abs(m.spec.drop_path[-1] - 0.1) < 1e-7 and m.spec.drop_path[0] == 0.0
    assert vits.__dict__["vit_small"] is vits.vit_small  # dino.py:143 looks archs up by name


def test_group_by_resolution():
    from dino_amd.models import group_by_resolution
    assert group_by_resolution([224, 224] + [96] * 10) == [(0, 2), (2, 12)]
    assert group_by_resolution([224, 224]) == [(0, 2)]
    assert group_by_resolution([96, 224, 224, 96]) == [(0, 1), (1, 3), (3, 4)]
    assert group_by_resolution([224]) == O.group_by_resolution([224])


def test_config_schema_and_overrides():
    from dino_amd.configs import default_config, get_cfg_from_file
    assert default_config.model.out_dim == 65536 and default_config.training.clip_grad == 3.0
    assert default_config.optim.wd == 1e-5 and isinstance(default_config.optim.wd, float)
    cfg = get_cfg_from_file(None, ["model.arch=vit_base", "training.batch_size_per_gpu=128"])
    assert cfg.model.arch == "vit_base" and cfg.wandb.tags == ["dino", "vit_base"]
    assert cfg.training.batch_size_per_gpu == 128 and cfg.speed.use_fp16 is True


def test_pos_taps_match_reference_resize(golden_dir):
    from dino_amd.utils.pos_interp import dense_matrix
    from oracle.detfill import det_normal
    ops = _load(golden_dir, "ops.npz")
    for name, (img, P, D, crop) in POS_CASES.items():
        side = int(img * 0.875 // P)
        dst = crop // P
        if dst == side:
            continue
        pos = det_normal((1, side * side + 1, D), f"pos:{name}")[0, 1:].double().numpy()
        got = dense_matrix(side, dst) @ pos
        assert np.abs(got - ops[f"pos/{name}"][0, 1:]).max() < 2e-5, name
        assert np.abs(dense_matrix(side, dst).sum(1) - 1).max() < 1e-5


def test_flat_layout_alignment():
    from dino_amd.engine.flat import FlatLayout
    from dino_amd._lib import DINO_CHUNK
    shapes = O.param_shapes(O.VitCfg(**FULL["vit"]), O.HeadCfg(in_dim=384, **FULL["head"]))
    lay = FlatLayout(shapes)
    assert all(off % 8 == 0 for off, _ in lay.slices.values())
    assert lay.total >= sum(int(np.prod(s)) for s in shapes.values())
    ch = lay.chunks.numpy()
    assert (ch[:, 2] <= DINO_CHUNK).all() and (ch[:, 1] % 8 == 0).all() and (ch[:, 2] % 8 == 0).all()
    # chunks tile each tensor exactly once, in order
    for t, (name, (off, n)) in enumerate(lay.slices.items()):
        mine = ch[ch[:, 0] == t]
        assert mine[0, 1] == off and mine[:, 2].sum() == (n + 7) // 8 * 8
    assert lay.tensor_chunk_begin.numpy()[-1] == len(ch)


def test_modules_fail_loudly_without_gpu():
    from dino_amd._lib import DinoHipError
    from dino_amd.models import MultiCropWrapper, vision_transformer as vits
    m = MultiCropWrapper(vits.vit_tiny(patch_size=16), vits.DINOHead(192, 64))
    with pytest.raises(DinoHipError):
        m(torch.zeros(1, 3, 32, 32))


def test_cancel_gradients_last_layer_and_helpers():
    from dino_amd.models import MultiCropWrapper, vision_transformer as vits
    from dino_amd.utils import cancel_gradients_last_layer, has_batchnorms
    m = MultiCropWrapper(vits.vit_tiny(patch_size=16), vits.DINOHead(192, 64, norm_last_layer=False))
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    cancel_gradients_last_layer(5, m, 5)
    assert all(p.grad is not None for p in m.parameters())
    cancel_gradients_last_layer(4, m, 5)
    for n, p in m.named_parameters():
        assert (p.grad is None) == ("last_layer" in n)
    assert not has_batchnorms(m)


def test_resume_from_checkpoint_contract(tmp_path):
    """Reference utils/utils.py:139-181: missing file -> 0; "module." prefixes dropped; modules load
    non-strictly, optimisers strictly; absent keys are skipped; returns the stored epoch."""
    import torch
    from dino_amd.utils import resume_from_checkpoint
    assert resume_from_checkpoint(str(tmp_path / "nope.pt"), model=torch.nn.Linear(2, 2)) == 0
    src = torch.nn.Linear(4, 3)
    opt_src = torch.optim.SGD(src.parameters(), lr=0.1, momentum=0.9)
    src(torch.ones(1, 4)).sum().backward()
    opt_src.step()
    wrapped = {"module." + k: v for k, v in src.state_dict().items()}     # as saved from a DP wrapper
    wrapped["module.extra"] = torch.zeros(1)                               # unknown key: non-strict
    path = str(tmp_path / "ck.pt")
    torch.save({"epoch": 7, "student": wrapped, "optimizer": opt_src.state_dict()}, path)
    dst = torch.nn.Linear(4, 3)
    opt_dst = torch.optim.SGD(dst.parameters(), lr=0.5, momentum=0.9)
    ep = resume_from_checkpoint(path, verbose=False, student=dst, optimizer=opt_dst, teacher=None,
                                dino_loss=torch.nn.Linear(1, 1))
    assert ep == 7
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    assert opt_dst.state_dict()["param_groups"][0]["lr"] == 0.1
    assert torch.equal(opt_dst.state_dict()["state"][0]["momentum_buffer"],
                       opt_src.state_dict()["state"][0]["momentum_buffer"])


def test_crop_prefetcher_on_a_host_device_is_a_pass_through_and_has_no_host_normalise():
    """CropPrefetcher moves nothing for a CPU device (batches come out as they went in) and refuses
    the uint8 -> normalised path there: that arithmetic exists only as a HIP kernel."""
    import pytest
    import torch
    from dino_amd.data import CropPrefetcher, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
    batches = [([torch.randn(2, 3, 8, 8), torch.randn(2, 3, 4, 4)], torch.tensor([i, i])) for i in range(3)]
    feed = CropPrefetcher(batches, "cpu")
    assert len(feed) == 3
    out = list(feed)
    assert all(o[0][0] is b[0][0] and o[1] is b[1] for o, b in zip(out, batches))
    with pytest.raises(RuntimeError):
        CropPrefetcher(batches, "cpu", normalize=(IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD))
    assert IMAGENET_DEFAULT_MEAN == (0.485, 0.456, 0.406) and IMAGENET_DEFAULT_STD == (0.229, 0.224, 0.225)


@pytest.mark.timeout(240)
def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment (the shape of the driver's N = 1 command) must start
    one rank per GPU itself, as the reference's main.py:28-44 does, and relay rank 0's JSON line -- not exit."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"launcher_selftest": True, "n_gpus": 2, "rank_sum": 1.0}
    # a world size that disagrees with --gpus is an error, not a silent N = 1 run
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launcher-selftest"],
                       cwd=root, env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1",
                                          MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       capture_output=True, text=True, timeout=100)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
