"""Pin the CPU oracle (oracle/dino_oracle.py) against fixtures produced by the reference
itself (tools/make_golden.py).  Both are torch-CPU fp32, so agreement is ~1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import detfill
from oracle import dino_oracle as O
from tests.golden_spec import (TINY, FULL, MULTI, SCHED_CASES, POS_CASES, LOSS_CASE, digest,
                               tiny_crops, full_crops, multi_crops)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _cfgs(spec):
    v, h = spec["vit"], spec["head"]
    return O.VitCfg(**v), O.HeadCfg(in_dim=v["embed_dim"], **h)


def _trainer(spec, nsteps):
    vit, head = _cfgs(spec)
    sd = detfill.fill_state_dict_(O.new_state_dict(vit, head), tag=spec["tag"])
    B = spec["batch"]
    hp = O.StepHP(ncrops=len(spec["sizes"]), clip_grad=3.0,
                  freeze_last_layer_iter=spec["freeze_iter"])
    lr = O.cosine_scheduler(spec.get("lr", 5e-4 * B / 256.0), 1e-6, 1, nsteps)
    wd = O.cosine_scheduler(0.04, 0.4, 1, nsteps)
    mom = O.cosine_scheduler(0.996, 1, 1, nsteps)
    tt = O.teacher_temp_schedule(*spec.get("ttemp", (0.04, 0.04, 0)), nsteps)
    return O.OracleTrainer(vit, head, hp, sd, lr, wd, mom, tt)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_param_order_matches_reference(golden_dir):
    g = _load(golden_dir, "vits16_b8_steps.npz")
    vit, head = _cfgs(FULL)
    shapes = O.param_shapes(vit, head)
    assert list(shapes) == [str(s) for s in g["param_names"]]
    ops = _load(golden_dir, "ops.npz")
    assert [str(tuple(s)) for s in shapes.values()] == [str(s) for s in ops["groups/shapes"]]
    sd = {k: torch.zeros(s, requires_grad=True) for k, s in shapes.items()}
    reg, noreg = O.param_groups(sd.items())
    assert reg == [str(s) for s in ops["groups/reg"]]
    assert noreg == [str(s) for s in ops["groups/noreg"]]


def test_schedules(golden_dir):
    ops = _load(golden_dir, "ops.npz")
    for name, (b, f, n, nit, we, sv) in SCHED_CASES.items():
        mine = O.cosine_scheduler(b, f, n, nit, warmup_epochs=we, start_warmup_value=sv)
        np.testing.assert_array_equal(mine, ops[f"cos/{name}"])
    np.testing.assert_array_equal(O.teacher_temp_schedule(0.02, 0.07, 0.3, 50), ops["ttemp/warm"])


def test_pos_interpolation(golden_dir):
    ops = _load(golden_dir, "ops.npz")
    for name, (img, P, D, crop) in POS_CASES.items():
        n = int(img * 0.875 // P) ** 2
        pos = detfill.det_normal((1, n + 1, D), f"pos:{name}")
        out = O.interpolate_pos_encoding(pos, (crop // P) ** 2, crop, crop, P)
        np.testing.assert_allclose(out.numpy(), ops[f"pos/{name}"], rtol=0, atol=1e-6)


def test_loss_k65536(golden_dir):
    ops = _load(golden_dir, "ops.npz")
    K, B, nc = LOSS_CASE
    s = detfill.det_normal((nc * B, K), "loss:s", std=2.0).requires_grad_(True)
    t = detfill.det_normal((2 * B, K), "loss:t", std=2.0)
    c = detfill.det_normal((1, K), "loss:c", std=0.5)
    val = O.dino_loss(s, t, c, nc, 0.04)
    val.backward()
    assert abs(val.item() - float(ops["loss/value"])) < 1e-5 * abs(float(ops["loss/value"]))
    assert _rel(digest(s.grad), ops["loss/grad_digest"]) < 1e-5
    assert _rel(digest(O.update_center(c, t)), ops["loss/center_digest"]) < 1e-6


def test_tiny_three_steps(golden_dir):
    g = _load(golden_dir, "tiny_steps.npz")
    tr = _trainer(TINY, 3)
    crops = tiny_crops()
    names = list(tr.student)
    for it in range(3):
        trace = {} if it == 0 else None
        out = tr.step(crops, it, trace=trace)
        assert abs(out["loss"].item() - float(g[f"loss{it}"])) < 2e-6 * abs(float(g[f"loss{it}"]))
        np.testing.assert_allclose(tr.center.numpy(), g[f"center{it}"], rtol=0, atol=1e-6)
        gn = np.array([out["norms"][n] for n in names])
        assert _rel(gn, g[f"gnorms{it}"]) < 1e-4
        st = out["stats"]
        assert _rel([st["grad_norm"], st["grad_max"], st["clip_fraction"]], g[f"gstats{it}"]) < 1e-4
        sdg = np.stack([digest(tr.student[n]) for n in names])
        tdg = np.stack([digest(tr.teacher[n]) for n in names])
        assert _rel(sdg, g[f"student_digest{it}"]) < 1e-5
        assert _rel(tdg, g[f"teacher_digest{it}"]) < 1e-5
        if it == 0:
            assert _rel(trace["student_out"].detach().numpy(), g["student_out0"]) < 1e-5
            assert _rel(trace["teacher_out"].numpy(), g["teacher_out0"]) < 1e-5
            assert _rel(trace["student_out"].grad.numpy(), g["dstudent_out0"]) < 1e-5
            for pi in range(2):
                for li in range(TINY["vit"]["depth"]):
                    a = trace[f"pass{pi}"][f"block{li}"].detach().numpy()
                    assert _rel(a, g[f"act0_block{li}_pass{pi}"]) < 1e-5
            for n in names:
                assert _rel(trace["grads_raw"][n].numpy(), g[f"grad0/{n}"]) < 2e-4, n


def test_tiny_loss_curve_24_steps(golden_dir):
    """The oracle against the reference's own 24-step loss curve (tests/golden/tiny_curve.npz)."""
    from tests.golden_spec import CURVE
    g = _load(golden_dir, "tiny_curve.npz")
    n = CURVE["nsteps"]
    tr = _trainer(CURVE, n)
    crops = tiny_crops()
    names = list(tr.student)
    for it in range(n):
        out = tr.step(crops, it)
        ref = float(g[f"loss{it}"])
        assert abs(out["loss"].item() - ref) < 2e-5 * abs(ref), (it, out["loss"].item(), ref)
    assert _rel(digest(tr.center), g[f"center{n - 1}"]) < 1e-4
    assert _rel(np.stack([digest(tr.student[k]) for k in names]), g[f"student_digest{n - 1}"]) < 1e-4
    assert _rel(np.stack([digest(tr.teacher[k]) for k in names]), g[f"teacher_digest{n - 1}"]) < 1e-4


def test_vits16_config1_steps(golden_dir):
    """BASELINE.json configs[0]: ViT-S/16, 2x224, B=8, K=65536 -- the reference's
    CPU-runnable plumbing case, two optimizer steps (keeps the CPU suite short)."""
    g = _load(golden_dir, "vits16_b8_steps.npz")
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    tr = _trainer(FULL, 3)
    crops = full_crops()
    names = list(tr.student)
    for it in range(2):
        out = tr.step(crops, it)
        assert abs(out["loss"].item() - float(g[f"loss{it}"])) < 1e-5 * abs(float(g[f"loss{it}"]))
        assert _rel(digest(tr.center), g[f"center{it}"]) < 1e-5
        gn = np.array([out["norms"][n] for n in names])
        # same torch fp32 norm, same thread count -> same figure; on another host torch's fp32 sum over
        # the 16.7 M-element prototype gradient moves by ~2e-3 (see gnorms64_* in tools/make_golden.py)
        assert _rel(gn, g[f"gnorms{it}"]) < 2e-3
        assert _rel(np.stack([digest(tr.teacher[n]) for n in names]), g[f"teacher_digest{it}"]) < 1e-5


def _digest_close(a, b, tol):
    """digest = [sum, l2, 32 samples]: l2 relative, samples against the largest sample."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert abs(a[..., 1] - b[..., 1]).max() <= tol * (np.abs(b[..., 1]).max() + 1e-30)
    assert _rel(a[..., 2:], b[..., 2:]) < tol


def test_vits16_multicrop_config2_shape(golden_dir):
    """BASELINE.json configs[1]'s shape (ViT-S/16 full depth, 2x224 + 10x96, K=65536) at B=2:
    per-block activations of both student passes, logits, d loss / d logits, parameter gradients,
    loss and centre of step 0, and step 1's loss (after one optimiser + EMA update)."""
    g = _load(golden_dir, "vits16_multicrop_steps.npz")
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    tr = _trainer(MULTI, 3)
    crops = multi_crops()
    names = list(tr.student)
    assert names == [str(s) for s in g["param_names"]]
    for it in range(2):
        trace = {} if it == 0 else None
        out = tr.step(crops, it, trace=trace)
        assert abs(out["loss"].item() - float(g[f"loss{it}"])) < 1e-5 * abs(float(g[f"loss{it}"]))
        assert _rel(digest(tr.center), g[f"center{it}"]) < 1e-5
        if it == 0:
            for pi in range(2):
                for li in range(MULTI["vit"]["depth"]):
                    _digest_close(digest(trace[f"pass{pi}"][f"block{li}"]), g[f"actd0_block{li}_pass{pi}"], 1e-5)
            _digest_close(digest(trace["student_out"]), g["student_out_digest0"], 1e-5)
            _digest_close(digest(trace["teacher_out"]), g["teacher_out_digest0"], 1e-5)
            _digest_close(digest(trace["student_out"].grad), g["dstudent_out_digest0"], 1e-5)
            gd = np.stack([digest(trace["grads_raw"][n]) for n in names])
            _digest_close(gd, g["grad_digest0"], 2e-4)
        _digest_close(np.stack([digest(tr.teacher[n]) for n in names]), g[f"teacher_digest{it}"], 1e-5)


def _oracle_last_selfattention(x, sd, vit, pre="backbone."):
    """VisionTransformer.get_last_selfattention (vision_transformer.py:385-392) from the oracle's pieces."""
    t = O.prepare_tokens(x, sd, pre, vit)
    for i in range(vit.depth - 1):
        t = O.block(t, sd, f"{pre}blocks.{i}.", vit)
    b = f"{pre}blocks.{vit.depth - 1}."
    ln = O.layer_norm(t, sd[b + "norm1.weight"], sd[b + "norm1.bias"], vit.ln_eps)
    return O.attention(ln, sd, b + "attn.", vit.num_heads)[1]


def test_last_selfattention(golden_dir):
    ops = _load(golden_dir, "ops.npz")
    vit, head = _cfgs(TINY)
    sd = detfill.fill_state_dict_(O.new_state_dict(vit, head), tag=TINY["tag"])
    for size in (56, 24):
        x = detfill.det_normal((2, 3, size, size), f"lastattn:img{size}")
        with torch.no_grad():
            got = _oracle_last_selfattention(x, sd, vit)
        np.testing.assert_allclose(got.numpy(), ops[f"lastattn/{size}"], rtol=0, atol=1e-6)
