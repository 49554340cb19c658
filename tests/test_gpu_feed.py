"""Host -> HBM feed (the part in scope: pinned double-buffered H2D and
ToTensor+Normalize on the device).  The normalisation is bit-exact against the oracle's restatement
of torchvision's arithmetic (parity unpinned: see oracle/input_oracle.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _u8(shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).integers(0, 256, size=shape, dtype=np.uint8))


@pytest.mark.parametrize("shape", [(3, 96, 96, 3), (2, 224, 224, 3), (5, 7, 9, 3), (1, 1, 1, 3), (2, 6, 6, 1)])
def test_to_tensor_normalize_bit_exact(shape):
    from dino_amd import ops
    from dino_amd.data import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
    from oracle.input_oracle import to_tensor_normalize
    C = shape[-1]
    mean, std = IMAGENET_DEFAULT_MEAN[:C], IMAGENET_DEFAULT_STD[:C]
    u8 = _u8(shape, 1)
    want = to_tensor_normalize(u8, mean, std)
    got = ops.to_tensor_normalize(u8.cuda(), mean, std).cpu()
    assert got.shape == want.shape and torch.equal(got, want)
    # all 256 values of every channel
    ramp = torch.arange(256, dtype=torch.uint8).view(1, 16, 16, 1).repeat(1, 1, 1, C).contiguous()
    assert torch.equal(ops.to_tensor_normalize(ramp.cuda(), mean, std).cpu(),
                       to_tensor_normalize(ramp, mean, std))
    # CHW source layout gives the same crops
    chw = u8.permute(0, 3, 1, 2).contiguous()
    assert torch.equal(ops.to_tensor_normalize(chw.cuda(), mean, std, channels_last=False).cpu(), want)


def test_to_tensor_normalize_rejects_bad_arguments():
    from dino_amd import ops
    u8 = _u8((1, 4, 4, 3), 0).cuda()
    with pytest.raises(RuntimeError):
        ops.to_tensor_normalize(u8, (0.5, 0.5, 0.5), (0.5, 0.0, 0.5))      # std == 0
    with pytest.raises(Exception):
        ops.to_tensor_normalize(u8.float(), (0.5,) * 3, (0.5,) * 3)


def _batches(n, sizes, pinned, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        B = 4 if i < n - 1 else 3                                            # ragged last batch
        crops = [torch.randn(B, 3, s, s, generator=g) for s in sizes]
        if pinned:
            crops = [c.pin_memory() for c in crops]
        out.append((crops, torch.full((B,), i)))
    return out


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_prefetcher_yields_every_batch_in_order(pinned, depth):
    from dino_amd.data import CropPrefetcher
    src = _batches(7, (32, 32, 16), pinned)
    feed = CropPrefetcher(src, "cuda:0", depth=depth)
    assert len(feed) == 7
    seen = 0
    burn = torch.randn(2048, 2048, device="cuda")
    for (crops, labels), (want, wl) in zip(feed, src):
        # a backlog on the consumer's stream between taking a batch and reading it: the buffers
        # must not be overwritten by a later copy meanwhile
        for _ in range(4):
            burn = (burn @ burn).tanh_()
        sums = [c.double().sum() for c in crops]
        for c, w, s in zip(crops, want, sums):
            assert c.is_cuda and c.shape == w.shape
            assert torch.equal(c.cpu(), w)
            assert s.item() == pytest.approx(w.double().sum().item(), rel=1e-9, abs=1e-6)
        assert torch.equal(labels, wl)
        seen += 1
    assert seen == 7
    # a second epoch over the same object reuses the staging
    assert sum(1 for _ in feed) == 7


def test_prefetcher_uint8_crops_are_normalised_on_device():
    from dino_amd.data import CropPrefetcher, IMAGENET_DEFAULT_MEAN as M, IMAGENET_DEFAULT_STD as S
    from oracle.input_oracle import to_tensor_normalize
    src = [([_u8((4, 32, 32, 3), 10 * i), _u8((4, 16, 16, 3), 10 * i + 1)], None) for i in range(5)]
    for (crops, _), (raw, _) in zip(CropPrefetcher(src, "cuda:0", normalize=(M, S)), src):
        for c, r in zip(crops, raw):
            assert c.dtype == torch.float32 and torch.equal(c.cpu(), to_tensor_normalize(r, M, S))


def test_prefetcher_passes_device_tensors_and_single_tensors_through():
    from dino_amd.data import CropPrefetcher
    x = torch.randn(2, 3, 16, 16, device="cuda")
    (crops, lab), = list(CropPrefetcher([([x], 5)], "cuda:0"))
    assert crops[0] is x and lab == 5
    h = torch.randn(2, 3, 16, 16)
    (crop, _), = list(CropPrefetcher([(h, 0)], "cuda:0"))
    assert torch.is_tensor(crop) and torch.equal(crop.cpu(), h)


def test_train_one_epoch_from_host_batches_matches_device_batches():
    """train_one_epoch fed host crops (through the prefetcher it wraps around the loader) follows
    the same trajectory as one fed device crops."""
    from oracle import detfill
    from tests.golden_spec import TINY
    from tests.helpers_model import build_pair, schedules
    from tests.test_gpu_api import _trainer
    from dino_amd.utils.train_utils import train_one_epoch
    niter, B = 4, TINY["batch"]
    host = [(detfill.det_crops(B, TINY["sizes"], tag=f"feed:img{i}"), torch.zeros(B, dtype=torch.long))
            for i in range(niter)]
    lr, wd, mom, _ = schedules(B, niter)
    results = []
    for on_device in (True, False):
        s, t = build_pair(TINY, torch.float32)
        lf, op = _trainer(s, t, niter, len(host[0][0]))
        loader = [([c.cuda() for c in crops], lab) for crops, lab in host] if on_device else host
        stats = train_one_epoch(s, t, t, lf, loader, op, lr, wd, mom, 0, 1, None, 3.0,
                                TINY["freeze_iter"], 0, None, 0)
        results.append((stats["loss"], torch.cat([p.detach().flatten().cpu() for p in s.parameters()])))
    assert results[0][0] == pytest.approx(results[1][0], rel=1e-6)
    assert torch.allclose(results[0][1], results[1][1], rtol=1e-5, atol=1e-7)
