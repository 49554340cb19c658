"""Device-side multi-crop augmentation (dino_amd/csrc/augment.hip through the C ABI) against Pillow's own
outputs (tests/golden/augment.npz) and against the numpy oracle on larger random cases -- bit for bit."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "augment.npz")


def _crops_u8(aug_mod, src_dev, S, rows):
    """The uint8 crops (before ToTensor / Normalize) of DeviceAugmentationDINO.apply."""
    aug = aug_mod.DeviceAugmentationDINO(S, S, (0.3, 1.0), (0.05, 0.3), 0, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0))
    out = aug.apply(src_dev, S, rows)                       # fp32 [n,3,S,S] = u8 / 255
    return torch.round(out * 255.0).to(torch.uint8).permute(0, 2, 3, 1).contiguous().cpu().numpy()


def test_whole_crops_match_pillow_fixture():
    from dino_amd.data import augment as D
    g = np.load(GOLD)
    src = torch.from_numpy(g["src"]).cuda()
    got = _crops_u8(D, src, int(g["S"]), g["rows"].astype(np.int32))
    for i in range(got.shape[0]):
        assert np.array_equal(got[i], g["crops"][i]), (i, np.abs(got[i].astype(int) - g["crops"][i]).max())


def test_resized_crop_boxes_match_pillow_fixture():
    from dino_amd import ops
    g = np.load(GOLD)
    src = torch.from_numpy(g["src"]).cuda()
    for i, (t, l, h, w, s) in enumerate(g["op/boxes"]):
        row = np.zeros((1, 24), dtype=np.int32)
        row[0, :5] = (0, t, l, h, w)
        kmax = int(np.ceil(2.0 * max(max(h, w) / s, 1.0))) * 2 + 1
        out = ops.aug_resized_crop(src, torch.from_numpy(row).cuda(), int(s), kmax, int(h))
        assert np.array_equal(out[0].cpu().numpy(), g[f"op/resize{i}"]), (t, l, h, w, s)


def test_single_ops_match_pillow_fixture():
    """Every colour op and the blur on a full (non-square source -> square crop of the same data) image:
    the 72 x 72 left part of fixture image 0, compared with the oracle (itself pinned on these fixtures) and,
    for the identity-size crop, with Pillow's stored outputs restricted to that window where the op is local."""
    from oracle import augment_oracle as A
    from dino_amd.data import augment as D
    g = np.load(GOLD)
    img = np.ascontiguousarray(g["src"][:, :, :72])          # [3,72,72,3]
    src = torch.from_numpy(img).cuda()

    def fb(x):
        return int(np.array([x], dtype=np.float32).view(np.int32)[0])
    rows = []
    for op, vals in ((1, g["op/factors"]), (2, g["op/factors"]), (3, g["op/factors"]), (4, g["op/hues"])):
        for v in vals:
            r = np.zeros(24, dtype=np.int32)
            r[:5] = (0, 0, 0, 72, 72)
            r[6] = op
            r[12] = D.hue_shift(float(v)) if op == 4 else fb(0.5 + v / 2 if op == 3 else v)
            rows.append(r)
    for rad in g["op/radii"]:
        r = np.zeros(24, dtype=np.int32)
        r[:5] = (1, 0, 0, 72, 72)
        r[18] = 1
        r[19], r[20], r[21] = D.box_blur_constants(float(rad))
        rows.append(r)
    for extra in ((10, 5), (11, 6)):
        r = np.zeros(24, dtype=np.int32)
        r[:5] = (2, 0, 0, 72, 72)
        r[extra[0]] = extra[1]
        rows.append(r)
    rows = np.stack(rows)
    got = _crops_u8(D, src, 72, rows)
    for i, r in enumerate(rows):
        assert np.array_equal(got[i], A.apply_params(img, r, 72)), (i, r[6:12], r[18:22])
    # point-wise ops commute with taking the window: Pillow's own outputs
    assert np.array_equal(got[0], g["op/brightness0"][:, :72])
    assert np.array_equal(got[14], g["op/saturation4"][:, :72])
    assert np.array_equal(got[15], g["op/hue0"][:, :72])


@pytest.mark.parametrize("S,scale", [(224, (0.32, 1.0)), (96, (0.05, 0.32))])
def test_random_parameter_rows_match_oracle_at_bench_sizes(S, scale):
    """256 x 256 sources, the reference's crop sizes and scale ranges, parameters drawn by the product's own
    sampler (every op, random order, blur, solarize): each crop equals the oracle's, bit for bit."""
    from oracle import augment_oracle as A
    from dino_amd.data import augment as D
    rng = np.random.default_rng(11)
    B = 6
    low = rng.integers(0, 256, (B, 34, 34, 3)).astype(np.float32)
    up = torch.nn.functional.interpolate(torch.from_numpy(low).permute(0, 3, 1, 2), size=(256, 256), mode="bilinear")
    src = np.clip(up.permute(0, 2, 3, 1).numpy() + rng.integers(-30, 31, (B, 256, 256, 3)), 0, 255).astype(np.uint8)
    aug = D.DeviceAugmentationDINO(S, S, scale, scale, 0, solarization=True, seed=5)
    rows = np.concatenate([aug._sample_kind(B, 256, 256, scale, 0.6, 0.3) for _ in range(3)])
    rows[0, 1:5] = (0, 0, 256, 256)       # the whole image
    rows[1, 1:5] = (16, 16, S, S)         # a box of exactly the output size (both passes are identities)
    got = _crops_u8(D, torch.from_numpy(src).cuda(), S, rows)
    for i, r in enumerate(rows):
        want = A.apply_params(src, r, S)
        assert np.array_equal(got[i], want), (i, r, np.abs(got[i].astype(int) - want).max())


def test_device_augmentation_returns_the_reference_crop_list():
    """PatchDataAugmentationDINO.__call__'s contract (augmentations.py:151-157): [global1, global2, local x n],
    normalised fp32 [B,3,S,S]; and the normalisation is ToTensor + Normalize of the uint8 crop."""
    from oracle import augment_oracle as A
    from oracle.input_oracle import to_tensor_normalize
    from dino_amd.data import augment as D
    from dino_amd.data import IMAGENET_DEFAULT_MEAN as M, IMAGENET_DEFAULT_STD as SD
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (4, 128, 160, 3)).astype(np.uint8)
    aug = D.DeviceAugmentationDINO(64, 32, (0.32, 1.0), (0.05, 0.32), 3, solarization=True, seed=9)
    crops = aug(torch.from_numpy(src).cuda())
    assert [tuple(c.shape) for c in crops] == [(4, 3, 64, 64)] * 2 + [(4, 3, 32, 32)] * 3
    assert all(c.dtype == torch.float32 and c.is_cuda for c in crops)
    # same draws again through the oracle
    aug2 = D.DeviceAugmentationDINO(64, 32, (0.32, 1.0), (0.05, 0.32), 3, solarization=True, seed=9)
    groups = aug2.sample(4, 128, 160)
    k = 0
    for S, P in groups:
        for j in range(P.shape[0] // 4):
            want = np.stack([A.apply_params(src, P[j * 4 + b], S) for b in range(4)])
            ref = to_tensor_normalize(torch.from_numpy(want), M, SD)
            assert torch.equal(crops[k].cpu(), ref), (S, j)
            k += 1


def test_augmenting_loader_yields_what_direct_calls_give():
    """Three host batches through DeviceAugmentingLoader (side stream, two ahead) == the same augmentation
    object called batch by batch with the same seed."""
    from dino_amd.data import DeviceAugmentationDINO, DeviceAugmentingLoader
    rng = np.random.default_rng(4)
    batches = [(torch.from_numpy(rng.integers(0, 256, (5, 96, 112, 3)).astype(np.uint8)), torch.arange(5) + 10 * i)
               for i in range(3)]
    mk = lambda: DeviceAugmentationDINO(48, 24, (0.32, 1.0), (0.05, 0.32), 4, solarization=True, seed=21)  # noqa: E731
    direct = mk()
    want = [[c.clone() for c in direct(b.cuda())] for b, _ in batches]
    got = []
    for crops, labels in DeviceAugmentingLoader(batches, mk(), "cuda"):
        s = sum(float(c.sum()) for c in crops)          # consumer work on the current stream
        assert np.isfinite(s)
        got.append(([c.clone() for c in crops], labels))
    assert len(got) == 3
    for i in range(3):
        assert torch.equal(got[i][1], batches[i][1])
        assert len(got[i][0]) == 6
        for a, b in zip(got[i][0], want[i]):
            assert torch.equal(a, b)


def test_augmenting_loader_with_device_resident_batches():
    """Batches that already live on the GPU (allocated on the consumer's stream) are read by the augmentation kernels
    on the loader's side stream: the loader marks them for that stream, so dropping the batch and allocating over it
    on the main stream while the side stream still reads must not change the crops."""
    from dino_amd.data import DeviceAugmentationDINO, DeviceAugmentingLoader
    rng = np.random.default_rng(6)
    host = [torch.from_numpy(rng.integers(0, 256, (6, 128, 128, 3)).astype(np.uint8)) for _ in range(4)]
    mk = lambda: DeviceAugmentationDINO(64, 32, (0.32, 1.0), (0.05, 0.32), 3, solarization=True, seed=33)  # noqa: E731
    direct = mk()
    want = [[c.clone() for c in direct(b.cuda())] for b in host]

    def gen():
        for i, b in enumerate(host):
            d = b.cuda()
            yield d, torch.full((6,), i)
            del d
            junk = torch.full((6, 128, 128, 3), 255, dtype=torch.uint8, device="cuda")   # may reuse the block
            junk.add_(1)
    got = [[c.clone() for c in crops] for crops, _ in DeviceAugmentingLoader(gen(), mk(), "cuda")]
    assert len(got) == 4
    for g_, w_ in zip(got, want):
        for a, b in zip(g_, w_):
            assert torch.equal(a, b)


def test_bad_arguments_are_rejected():
    from dino_amd import ops
    from dino_amd.data import augment as D
    src = torch.zeros((1, 32, 32, 3), dtype=torch.uint8, device="cuda")
    aug = D.DeviceAugmentationDINO(16, 16, (0.3, 1.0), (0.05, 0.3), 0)
    row = np.zeros((1, 24), dtype=np.int32)
    row[0, :5] = (0, 20, 0, 16, 16)           # box leaves the image
    with pytest.raises(ValueError):
        aug.apply(src, 16, row)
    with pytest.raises(Exception):
        ops.aug_resized_crop(src.cpu(), torch.from_numpy(row), 16, 5, 16)
    crops = torch.zeros((1, 16, 16, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        ops.aug_box_blur(crops, crops, torch.from_numpy(row).cuda(), 0)   # in place is not allowed
