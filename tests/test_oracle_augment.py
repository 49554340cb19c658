"""The augmentation oracle (oracle/augment_oracle.py, a numpy restatement of the Pillow routines behind
dino/data/augmentations.py:75-157) against outputs of Pillow itself (tests/golden/augment.npz, written by
tools/make_golden_augment.py), bit for bit; plus the host half of dino_amd.data.augment (no GPU needed)."""
import os

import numpy as np
import pytest

from oracle import augment_oracle as A

GOLD = os.path.join(os.path.dirname(__file__), "golden", "augment.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def test_single_operations_match_pillow(g):
    img = g["src"][0]
    assert np.array_equal(A.to_gray(img), g["op/gray"])
    assert np.array_equal(A.rgb_to_hsv(img), g["op/hsv"])
    assert np.array_equal(A.hsv_to_rgb(g["src"][1]), g["op/hsv2rgb"])
    assert np.array_equal(A.solarize(img), g["op/solarize"])
    for i, f in enumerate(g["op/factors"]):
        assert np.array_equal(A.adjust_brightness(img, f), g[f"op/brightness{i}"]), f
        assert np.array_equal(A.adjust_contrast(img, f), g[f"op/contrast{i}"]), f
        assert np.array_equal(A.adjust_saturation(img, 0.5 + f / 2), g[f"op/saturation{i}"]), f
    for i, f in enumerate(g["op/hues"]):
        assert np.array_equal(A.adjust_hue(img, f), g[f"op/hue{i}"]), f


def test_gaussian_blur_matches_pillow(g):
    img = g["src"][0]
    for i, r in enumerate(g["op/radii"]):
        assert np.array_equal(A.gaussian_blur(img, float(r)), g[f"op/blur{i}"]), r


def test_bicubic_resized_crop_matches_pillow(g):
    img = g["src"][0]
    for i, (t, l, h, w, s) in enumerate(g["op/boxes"]):
        assert np.array_equal(A.resized_crop(img, t, l, h, w, s), g[f"op/resize{i}"]), (t, l, h, w, s)


def test_whole_crops_from_parameter_rows_match_pillow(g):
    S = int(g["S"])
    for i, row in enumerate(g["rows"]):
        assert np.array_equal(A.apply_params(g["src"], row, S), g["crops"][i]), i


def test_hsv_round_trip_matches_pillow_on_a_colour_cube():
    """Exhaustive over a 52^3 sub-cube against Pillow itself when it is importable (it is in this image; the
    full 256^3 cube was checked once with the same code: 0 mismatches either way)."""
    Image = pytest.importorskip("PIL.Image")
    v = np.arange(0, 256, 5)
    cube = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(-1, 1, 3).astype(np.uint8)
    assert np.array_equal(A.rgb_to_hsv(cube), np.asarray(Image.fromarray(cube).convert("HSV")))
    assert np.array_equal(A.hsv_to_rgb(cube), np.asarray(Image.fromarray(cube, "HSV").convert("RGB")))


# ------------------------------------------------------------------ host side of the product (CPU only)
def test_host_constants_agree_with_the_oracle():
    from dino_amd.data import augment as D
    for r in np.linspace(0.1, 2.0, 97):
        assert D.box_blur_constants(float(r)) == tuple(A.box_blur_params(A.gaussian_box_radius(float(r))))
    for f in np.linspace(-0.1, 0.1, 41):
        assert D.hue_shift(float(f)) == A.hue_shift_u8(float(f))


def test_sampled_parameters_follow_the_reference_distributions():
    """RandomResizedCrop.get_params' contract: boxes inside the image, area and aspect ratio in range;
    op slots a permutation of the four jitter ops or all empty; blur always on for global crop 1."""
    from dino_amd.data import augment as D
    aug = D.DeviceAugmentationDINO(224, 96, (0.32, 1.0), (0.05, 0.32), 10, solarization=True, seed=3)
    B, H, W = 64, 256, 256
    groups = aug.sample(B, H, W)
    assert [s for s, _ in groups] == [224, 96] and groups[0][1].shape == (2 * B, 24) and groups[1][1].shape == (10 * B, 24)
    for (S, P), scale in zip(groups, ((0.32, 1.0), (0.05, 0.32))):
        top, left, h, w = P[:, 1], P[:, 2], P[:, 3], P[:, 4]
        assert (top >= 0).all() and (left >= 0).all() and (top + h <= H).all() and (left + w <= W).all()
        area = h * w / float(H * W)
        assert (area > scale[0] * 0.9).all() and (area < scale[1] * 1.1).all()
        ratio = w / h
        assert (ratio > 0.74).all() and (ratio < 1.35).all()
        assert np.array_equal(P[:, 0], np.tile(np.arange(B), P.shape[0] // B))
        ops4 = np.sort(P[:, 6:10], axis=1)
        assert (((ops4 == [1, 2, 3, 4]).all(1)) | ((ops4 == 0).all(1))).all()
        assert 0.7 < (ops4[:, 0] > 0).mean() < 0.9
        assert set(np.unique(P[:, 10])) <= {0, 5} and set(np.unique(P[:, 5])) <= {0, 1}
    g = groups[0][1]
    assert g[:B, 18].all() and g[B:, 18].mean() < 0.3          # GaussianBlur(1.0) / GaussianBlur(0.1)
    assert (g[:B, 11] == 0).all() and set(np.unique(g[B:, 11])) <= {0, 6}
    assert 0.3 < groups[1][1][:, 18].mean() < 0.7
