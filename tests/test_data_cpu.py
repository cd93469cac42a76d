"""Input pipeline and launcher helpers: the PIL/numpy port of data_loading/pytorch_loader.py on
a generated miniature xBD tree, and the sysfs-based CPU affinity planner."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from xview2_amd.data_loading import pytorch_loader as pl
from xview2_amd.data_loading.data_module import DataModule
from xview2_amd.utils import gpu_affinity as ga

S = 600


def _tile_tree(root, n=4, S=S):
    rng = np.random.default_rng(0)
    for split in ("train", "test", "holdout"):
        for sub in ("images", "targets"):
            os.makedirs(os.path.join(root, split, sub), exist_ok=True)
        for i in range(n):
            for kind in ("pre", "post"):
                img = rng.integers(1, 256, (S, S, 3), dtype=np.uint8)
                Image.fromarray(img).save(os.path.join(root, split, "images", "t%02d_%s_disaster.png" % (i, kind)))
                m = np.zeros((S, S), np.uint8)
                if i != 1:       # tile 1 has an empty mask
                    y, x = 40 + 90 * i, 500 - 100 * i
                    m[y:y + 30, x:x + 30] = 1 if kind == "pre" else 1 + (i % 4)
                Image.fromarray(m).save(os.path.join(root, split, "targets", "t%02d_%s_disaster_target.png" % (i, kind)))
    with open(os.path.join(root, "index.csv"), "w") as f:
        f.write("idx,1,2,3,4\n0,1,0,0,0\n2,0,0,1,0\n3,0,0,0,0\n")
    return os.path.join(root, "index.csv")


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("xbd"))
    return root, _tile_tree(root)


def test_load_pair_is_bgr_like_cv2_and_eval_sample_is_normalised_exactly(tree):
    root, _ = tree
    ds = pl.TestDataset(os.path.join(root, "holdout"), "post")
    assert len(ds) == 4
    s = ds[2]
    rgb_pre = np.asarray(Image.open(ds.imgs_pre[2]).convert("RGB")).astype(np.float32)
    rgb_post = np.asarray(Image.open(ds.imgs_post[2]).convert("RGB")).astype(np.float32)
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    want = np.concatenate([(x[:, :, ::-1] / 255.0 - mean) / std for x in (rgb_pre, rgb_post)], 2).transpose(2, 0, 1)
    assert s["image"].dtype == np.float32 and s["image"].shape == (6, S, S)
    np.testing.assert_allclose(s["image"], want, rtol=0, atol=1e-6)
    # raw hand-over form of the same sample (device-side normalisation): the uint8 HWC tile in B,G,R order
    raw = pl.TestDataset(os.path.join(root, "holdout"), "post", raw_u8=True)[2]
    assert raw["image"].dtype == np.uint8 and raw["image"].shape == (S, S, 6)
    assert np.array_equal(raw["image"][:, :, :3], rgb_pre[:, :, ::-1].astype(np.uint8))
    assert np.array_equal(pl._finish(raw["image"], raw["mask"], False)["image"], s["image"])
    assert s["mask"].dtype == np.uint8 and set(np.unique(s["mask"]).tolist()) == {0, 3}   # the POST label (loader :166)
    assert pl.TestDataset(os.path.join(root, "holdout"), "pre")[2]["image"].shape == (3, S, S)


def test_normalize_is_albumentations_arithmetic_known_answers():
    """A.Normalize() (pytorch_loader.py:63) = albumentations functional.normalize: fp32 mean*255, fp32 reciprocal of
    std*255, one subtraction, one multiplication.  Hand-computed answers in exactly that arithmetic; the device kernel
    (xv2_normalize_u8_to_nhwc) is held to the same bits in tests/test_ops_gpu.py."""
    f = np.float32
    img = np.array([[[0, 128, 255]]], np.uint8)
    out = pl.normalize(img)
    assert out.dtype == np.float32
    for c, v in enumerate((0, 128, 255)):
        m255 = f(pl.MEAN[c]) * f(255.0)
        den = f(1.0) / (f(pl.STD[c]) * f(255.0))
        assert out[0, 0, c] == (f(v) - m255) * den
    # and it is the textbook (v/255 - mean)/std to fp32 rounding
    np.testing.assert_allclose(out[0, 0], (np.array([0, 128, 255], f) / 255.0 - pl.MEAN) / pl.STD, rtol=0, atol=5e-7)
    from xview2_amd import data as syn
    t = syn.normalize_host(torch.from_numpy(np.arange(6 * 4, dtype=np.uint8).reshape(2, 2, 6)))
    want = np.concatenate([pl.normalize(np.arange(24, dtype=np.uint8).reshape(2, 2, 6)[:, :, i:i + 3]) for i in (0, 3)], 2)
    assert t.shape == (6, 2, 2) and np.array_equal(t.numpy(), want.transpose(2, 0, 1))


def test_training_samples_follow_the_index_and_always_contain_buildings(tree):
    root, csv = tree
    pl._rng_holder["rng"] = np.random.default_rng(3)
    pre = pl.TrainPreDataset(os.path.join(root, "train"), "pre", False, csv)
    post = pl.TrainPostDataset(os.path.join(root, "train"), "post", False, csv)
    assert pre.idx == [0, 2, 3]            # the idx column (pytorch_loader.py:65-66)
    assert post.idx == [0, 2]              # rows with any damage class flagged (:104-110)
    for ds, c in ((pre, 3), (post, 6)):
        for rep in range(6):
            s = ds[rep % len(ds)]
            assert s["image"].shape == (c, 512, 512) and s["image"].dtype == np.float32
            assert s["mask"].shape == (512, 512) and s["mask"].dtype == np.uint8
            assert s["mask"].any(), "CropNonEmptyMaskIfExists must keep building pixels in the crop"


def test_geometric_augmentations_move_image_and_mask_together():
    from xview2_amd.data_loading import device_aug as da
    pl._rng_holder["rng"] = np.random.default_rng(0)
    img = np.zeros((64, 64, 3), np.uint8)
    mask = np.zeros((64, 64), np.uint8)
    img[5:9, 50:60] = 200
    mask[5:9, 50:60] = 1
    base = {"H": 64, "W": 64, "h": 64, "w": 64, "y0": 0, "x0": 0, "hflip": False, "vflip": False, "noise": [None], "lut": [None]}
    for key in ("hflip", "vflip"):
        a, b = da.apply_params_numpy(img, mask, dict(base, **{key: True}))
        assert np.array_equal(a[:, :, 0] > 0, b > 0)
        assert not np.array_equal(b, mask)
    big_i, big_m = pl.random_scale(np.tile(img, (10, 10, 1)), np.tile(mask, (10, 10)), p=1.0)
    assert big_i.shape[:2] == big_m.shape and 640 <= big_m.shape[0] <= 832
    flat = np.full((32, 32, 3), 128, np.uint8)
    noisy, _ = da.apply_params_numpy(flat, np.zeros((32, 32), np.uint8),
                                     dict(base, H=32, W=32, h=32, w=32, noise=[(30.0 ** 0.5, 12345)]))
    assert noisy.dtype == np.uint8 and 4.5 < noisy.astype(np.float32).std() < 6.5       # sigma = sqrt(30) = 5.48


def test_decisions_follow_the_reference_recipe():
    """device_aug.draw_params: probabilities and ranges of pytorch_loader.py:57-63 (0.33 / 0.33 flips, GaussNoise p = 0.1 with
    var in (10, 50), RandomBrightnessContrast p = 0.2 with limits 0.2), crop windows that contain a foreground pixel"""
    from xview2_amd.data_loading import device_aug as da
    rng = np.random.default_rng(3)
    mask = np.zeros((1024, 1024), np.uint8)
    mask[700:720, 100:130] = 1
    n = 4000
    ps = [da.draw_params(rng, mask, 2) for _ in range(n)]
    assert abs(sum(p["hflip"] for p in ps) / n - 0.33) < 0.03 and abs(sum(p["vflip"] for p in ps) / n - 0.33) < 0.03
    nz = [q for p in ps for q in p["noise"]]
    assert abs(sum(q is not None for q in nz) / len(nz) - 0.1) < 0.02
    assert all(10.0 <= q[0] ** 2 <= 50.0 for q in nz if q is not None)
    luts = [q for p in ps for q in p["lut"]]
    assert abs(sum(q is not None for q in luts) / len(luts) - 0.2) < 0.03
    for p in ps[:200]:
        assert 0 <= p["y0"] <= 512 and 0 <= p["x0"] <= 512
        assert mask[p["y0"]:p["y0"] + 512, p["x0"]:p["x0"] + 512].any()
    f = da.hash_normal_field(99, 1 << 20, 3.0)                 # the counter-based Gaussian field: moments of N(0, 9)
    assert abs(float(f.mean())) < 0.02 and abs(float(f.std()) - 3.0) < 0.02
    assert abs(float((np.abs(f) < 3.0).mean()) - 0.6827) < 0.003
    assert np.array_equal(f[:1000], da.hash_normal_field(99, 1000, 3.0))      # element i depends on (seed, i) only


def test_build_index_flags_classes_like_generate_idx(tree):
    root, _ = tree
    ix = pl.build_index(os.path.join(root, "train"))
    assert ix["idx"] == [0, 1, 2, 3]
    assert [ix[c][0] for c in "1234"] == [1, 0, 0, 0] and [ix[c][2] for c in "1234"] == [0, 0, 1, 0]
    assert [ix[c][1] for c in "1234"] == [0, 0, 0, 0]


class _Args:
    type = "post"
    batch_size = 2
    val_batch_size = 3
    num_workers = 0
    autoaugment = False


def test_data_module_batches_and_rank_sharding(tree, monkeypatch):
    root, csv = tree
    monkeypatch.setattr(pl, "DEFAULT_INDEX", csv)
    a = _Args()
    a.data = root
    dm = DataModule(a, device="cpu")
    b = next(iter(dm.train_dataloader()))
    assert b["image"].shape == (2, 6, 512, 512) and b["image"].dtype == torch.float32
    assert b["mask"].shape == (2, 512, 512) and b["mask"].dtype == torch.uint8
    sizes = [x["image"].shape[0] for x in dm.test_dataloader()]
    assert sizes == [3, 1]                                     # drop_last False, order kept (data_module.py:23-28)
    seen = []
    for r in range(2):
        dmr = DataModule(a, device="cpu", rank=r, world_size=2)
        seen.append(sum(x["image"].shape[0] for x in dmr.val_dataloader()))
    assert seen == [2, 2]                                      # 4 tiles split over 2 ranks


def test_affinity_planner_modes():
    assert ga.parse_cpulist("0-3,8-9\n") == [0, 1, 2, 3, 8, 9]
    fake = {0: [0, 1, 2, 3, 8, 9, 10, 11], 1: [0, 1, 2, 3, 8, 9, 10, 11],
            2: [4, 5, 6, 7, 12, 13, 14, 15], 3: [4, 5, 6, 7, 12, 13, 14, 15]}
    sib = [(i, i + 8) for i in range(8)]
    plan = lambda i, m: ga.plan_affinity(i, 4, m, lambda k: fake[k], sib)
    assert plan(2, "socket") == fake[2] and plan(2, "single") == [4]
    assert [plan(i, "single_unique") for i in range(4)] == [[0], [1], [4], [5]]
    assert [plan(i, "socket_unique_interleaved") for i in range(4)] == [[0, 2, 8, 10], [1, 3, 9, 11], [4, 6, 12, 14],
                                                                       [5, 7, 13, 15]]
    assert plan(3, "socket_unique_continuous") == [6, 7, 14, 15]
    with pytest.raises(RuntimeError):
        plan(0, "nonsense")
    assert set(ga.set_affinity(0, "socket")) <= set(range(os.cpu_count()))      # no GPU here: keeps the current set


def test_autoaugment_policy_keeps_geometry_of_image_and_mask_together(tree):
    from xview2_amd.data_loading import autoaugment as aa
    assert len(aa.POLICY) == 25 and aa.magnitude("rotate", 9) == 30 and aa.magnitude("posterize", 8) == 4
    assert abs(aa.magnitude("solarize", 5) - (256 - 5 * 256 / 9)) < 1e-9 and aa.magnitude("equalize", 3) == 0
    img = np.zeros((128, 128, 3), np.uint8)
    img[20:60, 70:110] = 255
    mask = (img[:, :, 0] > 0).astype(np.uint8)
    pim, pmask = Image.fromarray(img), Image.fromarray(mask)
    for op, idx in (("rotate", 9), ("shearX", 5), ("translateX", 3), ("shearY", 7)):
        for sign in (1, -1):
            a = np.asarray(aa.apply_op(pim, op, aa.magnitude(op, idx), sign))
            b = np.asarray(aa.apply_op(pmask, op, aa.magnitude(op, idx), sign))
            inter = ((a[:, :, 0] > 127) & (b > 0)).sum()
            assert inter >= 0.9 * (b > 0).sum() > 0, (op, sign)          # same warp for tile and mask
    for op in ("posterize", "solarize", "autocontrast", "equalize", "invert", "color", "contrast", "sharpness"):
        out = aa.apply_op(pim, op, aa.magnitude(op, 4), 1)
        assert out.size == pim.size and out.mode == "RGB"
    pol = aa.ImageNetPolicy(rng=np.random.default_rng(1))
    for _ in range(30):
        o = pol(pim, pmask, pim)
        assert len(o) == 3 and o[1].mode == pmask.mode and o[0].size == pim.size
    assert len(pol(pim, pmask)) == 2
    # and through the dataset: 6-channel sample, crop kept, mask still uint8
    root, csv = tree
    pl._rng_holder["rng"] = np.random.default_rng(5)
    ds = pl.TrainPostDataset(os.path.join(root, "train"), "post", True, csv)
    s = ds[0]
    assert s["image"].shape == (6, 512, 512) and s["mask"].shape == (512, 512) and s["mask"].dtype == np.uint8


def test_device_loader_indexes_one_off_tiles_after_the_whole_batch_is_cached(tree, monkeypatch):
    """a zoomed sample FOLLOWED by a not yet cached one in the same batch - the one-off 512 x 512 window
    must keep its index although the later sample grows the tile cache; and the loader's random stream follows
    (seed, rank, epoch) instead of the never-seeded module default."""
    from xview2_amd.data_loading import data_module as dm
    root, csv = tree
    monkeypatch.setattr(pl, "DEFAULT_INDEX", csv)
    ds = pl.fetch_pytorch_loader(os.path.join(root, "train"), "pre", True, {"batch_size": 1}, False, True).dataset
    seen = []

    def fake_aug(cache):
        def run(plist, rows, extra=()):
            for p, r in zip(plist, rows):
                src = cache.imgs[r] if r < len(cache) else extra[r - len(cache)][0]
                assert tuple(src.shape)[:2] == (p["H"], p["W"]), (r, len(cache), tuple(src.shape), p["H"], p["W"])
                seen.append(r >= len(cache))
            n = len(plist)
            return torch.zeros(n, 512, 512, 3, dtype=torch.uint8), torch.zeros(n, 512, 512, dtype=torch.uint8)
        return run

    class _Img:                                   # ops.DeviceImage needs the HIP library: not under test here
        def __init__(self, u8):
            self.u8 = u8
    import xview2_amd.ops as ops
    monkeypatch.setattr(ops, "DeviceImage", _Img)
    calls = {"n": 0}

    def scale_first_of_each_batch(rng, p=0.2, scale_limit=(0.0, 0.3)):
        calls["n"] += 1
        rng.random()
        return 1.0 + rng.uniform(*scale_limit) if calls["n"] % 2 == 1 else None
    monkeypatch.setattr(pl, "draw_scale", scale_first_of_each_batch)
    loader = dm.DeviceAugLoader(ds, 2, "cpu", seed=3, threads=1)
    loader.aug = fake_aug(loader.cache)
    batches = list(loader)                        # first epoch: every tile uncached when its batch starts
    assert len(batches) == len(loader) and seen[0::2] == [True] * len(batches) and not any(seen[1::2])
    # stream: a function of (seed, rank, epoch)
    def first_draw(seed, rank, epoch):
        ld = dm.DeviceAugLoader(ds, 1, "cpu", rank=rank, world_size=2 if rank else 1, seed=seed, threads=1)
        ld.aug = fake_aug(ld.cache)
        ld.set_epoch(epoch)
        calls["n"] = 0
        next(iter(ld))
        return ld.rng.bit_generator.state["state"]["state"]
    a = first_draw(3, 0, 0)
    assert a == first_draw(3, 0, 0) and a != first_draw(4, 0, 0) and a != first_draw(3, 0, 1) and a != first_draw(3, 1, 0)
