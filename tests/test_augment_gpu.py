"""Training augmentation on the device (include/xv2.h xv2_augment_u8): ONE launch crops, flips, adds the
counter-based Gaussian field and applies the brightness / contrast tables to image and mask - the bytes must equal the host
path of the loader (xview2_amd.data_loading.device_aug.apply_params_numpy = the numpy port of pytorch_loader.py:77-91), and
a whole epoch of the device loader must equal the dataset's own samples drawn from the same random stream."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tile(rng, H, W, C):
    img = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
    mask = np.zeros((H, W), np.uint8)
    for _ in range(6):
        y, x = int(rng.integers(0, H - 40)), int(rng.integers(0, W - 40))
        mask[y:y + 30, x:x + 25] = rng.integers(1, 5)
    return img, mask


@pytest.mark.parametrize("C", [3, 6])
def test_device_augmentation_equals_the_numpy_port_bit_for_bit(C):
    from xview2_amd.data_loading import device_aug as da
    rng = np.random.default_rng(11 + C)
    cache = da.DeviceTileCache(DEV)
    tiles = [_tile(rng, H, W, C) for H, W in ((1024, 1024), (640, 700), (512, 512), (1331, 1200))]
    for img, mask in tiles:
        cache.add(img, mask)
    aug = da.DeviceAugmenter(cache)
    drawn = np.random.default_rng(5)
    seen = {"hflip": 0, "vflip": 0, "noise": 0, "lut": 0}
    for batch in range(12):
        rows = [int(drawn.integers(0, len(tiles))) for _ in range(4)]
        plist = [da.draw_params(drawn, tiles[r][1], C // 3) for r in rows]
        if batch % 3 == 0:          # decisions with p = 0.1 / 0.2 are rare: force every branch through the kernel regularly
            plist[0]["noise"] = [(float(drawn.uniform(10, 50)) ** 0.5, int(drawn.integers(0, 2 ** 63))) for _ in range(C // 3)]
            plist[1]["lut"] = [np.clip(np.arange(256, dtype=np.float32) * 1.17 + -0.11 * 255.0, 0, 255).astype(np.uint8)
                               for _ in range(C // 3)]
            plist[2]["hflip"], plist[2]["vflip"] = True, True
            plist[3]["noise"] = plist[0]["noise"][::-1]
            plist[3]["lut"] = plist[1]["lut"]
        img, mask = aug(plist, rows)
        torch.cuda.synchronize()
        for i, (p, r) in enumerate(zip(plist, rows)):
            want_i, want_m = da.apply_params_numpy(tiles[r][0], tiles[r][1], p)
            assert np.array_equal(img[i].cpu().numpy(), want_i), (batch, i, p)
            assert np.array_equal(mask[i].cpu().numpy(), want_m), (batch, i)
            seen["hflip"] += p["hflip"]
            seen["vflip"] += p["vflip"]
            seen["noise"] += sum(q is not None for q in p["noise"])
            seen["lut"] += sum(q is not None for q in p["lut"])
    assert all(v > 0 for v in seen.values()), seen


def test_counter_based_gaussian_field_is_the_same_on_host_and_device():
    """noise only, mid-grey tile: (out - 128) IS the truncated field; compare a million values"""
    from xview2_amd.data_loading import device_aug as da
    cache = da.DeviceTileCache(DEV)
    cache.add(np.full((512, 512, 3), 128, np.uint8), np.zeros((512, 512), np.uint8))
    base = {"H": 512, "W": 512, "h": 512, "w": 512, "y0": 0, "x0": 0, "hflip": False, "vflip": False, "lut": [None]}
    for seed, sigma in ((1, 10.0 ** 0.5), (2 ** 62 + 12345, 50.0 ** 0.5), (987654321987, 5.0)):
        p = dict(base, noise=[(sigma, seed)])
        img, _ = da.DeviceAugmenter(cache)([p], [0])
        want, _ = da.apply_params_numpy(cache.imgs[0].cpu().numpy(), cache.masks[0].cpu().numpy(), p)
        got = img[0].cpu().numpy()
        assert np.array_equal(got, want)
        assert abs(float(got.astype(np.float32).std()) - sigma) < 0.1 * sigma


def test_device_loader_delivers_the_worker_paths_samples(tmp_path, monkeypatch):
    """DataModule's device loader (tiles resident in HBM, decisions on the host, one launch per batch) against the dataset's
    own __getitem__ driven by the same random stream in the same order - zoomed samples (host bicubic) included"""
    from tests.test_data_cpu import _tile_tree
    from xview2_amd.data_loading import data_module as dm, pytorch_loader as pl
    root = str(tmp_path / "xbd")
    os.makedirs(root)
    csv = _tile_tree(root, n=4, S=640)
    monkeypatch.setattr(pl, "DEFAULT_INDEX", csv)
    for mode, C in (("pre", 3), ("post", 6)):
        ds = pl.fetch_pytorch_loader(os.path.join(root, "train"), mode, True, {"batch_size": 1}, False, True).dataset
        loader = dm.DeviceAugLoader(ds, 2, DEV, seed=3, threads=2)
        zoomed = 0
        for epoch in range(6):
            loader.set_epoch(epoch)
            pl._rng_holder["rng"] = np.random.default_rng(100 + epoch)
            loader.rng = pl._rng()
            got = [(b["image"].u8.cpu().numpy(), b["mask"].cpu().numpy()) for b in loader]
            assert len(got) == len(loader) and got[0][0].shape == (2, 512, 512, C)
            pl._rng_holder["rng"] = np.random.default_rng(100 + epoch)      # replay: the worker path, same order, same stream
            order = loader._order()
            for b, (gi, gm) in enumerate(got):
                for j, i in enumerate(order[2 * b:2 * b + 2]):
                    probe = np.random.default_rng(0)
                    probe.bit_generator.state = pl._rng().bit_generator.state
                    zoomed += pl.draw_scale(probe) is not None
                    s = ds[i]
                    assert np.array_equal(gi[j], s["image"]) and np.array_equal(gm[j], s["mask"]), (mode, epoch, b, j)
        assert zoomed > 0, "no zoomed sample in 6 epochs: the one-off source path was not exercised"
        assert len(loader.cache) == len({ds.key(i) for i in range(len(ds))})      # every tile decoded and uploaded once
