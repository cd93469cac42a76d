"""The training augmentations of data_loading/pytorch_loader.py:57-63,77-91 split into DECISIONS and BYTES.

`draw_params` draws every random decision of one sample in the order the reference's pipeline consumes them (crop around a
random foreground pixel, horizontal flip, vertical flip, per image: Gaussian-noise variance + field seed, brightness /
contrast) - a few hundred bytes.  `apply_params_numpy` turns (tile, mask, params) into the augmented uint8 tile on the
host; `DeviceAugmenter` does the same with ONE launch on the GPU (include/xv2.h xv2_augment_u8), bit for bit the same
bytes (tests/test_augment_gpu.py).  With `DeviceTileCache` the decoded tiles stay in HBM for the whole run (xBD's 2799
training pairs: 17.6 GB of images + 2.9 GB of masks of the 288 GB), so a training batch costs the host its parameters and no
PCIe traffic; RandomScale (p = 0.2, bicubic) stays on the host - a zoomed sample is cropped there and uploaded as a one-off
512 x 512 tile.

The Gaussian field is counter-based (splitmix64 of seed and element index -> Box-Muller in fp64 -> float32) so that host and
device generate the same field from the seed alone; everything else is integer / table arithmetic.  The only non-integer
operations, fp64 log / cos / sqrt, go through two math libraries (numpy's, ROCm's): a last-place difference there survives the
rounding to float32 with probability ~1e-8 per value and the truncation to uint8 far more rarely - the tests pin fixed seeds."""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
M1, M2 = np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)


def _splitmix64(z):
    z = (z ^ (z >> np.uint64(30))) * M1
    z = (z ^ (z >> np.uint64(27))) * M2
    return z ^ (z >> np.uint64(31))


def hash_normal_field(seed, n, sigma):
    """float32 [n]: element i = float32(sigma * sqrt(-2 ln u1) cos(2 pi u2)), (u1, u2) from splitmix64(seed, i) - the field of
    csrc/augment.hip hash_normal"""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z1 = _splitmix64(np.uint64(seed) + i * GOLDEN)
        z2 = _splitmix64(z1)
    u1 = ((z1 >> np.uint64(11)) + np.uint64(1)).astype(np.float64) * 2.0 ** -53
    u2 = (z2 >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586 * u2)
    return (float(np.float32(sigma)) * g).astype(np.float32)


def draw_params(rng, mask, parts, height=512, width=512):
    """the random decisions of one sample, in pipeline order (pytorch_loader.py:77-91): crop, hflip, vflip, then per image the
    noise, then per image brightness / contrast.  `mask`: the (possibly zoomed) uint8 mask, `parts`: 1 (pre) or 2 (pre | post)"""
    H, W = mask.shape[:2]
    if H < height or W < width:
        raise ValueError("crop %dx%d larger than the tile %dx%d" % (height, width, H, W))
    ys, xs = np.nonzero(mask)
    if ys.size:       # A.CropNonEmptyMaskIfExists: a window around a random foreground pixel
        k = int(rng.integers(0, ys.size))
        y0 = int(np.clip(ys[k] - rng.integers(0, height), 0, H - height))
        x0 = int(np.clip(xs[k] - rng.integers(0, width), 0, W - width))
    else:
        y0 = int(rng.integers(0, H - height + 1))
        x0 = int(rng.integers(0, W - width + 1))
    p = {"H": H, "W": W, "h": height, "w": width, "y0": y0, "x0": x0,
         "hflip": bool(rng.random() < 0.33), "vflip": bool(rng.random() < 0.33), "noise": [], "lut": []}
    for _ in range(parts):      # A.GaussNoise(p=0.1, var_limit=(10, 50)): one call per image
        if rng.random() < 0.1:
            # (sigma travels to the device as a float32: it IS a float32 on both sides)
            p["noise"].append((float(np.float32(float(rng.uniform(10.0, 50.0)) ** 0.5)), int(rng.integers(0, 2 ** 63))))
        else:
            p["noise"].append(None)
    for _ in range(parts):      # A.RandomBrightnessContrast(p=0.2, limit 0.2, brightness_by_max): a lookup table on uint8
        if rng.random() < 0.2:
            alpha = 1.0 + rng.uniform(-0.2, 0.2)
            beta = rng.uniform(-0.2, 0.2)
            p["lut"].append(np.clip(np.arange(256, dtype=np.float32) * alpha + beta * 255.0, 0, 255).astype(np.uint8))
        else:
            p["lut"].append(None)
    return p


def apply_params_numpy(img, mask, p):
    """(uint8 [H, W, 3 * parts], uint8 [H, W], params) -> augmented (uint8 [h, w, 3 * parts], uint8 [h, w])"""
    h, w = p["h"], p["w"]
    img = img[p["y0"]:p["y0"] + h, p["x0"]:p["x0"] + w]
    mask = mask[p["y0"]:p["y0"] + h, p["x0"]:p["x0"] + w]
    if p["hflip"]:
        img, mask = img[:, ::-1], mask[:, ::-1]
    if p["vflip"]:
        img, mask = img[::-1], mask[::-1]
    parts = []
    for k, (nz, lut) in enumerate(zip(p["noise"], p["lut"])):
        part = np.ascontiguousarray(img[:, :, 3 * k:3 * k + 3])
        if nz is not None:
            sigma, seed = nz
            noisy = part.astype(np.float32) + hash_normal_field(seed, part.size, sigma).reshape(part.shape)
            part = np.clip(noisy, 0, 255).astype(np.uint8)
        if lut is not None:
            part = lut[part]
        parts.append(part)
    return np.concatenate(parts, 2) if len(parts) > 1 else parts[0], np.ascontiguousarray(mask)


def pack_params(plist, src_rows):
    """-> (int32 [N, 16] table, uint8 [N, 2, 256] tables) for xv2_augment_u8; src_rows[i] = row of the pointer tables"""
    n = len(plist)
    tab = np.zeros((n, 16), dtype=np.int32)
    luts = np.zeros((n, 2, 256), dtype=np.uint8)
    f = tab.view(np.float32)
    u = tab.view(np.uint32)
    for i, (p, row) in enumerate(zip(plist, src_rows)):
        tab[i, 0:7] = (row, p["H"], p["W"], p["y0"], p["x0"], int(p["hflip"]), int(p["vflip"]))
        for k, nz in enumerate(p["noise"]):
            if nz is not None:
                tab[i, 7 + k] = 1
                f[i, 9 + k] = np.float32(nz[0])
                u[i, 11 + k] = nz[1] & 0xFFFFFFFF
                u[i, 13 + k] = nz[1] >> 32
        bits = 0
        for k, lut in enumerate(p["lut"]):
            if lut is not None:
                bits |= 1 << k
                luts[i, k] = lut
        tab[i, 15] = bits
    return tab, luts


class DeviceTileCache:
    """decoded uint8 tiles and masks resident in HBM: rows of two device pointer tables (xv2_augment_u8's src_img / src_mask)"""

    def __init__(self, device):
        import torch
        self.device = torch.device(device)
        self.imgs, self.masks = [], []
        self._ptrs = None

    def add(self, img_u8, mask_u8):
        import torch
        self.imgs.append(torch.from_numpy(np.ascontiguousarray(img_u8)).to(self.device))
        self.masks.append(torch.from_numpy(np.ascontiguousarray(mask_u8)).to(self.device))
        self._ptrs = None
        return len(self.imgs) - 1

    def __len__(self):
        return len(self.imgs)

    def nbytes(self):
        return sum(t.numel() for t in self.imgs) + sum(t.numel() for t in self.masks)

    def pointer_tables(self, extra=()):
        """(img pointers, mask pointers) as int64 device tensors; `extra` = [(img tensor, mask tensor)] one-off tiles appended
        behind the cached rows (zoomed samples)"""
        import torch
        if self._ptrs is None:
            self._ptrs = (torch.tensor([t.data_ptr() for t in self.imgs], dtype=torch.int64),
                          torch.tensor([t.data_ptr() for t in self.masks], dtype=torch.int64))
        pi, pm = self._ptrs
        if extra:
            pi = torch.cat([pi, torch.tensor([e[0].data_ptr() for e in extra], dtype=torch.int64)])
            pm = torch.cat([pm, torch.tensor([e[1].data_ptr() for e in extra], dtype=torch.int64)])
        return pi.to(self.device, non_blocking=True), pm.to(self.device, non_blocking=True)


class DeviceAugmenter:
    """(params of a batch, source rows) -> augmented uint8 batch [N, h, w, C] + masks [N, h, w] on the device: one launch"""

    def __init__(self, cache):
        self.cache = cache

    def __call__(self, plist, rows, extra=()):
        import torch
        from .._capi import call
        if not plist:
            raise ValueError("empty batch")
        h, w = plist[0]["h"], plist[0]["w"]
        C = 3 * len(plist[0]["noise"])
        if any((p["h"], p["w"], 3 * len(p["noise"])) != (h, w, C) for p in plist):
            raise ValueError("the samples of a batch must share crop size and channel count")
        for p, r in zip(plist, rows):
            src = self.cache.imgs[r] if r < len(self.cache) else extra[r - len(self.cache)][0]
            if tuple(src.shape) != (p["H"], p["W"], C) or src.dtype != torch.uint8 or not src.is_contiguous():
                raise ValueError("source tile %d is %s %s, the parameters were drawn for %s" % (
                    r, tuple(src.shape), src.dtype, (p["H"], p["W"], C)))
        dev = self.cache.device
        tab, luts = pack_params(plist, rows)
        pi, pm = self.cache.pointer_tables(extra)
        n = len(plist)
        img = torch.empty((n, h, w, C), dtype=torch.uint8, device=dev)
        mask = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
        call("xv2_augment_u8", torch.from_numpy(tab).to(dev), pi, pm, torch.from_numpy(luts).to(dev), n, C, h, w, img, mask)
        return img, mask
