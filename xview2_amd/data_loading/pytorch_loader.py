"""xBD tile loader with the contract of the reference's data_loading/pytorch_loader.py (same class and function
names, same sample dict, same augmentation recipe) on PIL + numpy - cv2 and albumentations are not in this image.

    sample = {"image": float32 [3|6, H, W] (A.Normalize() statistics), "mask": uint8 [H, W]}

What is reproduced exactly: file discovery (`load_data`, pytorch_loader.py:31-35), the channel order (cv2.imread
gives B,G,R and the reference normalises those with the R,G,B ImageNet statistics - pytorch_loader.py:38,63,90 - so
this loader flips PIL's RGB to BGR), the index file semantics (utils/index.csv: `idx` column for localization, the
union of rows flagged 1..4 for damage, pytorch_loader.py:65-66,104-110), the order and probabilities of the
augmentations (:57-63,:77-91) and the evaluation dataset (:151-171).  What can only match in distribution: the random
streams (albumentations draws from `random`, this module from one numpy Generator per worker).
`--autoaugment` selects data_loading/autoaugment.py's policy instead of flips/noise/brightness, as in the reference."""
import os
from glob import glob

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader, Dataset

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
DEFAULT_INDEX = os.environ.get("XV2_INDEX_CSV", "/workspace/xview2/utils/index.csv")   # pytorch_loader.py:64


def seed_worker(worker_id):  # pytorch_loader.py:16-19
    worker_seed = torch.initial_seed() % 2 ** 32
    np.random.seed(worker_seed)
    _rng_holder["rng"] = np.random.default_rng(worker_seed)


_rng_holder = {"rng": np.random.default_rng(0)}


def _rng():
    return _rng_holder["rng"]


def load_data(path, dtype):  # pytorch_loader.py:31-35
    imgs = sorted(glob(os.path.join(path, "images", "*%s*" % dtype)))
    lbls = sorted(glob(os.path.join(path, "targets", "*%s*" % dtype)))
    assert len(imgs) == len(lbls) and len(imgs) > 0
    return imgs, lbls


def load_pair(img, lbl):
    """cv2.imread semantics: 8-bit B,G,R image; label read unchanged (single-channel uint8)"""
    im = np.asarray(Image.open(img).convert("RGB"))[:, :, ::-1]
    lb = np.asarray(Image.open(lbl))
    if lb.ndim == 3:
        lb = lb[:, :, 0]
    return np.ascontiguousarray(im), np.ascontiguousarray(lb.astype(np.uint8))


def read_index(path):
    """utils/index.csv: header idx,1,2,3,4 (utils/generate_idx.py:35-41) -> dict of int lists"""
    cols = None
    rows = []
    with open(path) as f:
        for line in f:
            parts = line.strip().split(",")
            if not parts or parts == [""]:
                continue
            if cols is None:
                cols = parts
                continue
            rows.append([int(float(v)) for v in parts])
    arr = np.array(rows, dtype=np.int64).reshape(-1, len(cols))
    return {c: arr[:, i].tolist() for i, c in enumerate(cols)}


def build_index(train_path, out_csv=None):
    """utils/generate_idx.py without its exclusion list: tiles with a >= 512 x 512 foreground overlap, flagged by
    the damage classes their post mask contains."""
    imgs_pre, _ = load_data(train_path, "pre")
    imgs_post, lbls_post = load_data(train_path, "post")
    rows = []
    for idx in range(len(imgs_post)):
        pre, _ = load_pair(imgs_pre[idx], lbls_post[idx])
        post, lbl = load_pair(imgs_post[idx], lbls_post[idx])
        ok = []
        for im in (pre, post):
            ys, xs = np.where(im.max(2) > 0)
            ok.append((ys.min(), ys.max(), xs.min(), xs.max()) if ys.size else None)
        if None in ok:
            continue
        h = min(ok[0][1], ok[1][1]) - max(ok[0][0], ok[1][0])
        w = min(ok[0][3], ok[1][3]) - max(ok[0][2], ok[1][2])
        if h < 512 or w < 512:
            continue
        present = set(np.unique(lbl).tolist())
        rows.append([idx] + [1 if c in present else 0 for c in (1, 2, 3, 4)])
    if out_csv:
        with open(out_csv, "w") as f:
            f.write("idx,1,2,3,4\n")
            for r in rows:
                f.write(",".join(str(v) for v in r) + "\n")
    arr = np.array(rows, dtype=np.int64).reshape(-1, 5)
    return {c: arr[:, i].tolist() for i, c in enumerate(("idx", "1", "2", "3", "4"))}


# ---- augmentations (albumentations 0.5 recipes named in pytorch_loader.py:57-63) ---------------------------------
def draw_scale(rng, p=0.2, scale_limit=(0.0, 0.3)):
    """the decision of A.RandomScale(p=0.2, scale_limit=(0, 0.3)): None, or the zoom factor"""
    if rng.random() >= p:
        return None
    return 1.0 + rng.uniform(*scale_limit)


def random_scale(img, mask, p=0.2, scale_limit=(0.0, 0.3)):
    """A.RandomScale(p=0.2, scale_limit=(0, 0.3), interpolation=cv2.INTER_CUBIC); masks use nearest"""
    s = draw_scale(_rng(), p, scale_limit)
    return (img, mask) if s is None else apply_scale(img, mask, s)


def apply_scale(img, mask, s):
    h, w = mask.shape[:2]
    nh, nw = int(round(h * s)), int(round(w * s))
    chans = [np.asarray(Image.fromarray(np.ascontiguousarray(img[:, :, i:i + 3])).resize((nw, nh), Image.BICUBIC))
             for i in range(0, img.shape[2], 3)]
    mask = np.asarray(Image.fromarray(mask).resize((nw, nh), Image.NEAREST))
    return np.concatenate(chans, 2), mask


# crop / flips / noise / brightness-contrast: decisions drawn by device_aug.draw_params (same order and probabilities as
# pytorch_loader.py:77-91), bytes moved by device_aug.apply_params_numpy here or by ONE launch on the GPU (xv2_augment_u8)
def crop_non_empty_mask_if_exists(img, mask, height=512, width=512):
    """A.CropNonEmptyMaskIfExists(p=1): a window around a random foreground pixel, else a random window"""
    H, W = mask.shape[:2]
    if H < height or W < width:
        raise ValueError("crop %dx%d larger than the tile %dx%d" % (height, width, H, W))
    ys, xs = np.nonzero(mask)
    if ys.size:
        k = int(_rng().integers(0, ys.size))
        y0 = int(np.clip(ys[k] - _rng().integers(0, height), 0, H - height))
        x0 = int(np.clip(xs[k] - _rng().integers(0, width), 0, W - width))
    else:
        y0 = int(_rng().integers(0, H - height + 1))
        x0 = int(_rng().integers(0, W - width + 1))
    return img[y0:y0 + height, x0:x0 + width], mask[y0:y0 + height, x0:x0 + width]


def normalize(img):
    """A.Normalize() with its defaults, in albumentations' own arithmetic (functional.normalize: fp32 mean * 255,
    fp32 reciprocal of std * 255, subtract, multiply), statistics applied in STORED channel order (B,G,R here).  The
    device hand-over (include/xv2.h xv2_normalize_u8_to_nhwc) computes the same two roundings bit for bit."""
    mean = MEAN * np.float32(255.0)
    denominator = np.reciprocal(STD * np.float32(255.0), dtype=np.float32)
    out = img.astype(np.float32)
    out -= mean
    out *= denominator
    return out


# RAW_U8 datasets skip normalise + HWC->CHW (pytorch_loader.py:90-91,145-147,166-170) and hand the uint8 HWC tile to the
# loader; the device normalises it into the stem's NHWC input (ops.DeviceImage).  Augmentations are unchanged: they all
# run on uint8 tiles BEFORE the normalisation in the reference too.
def _finish(img_u8, mask, raw_u8):
    if raw_u8:
        return {"image": np.ascontiguousarray(img_u8), "mask": np.ascontiguousarray(mask)}
    parts = [normalize(img_u8[:, :, i:i + 3]) for i in range(0, img_u8.shape[2], 3)]
    img = np.concatenate(parts, 2)
    return {"image": np.ascontiguousarray(np.transpose(img, (2, 0, 1))), "mask": np.ascontiguousarray(mask)}


class _TrainBase(Dataset):
    def __init__(self, autoaugment, raw_u8=False):
        self.use_autoaugment = bool(autoaugment)
        self.raw_u8 = bool(raw_u8)
        if self.use_autoaugment:
            from .autoaugment import ImageNetPolicy
            self.autoaugment = ImageNetPolicy()

    def _augment(self, img, mask):
        if self.use_autoaugment:        # pytorch_loader.py:75-84,:125-138: no zoom / flips / noise in this mode
            img, mask = crop_non_empty_mask_if_exists(img, mask)
            parts = [Image.fromarray(np.ascontiguousarray(img[:, :, i:i + 3])) for i in range(0, img.shape[2], 3)]
            out = self.autoaugment(parts[0], Image.fromarray(np.ascontiguousarray(mask)), *parts[1:])
            mask = np.asarray(out[1])
            img = np.concatenate([np.asarray(p) for p in (out[0],) + tuple(out[2:])], 2)
            return _finish(img, mask, self.raw_u8)
        from .device_aug import apply_params_numpy, draw_params
        img, mask = random_scale(img, mask)
        prm = draw_params(_rng(), mask, img.shape[2] // 3)      # crop, hflip, vflip, noise per image, brightness per image
        img, mask = apply_params_numpy(img, mask, prm)          # (data_module.DeviceAugLoader: the same bytes in one GPU launch)
        return _finish(img, mask, self.raw_u8)


class TrainPreDataset(_TrainBase):  # pytorch_loader.py:53-94
    def __init__(self, path, _, autoaugment=False, index_csv=None, raw_u8=False):
        super().__init__(autoaugment, raw_u8)
        self.imgs_pre, self.lbls_pre = load_data(path, "pre")
        self.idx = _index(path, index_csv)["idx"]

    def __len__(self):
        return len(self.idx)

    def key(self, i):
        return self.idx[i]

    def load(self, i):
        """the decoded, un-augmented tile of sample i: (uint8 [H, W, 3] in B,G,R order, uint8 mask)"""
        return load_pair(self.imgs_pre[self.idx[i]], self.lbls_pre[self.idx[i]])

    def __getitem__(self, i):
        return self._augment(*self.load(i))


class TrainPostDataset(_TrainBase):  # pytorch_loader.py:97-148
    def __init__(self, path, _, autoaugment=False, index_csv=None, raw_u8=False):
        super().__init__(autoaugment, raw_u8)
        self.imgs_pre, self.lbls_pre = load_data(path, "pre")
        self.imgs_post, self.lbls_post = load_data(path, "post")
        assert len(self.imgs_pre) == len(self.imgs_post)
        assert len(self.imgs_post) == len(self.lbls_post)
        ix = _index(path, index_csv)
        keep = set()
        for c in ("1", "2", "3", "4"):
            keep.update(i for i, f in zip(ix["idx"], ix[c]) if f == 1)
        self.idx = sorted(keep)

    def __len__(self):
        return len(self.idx)

    def key(self, i):
        return self.idx[i]

    def load(self, i):
        """the decoded, un-augmented pair of sample i: (uint8 [H, W, 6] = pre | post in B,G,R order, uint8 post mask)"""
        k = self.idx[i]
        img_pre, _ = load_pair(self.imgs_pre[k], self.lbls_pre[k])
        img_post, lbl = load_pair(self.imgs_post[k], self.lbls_post[k])
        return np.concatenate((img_pre, img_post), 2), lbl

    def __getitem__(self, i):
        return self._augment(*self.load(i))


class TestDataset(Dataset):  # pytorch_loader.py:151-171
    def __init__(self, path, mode, _=False, raw_u8=False):
        self.mode = mode
        self.raw_u8 = bool(raw_u8)
        self.imgs_pre, self.lbls_pre = load_data(path, "pre")
        self.imgs_post, self.lbls_post = load_data(path, "post")
        assert len(self.imgs_pre) == len(self.imgs_post)
        assert len(self.imgs_post) == len(self.lbls_post)

    def __len__(self):
        return len(self.imgs_pre)

    def __getitem__(self, i):
        img, lbl = load_pair(self.imgs_pre[i], self.lbls_pre[i])
        if self.mode == "post":
            img_post, lbl = load_pair(self.imgs_post[i], self.lbls_post[i])
            img = np.concatenate((img, img_post), 2)
        return _finish(img, lbl, self.raw_u8)


def _index(path, index_csv):
    csv = index_csv or DEFAULT_INDEX
    if os.path.exists(csv):
        return read_index(csv)
    local = os.path.join(path, "index.csv")
    if os.path.exists(local):
        return read_index(local)
    return build_index(path, local if os.access(path, os.W_OK) else None)


def fetch_pytorch_loader(path, mode, training, loader_kwargs, autoaugment=False, raw_u8=False):  # pytorch_loader.py:22-28
    if not training:
        dataset = TestDataset
    elif mode == "pre":
        dataset = TrainPreDataset
    else:
        dataset = TrainPostDataset
    return DataLoader(dataset(path, mode, autoaugment, raw_u8=raw_u8), worker_init_fn=seed_worker, **loader_kwargs)
