"""Data modules.  The reference's pipeline (data_loading/*: cv2 + albumentations over the xBD PNGs) is outside the hot
path and its dependencies are not installable here, so training/benchmark runs use a synthetic module that reproduces
the tensor CONTRACT of data_loading/pytorch_loader.py: ``{"image": f32 [B, 3|6, S, S] normalised with the ImageNet
mean/std of A.Normalize(), "mask": u8 [B, S, S]}`` (pre: {0,1}; post: {0..4}) with every sample containing
building pixels (CropNonEmptyMaskIfExists, pytorch_loader.py:57)."""
import os

import torch

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def synthetic_tile(task, size, generator):
    """uint8 HWC tile [S, S, 3|6] (what cv2.imread + np.concatenate produce, pytorch_loader.py:38,113) and its mask"""
    c = 3 if task == "pre" else 6
    img = torch.randint(0, 256, (size, size, c), generator=generator, dtype=torch.uint8)
    mask = torch.zeros(size, size, dtype=torch.uint8)
    hi = 2 if task == "pre" else 5
    for _ in range(12):
        h, w = [int(v) for v in torch.randint(max(2, size // 16), max(3, size // 5), (2,), generator=generator)]
        y0 = int(torch.randint(0, size - h, (1,), generator=generator))
        x0 = int(torch.randint(0, size - w, (1,), generator=generator))
        mask[y0:y0 + h, x0:x0 + w] = int(torch.randint(1, hi, (1,), generator=generator))
    return img, mask


def normalize_host(img_u8_hwc):
    """A.Normalize() + HWC->CHW on the host (pytorch_loader.py:90-91), albumentations' arithmetic: fp32 (v - mean*255) *
    (1 / (std*255)) - bit-identical to the device hand-over (xv2_normalize_u8_to_nhwc)"""
    c = img_u8_hwc.shape[-1]
    mean = torch.tensor(MEAN * (c // 3), dtype=torch.float32) * 255.0
    denom = torch.reciprocal(torch.tensor(STD * (c // 3), dtype=torch.float32) * 255.0)
    return ((img_u8_hwc.float() - mean) * denom).movedim(-1, -3).contiguous()


def synthetic_sample(task, size, generator):
    """normalised fp32 CHW image + mask: the reference loader's sample contract"""
    img, mask = synthetic_tile(task, size, generator)
    return normalize_host(img), mask


class SyntheticLoader:
    """device_u8 (default on a GPU unless XV2_HOST_NORMALIZE=1): batches carry the tiles as uint8 HWC on the device
    (ops.DeviceImage) and the network's first launch normalises them into NHWC - the input hand-over of SURVEY 8f row 4"""

    def __init__(self, task, batch_size, size, steps, seed, device, device_u8=None):
        self.task, self.bs, self.size, self.steps, self.seed, self.device = task, batch_size, size, steps, seed, device
        if device_u8 is None:
            device_u8 = str(device).startswith("cuda") and os.environ.get("XV2_HOST_NORMALIZE", "0") != "1"
        self.device_u8 = device_u8

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.steps):
            pairs = [synthetic_tile(self.task, self.size, g) for _ in range(self.bs)]
            u8 = torch.stack([p[0] for p in pairs])
            mask = torch.stack([p[1] for p in pairs]).to(self.device, non_blocking=True)
            if self.device_u8:
                from .ops import DeviceImage
                yield {"image": DeviceImage(u8.to(self.device, non_blocking=True)), "mask": mask}
            else:
                yield {"image": normalize_host(u8).to(self.device, non_blocking=True), "mask": mask}


class SyntheticDataModule:
    """Stand-in for data_loading/data_module.py:DataModule (train 512x512 crops, eval 1024x1024 tiles)."""

    def __init__(self, args, device="cuda", rank=0, train_size=512, eval_size=1024, steps_per_epoch=8, eval_steps=2):
        self.args, self.device, self.rank = args, device, rank
        self.train_size, self.eval_size = train_size, eval_size
        self.steps_per_epoch, self.eval_steps = steps_per_epoch, eval_steps

    def train_dataloader(self):
        return SyntheticLoader(self.args.type, self.args.batch_size, self.train_size, self.steps_per_epoch,
                               self.args.seed + 1000 * self.rank, self.device)

    def val_dataloader(self):
        return SyntheticLoader(self.args.type, self.args.val_batch_size, self.eval_size, self.eval_steps,
                               self.args.seed + 7 + 1000 * self.rank, self.device)

    def test_dataloader(self):
        return SyntheticLoader(self.args.type, self.args.val_batch_size, self.eval_size, self.eval_steps,
                               self.args.seed + 13 + 1000 * self.rank, self.device)
