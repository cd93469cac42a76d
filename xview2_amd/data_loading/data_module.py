"""data_loading/data_module.py:DataModule with the same directory layout (<data>/train, <data>/test, <data>/holdout),
loader options (:15-28) and method names; batches are moved to the training device as they are yielded (pinned
host buffers, non-blocking copies) because the HIP path consumes device tensors."""
import os

import torch

from .pytorch_loader import fetch_pytorch_loader, seed_worker


class _OnDevice:
    def __init__(self, loader, device):
        self.loader, self.device = loader, device

    def __len__(self):
        return len(self.loader)

    def set_epoch(self, epoch):
        sampler = getattr(self.loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)

    def __iter__(self):
        for batch in self.loader:
            out = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            img = out.get("image")
            if torch.is_tensor(img) and img.dtype == torch.uint8 and img.is_cuda:
                # raw uint8 HWC tiles: normalised on the device straight into the stem's NHWC input
                from ..ops import DeviceImage
                out["image"] = DeviceImage(img)
            yield out


class DataModule:
    def __init__(self, args, device="cuda", rank=0, world_size=1):
        self.args, self.device, self.rank, self.world_size = args, device, rank, world_size
        self.train_path = os.path.join(args.data, "train")
        self.val_path = os.path.join(args.data, "test")
        self.test_path = os.path.join(args.data, "holdout")
        pin = str(device).startswith("cuda")
        # device-side input hand-over: workers deliver uint8 HWC tiles, the GPU normalises them into
        # NHWC (xv2_normalize_u8_to_nhwc).  XV2_HOST_NORMALIZE=1 restores the reference's host normalise + CHW transpose.
        self.raw_u8 = pin and os.environ.get("XV2_HOST_NORMALIZE", "0") != "1"
        self.train_loader_kwargs = {"batch_size": args.batch_size, "pin_memory": pin, "num_workers": args.num_workers,
                                    "drop_last": True, "shuffle": True}
        self.test_loader_kwargs = {"batch_size": args.val_batch_size, "pin_memory": pin,
                                   "num_workers": args.num_workers, "drop_last": False, "shuffle": False}

    def _loader(self, path, training, kwargs):
        kwargs = dict(kwargs)
        if self.world_size > 1:     # what PL's ddp accelerator adds: one shard of the dataset per rank
            from torch.utils.data.distributed import DistributedSampler
            # the dataset is built once and re-wrapped with the sampler (no second glob / index parse)
            probe = fetch_pytorch_loader(path, self.args.type, training, {"batch_size": 1},
                                         getattr(self.args, "autoaugment", False), self.raw_u8).dataset
            kwargs["sampler"] = DistributedSampler(probe, self.world_size, self.rank, shuffle=kwargs.pop("shuffle"))
            loader = torch.utils.data.DataLoader(probe, worker_init_fn=seed_worker, **kwargs)
        else:
            loader = fetch_pytorch_loader(path, self.args.type, training, kwargs,
                                          getattr(self.args, "autoaugment", False), self.raw_u8)
        return _OnDevice(loader, self.device)

    def train_dataloader(self):
        if getattr(self, "_train", None) is None:
            self._train = self._loader(self.train_path, True, self.train_loader_kwargs)
        return self._train

    def val_dataloader(self):
        return self._loader(self.val_path, False, self.test_loader_kwargs)

    def test_dataloader(self):
        return self._loader(self.test_path, False, self.test_loader_kwargs)
