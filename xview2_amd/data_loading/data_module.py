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


class DeviceAugLoader:
    """Training loader with the augmentation ON THE DEVICE (include/xv2.h xv2_augment_u8).

    Every tile is decoded once (PIL, a small thread pool running ahead of the consumer) and then lives in HBM as uint8
    (device_aug.DeviceTileCache: xBD's training set is 20 GB of 288); per sample the host only draws the decisions
    (device_aug.draw_params - the same stream, in the same order, as the worker path of pytorch_loader._TrainBase._augment,
    so both paths deliver the same bytes from the same seed) and ONE launch per batch crops, flips, adds the noise and applies
    the brightness / contrast tables to image and mask; the result goes to the network as an ops.DeviceImage.  RandomScale
    (p = 0.2, bicubic) stays on the host: the tile comes back from the cache, is zoomed and cropped there and travels as a
    one-off 512 x 512 source.  Sampling = DataLoader(shuffle=True, drop_last=True) / DistributedSampler semantics."""

    def __init__(self, dataset, batch_size, device, rank=0, world_size=1, seed=0, threads=4, shuffle=True):
        from .device_aug import DeviceAugmenter, DeviceTileCache
        self.ds, self.bs, self.device = dataset, int(batch_size), torch.device(device)
        self.rank, self.world, self.seed, self.epoch, self.shuffle = rank, world_size, seed, 0, shuffle
        self.cache = DeviceTileCache(self.device)
        self.aug = DeviceAugmenter(self.cache)
        self.rows, self.host_masks = {}, {}
        self.threads = max(1, int(threads))
        self._rng, self._rng_pinned, self._rng_epoch = None, False, None

    @property
    def rng(self):
        return self._rng

    @rng.setter
    def rng(self, g):                              # an assigned stream is kept as it is (replay tests)
        self._rng, self._rng_pinned = g, g is not None

    def __len__(self):
        return (len(self.ds) // self.world) // self.bs

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _order(self):
        import numpy as np
        n = len(self.ds)
        order = np.random.default_rng(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        per = n // self.world                      # (DistributedSampler pads; drop_last training can simply truncate)
        return [int(i) for i in order[self.rank:per * self.world:self.world]]

    def _row(self, i, pending):
        k = self.ds.key(i)
        if k not in self.rows:
            img, lbl = pending.pop(i).result() if i in pending else self.ds.load(i)
            self.rows[k] = self.cache.add(img, lbl)
            self.host_masks[k] = lbl
        return self.rows[k], self.host_masks[k]

    def __iter__(self):
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor
        from . import pytorch_loader as pl
        from .device_aug import draw_params
        from ..ops import DeviceImage
        if self.rng is None or self._rng_epoch != self.epoch:
            # a private stream per (seed, rank, epoch): follows --seed, differs between the ranks of a data-parallel job
            # and does not restart from the same state after a resume (the worker path seeds per worker and rank from
            # torch.initial_seed(), pytorch_loader.seed_worker); tests that replay the worker path assign self.rng.
            if not self._rng_pinned:
                self._rng = np.random.default_rng([int(self.seed), int(self.rank), int(self.epoch)])
            self._rng_epoch = self.epoch
        order = self._order()
        ahead = 4 * self.bs
        with ThreadPoolExecutor(self.threads) as pool:
            pending = {}

            def prefetch(pos):
                for j in order[pos:pos + ahead]:
                    if self.ds.key(j) not in self.rows and j not in pending:
                        pending[j] = pool.submit(self.ds.load, j)
            for b in range(len(order) // self.bs):
                prefetch(b * self.bs)
                plist, rows, extra = [], [], []
                # every cache row of the batch is resolved BEFORE a one-off tile gets its index: cache.add() of a later,
                # not yet cached sample would otherwise shift len(cache) under an earlier zoomed sample's index
                resolved = [self._row(i, pending) for i in order[b * self.bs:(b + 1) * self.bs]]
                base = len(self.cache)
                for row, mask in resolved:
                    s = pl.draw_scale(self.rng)
                    if s is None:
                        plist.append(draw_params(self.rng, mask, self.cache.imgs[row].shape[2] // 3))
                        rows.append(row)
                        continue
                    # zoomed sample: resize on the host, crop there, upload the window as a one-off source tile
                    img, zmask = pl.apply_scale(self.cache.imgs[row].cpu().numpy(), mask, s)
                    p = draw_params(self.rng, zmask, img.shape[2] // 3)
                    y0, x0 = p["y0"], p["x0"]
                    win = np.ascontiguousarray(img[y0:y0 + p["h"], x0:x0 + p["w"]])
                    wm = np.ascontiguousarray(zmask[y0:y0 + p["h"], x0:x0 + p["w"]])
                    p.update(H=p["h"], W=p["w"], y0=0, x0=0)
                    extra.append((torch.from_numpy(win).to(self.device), torch.from_numpy(wm).to(self.device)))
                    plist.append(p)
                    rows.append(base + len(extra) - 1)
                img, mask = self.aug(plist, rows, extra)
                yield {"image": DeviceImage(img), "mask": mask}


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
            # augmentation on the device, tiles resident in HBM (DeviceAugLoader); XV2_DEVICE_AUG=0, --autoaugment (a PIL policy)
            # or the host-normalise switch keep the reference's worker pipeline
            if (self.raw_u8 and os.environ.get("XV2_DEVICE_AUG", "1") != "0" and
                    not getattr(self.args, "autoaugment", False)):
                ds = fetch_pytorch_loader(self.train_path, self.args.type, True, {"batch_size": 1}, False, True).dataset
                self._train = DeviceAugLoader(ds, self.args.batch_size, self.device, self.rank, self.world_size,
                                              seed=getattr(self.args, "seed", 0) or 0, threads=max(2, self.args.num_workers))
            else:
                self._train = self._loader(self.train_path, True, self.train_loader_kwargs)
        return self._train

    def val_dataloader(self):
        return self._loader(self.val_path, False, self.test_loader_kwargs)

    def test_dataloader(self):
        return self._loader(self.test_path, False, self.test_loader_kwargs)
