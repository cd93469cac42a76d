"""Minimal trainer with the part of the PyTorch-Lightning 1.0 ``Trainer`` contract that the reference's main.py uses
(main.py:96-122): fit / test, one process per GPU with SyncBatchNorm when gpus > 1, per-step LR scheduler, validation
every epoch driving ``f1_score`` checkpointing (ModelCheckpoint(monitor="f1_score", mode="max", save_last=True)),
resume.  Checkpoints keep PL's layout ``{"state_dict": ..., "hyper_parameters": {"args": ...}, "epoch": ...}`` with the
reference's key names (``model.unet.enc_l1.0.weight`` ...), so files move between the two code bases.
``precision=16`` (the reference's fp16 AMP) selects XV2_MATH_BF16_STORE: bf16 activations in HBM, bf16 MFMA with fp32
accumulation, fp32 BatchNorm statistics and master weights."""
import gc
import os

import torch

from . import dist as xdist
from .optim import FlatAdamW


class Trainer:
    def __init__(self, gpus=1, precision=32, max_epochs=1, min_epochs=None, sync_batchnorm=False, accelerator=None,
                 default_root_dir=".", resume_from_checkpoint=None, checkpoint_callback=True, callbacks=None,
                 benchmark=True, deterministic=False, num_sanity_val_steps=0, logger=False, log_every=0):
        self.gpus, self.max_epochs, self.root = gpus, max_epochs, default_root_dir
        self.sync_batchnorm, self.resume = sync_batchnorm, resume_from_checkpoint
        self.checkpointing = bool(checkpoint_callback)
        self.log_every = log_every
        from . import ops
        # the reference's --precision 16 is fp16 autocast (activations and convolutions in half precision, fp32
        # accumulate, fp32 BN statistics / master weights); here the 16-bit type is bf16 (no loss scaling needed)
        ops.MATH_MODE = ops.MATH_BF16 if precision == 16 else ops.fp32_math()
        ops.set_storage_dtype(torch.bfloat16 if precision == 16 else None)
        self.rank, self.local_rank, self.world = xdist.init_from_env()
        self.device = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.device)
        self.best_score = None
        self.global_step = 0

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, model, path, epoch, optimizer=None):
        if self.rank != 0:
            return
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        ckpt = {"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "hyper_parameters": {"args": model.args}, "epoch": epoch, "global_step": self.global_step}
        if isinstance(optimizer, FlatAdamW):
            ckpt["optimizer_states"] = [{k: (v.cpu() if torch.is_tensor(v) else v)
                                         for k, v in optimizer.state_dict().items()}]
        torch.save(ckpt, path)

    # ------------------------------------------------------------------ fit
    def fit(self, model, datamodule):
        model.to(self.device)
        start_epoch = 0
        # the loader (dataset index, worker pool, sampler) is built once; its length is this rank's steps per epoch
        train_loader = datamodule.train_dataloader()
        try:
            model.steps_per_epoch_hint = len(train_loader)
        except TypeError:
            model.steps_per_epoch_hint = None
        opt_cfg = model.configure_optimizers()
        if isinstance(opt_cfg, dict):
            optimizer, scheduler = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
        else:
            optimizer, scheduler = opt_cfg, None
        if self.resume and os.path.exists(self.resume):
            ckpt = torch.load(self.resume, map_location="cpu", weights_only=False)
            model.load_state_dict(ckpt["state_dict"])
            start_epoch = ckpt.get("epoch", -1) + 1
            self.global_step = ckpt.get("global_step", 0)
            if isinstance(optimizer, FlatAdamW) and ckpt.get("optimizer_states"):
                st = ckpt["optimizer_states"][0]
                optimizer.load_state_dict({k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in st.items()})
            if scheduler is not None:
                # NoamLR counts from 1 (its constructor steps once): after n optimizer steps it stands at n + 1
                scheduler.step(self.global_step + 1)
                if isinstance(optimizer, FlatAdamW):
                    optimizer.sync_lr()
        flat = isinstance(optimizer, FlatAdamW)
        reducer = xdist.GradReducer(optimizer, sync_bn=self.sync_batchnorm or self.world > 1) if flat else None
        frozen = False
        for epoch in range(start_epoch, self.max_epochs):
            model.current_epoch = epoch
            model.train()
            if hasattr(train_loader, "set_epoch"):
                train_loader.set_epoch(epoch)        # DistributedSampler reshuffles per epoch (PL's ddp does this)
            for i, batch in enumerate(train_loader):
                optimizer.zero_grad()
                if reducer:
                    reducer.prepare()
                loss = model.training_step(batch, i)
                loss.backward()
                if flat:
                    optimizer.step(reducer.finish())
                else:
                    optimizer.step()
                if scheduler is not None:
                    scheduler.step()
                self.global_step += 1
                if not frozen:
                    # everything alive after the first step (modules, parameters, packed layouts, the loader's index) is there for the
                    # whole run: park it in the permanent generation so that the cyclic collector's full passes walk only what the
                    # steps create - a full pass over the model's object graph is 10 - 30 ms, more than a bf16 step (bench.quiet_gc)
                    gc.collect()
                    gc.freeze()
                    frozen = True
                if self.log_every and self.rank == 0 and self.global_step % self.log_every == 0:
                    print("epoch %d step %d loss %.5f" % (epoch, self.global_step, float(loss)))
            xdist.check_peer_exchange()          # one-shot SyncBatchNorm exchange: a timed-out exchange is an error, per epoch
            score = self.validate(model, datamodule)
            if self.checkpointing:
                ckdir = os.path.join(self.root, "checkpoints")
                self.save_checkpoint(model, os.path.join(ckdir, "last.ckpt"), epoch, optimizer)
                if score is not None and (self.best_score is None or score >= self.best_score):
                    self.best_score = score
                    self.save_checkpoint(model, os.path.join(ckdir, "best.ckpt"), epoch, optimizer)
            if self.world > 1:
                # rank 0 alone wrote the checkpoints: the others wait here instead of inside the next epoch's first SyncBatchNorm
                # exchange (a collective would wait anyway; the one-shot peer exchange has a bounded wait - ADVICE r04)
                import torch.distributed as tdist
                tdist.barrier()
        if self.world > 1:
            xdist.reset_peer_exchange()          # unmap the peers' exchange buffers, free this rank's own
        if frozen:
            gc.unfreeze()
        return model

    @torch.no_grad()
    def validate(self, model, datamodule):
        model.eval()
        model.on_validation_epoch_start()
        outs = [model.validation_step(b, i) for i, b in enumerate(datamodule.val_dataloader())]
        model.validation_epoch_end(outs)
        score = getattr(model, "logged", {}).get("f1_score")
        return None if score is None else float(score)

    @torch.no_grad()
    def test(self, model, test_dataloaders=None):
        model.to(self.device).eval()
        model.on_test_epoch_start()
        for i, b in enumerate(test_dataloaders):
            model.test_step(b, i)
        model.test_epoch_end(None)
