"""``Model``: the training / evaluation module the reference defines in model/plt.py, re-hosted on the HIP path.

What is kept is the SURFACE a user of the reference touches - ``Model(args)``, the step / epoch hooks a
Lightning-style trainer calls, ``configure_optimizers``, ``add_model_specific_args`` (every flag, choice and default of
model/plt.py:181-234) and the artefacts written during evaluation (dllogger-style JSON lines, ``.npy`` probabilities
with ``_target.png`` label maps, model/plt.py:105-144).  The implementation underneath is this package's: the
networks and the loss run through the HIP C ABI, AdamW is the flat fused optimizer, F1 bookkeeping is one counting
launch per batch.  pytorch_lightning is not installed in this image; when it is importable ``Model`` derives from
``pl.LightningModule``, otherwise from a small stand-in driven by xview2_amd.trainer.
"""
import json
import os
from argparse import ArgumentParser

import numpy as np
import torch
from torch import nn

from . import criterion, networks, ops
from .optim import FlatAdamW
from .utils.f1 import F1
from .utils.scheduler import NoamLR

try:  # pragma: no cover - not installed here
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    class _Base(nn.Module):
        current_epoch = 0

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, name, value, **k):
            self.logged = getattr(self, "logged", {})
            self.logged[name] = value


# Command-line surface of model/plt.py:185-233: (flag, argparse keywords).  Names, types, choices and defaults are the
# reference's; the help strings are ours.
ENCODERS = ["resnest50", "resnest101", "resnest200", "resnest269", "resnet50", "resnet101", "resnet152"]
DAMAGE_MODELS = ["siamese", "siameseEnc", "fused", "fusedEnc", "parallel", "parallelEnc", "diff", "cat"]
OPTIMIZERS = ["sgd", "adam", "adamw", "radam", "adabelief", "adabound", "adamp", "novograd"]
_SWITCHES = {
    "use_scheduler": "Noam warm-up / decay schedule instead of a constant rate",
    "tta": "average the prediction over the identity and three flips at evaluation time",
    "ppm": "pyramid pooling module on the deepest encoder feature",
    "aspp": "atrous spatial pyramid pooling on the deepest encoder feature",
    "no_skip": "decoder without encoder skip connections",
    "deep_supervision": "auxiliary heads at 1/2 and 1/4 resolution during training",
    "attention": "attention gates on the skip connections",
    "autoaugment": "AutoAugment (ImageNet policy) instead of flips / noise / brightness",
    "interpolate": "upsample the encoder output directly, no decoder",
    "dec_interp": "bilinear upsampling + conv in the decoder instead of transposed conv",
}
_VALUED = [
    ("optimizer", dict(type=str, default="adamw", choices=OPTIMIZERS, help="update rule")),
    ("dmg_model", dict(type=str, default="siamese", choices=DAMAGE_MODELS, help="network family of the damage task")),
    ("encoder", dict(type=str, default="resnest200", choices=ENCODERS, help="backbone of the U-Net")),
    ("loss_str", dict(type=str, default="focal+dice",
                      help="'+'-joined terms out of dice, focal, ce, ohem, mse, coral")),
    ("warmup", dict(type=int, default=1, help="Noam schedule: epochs of linear warm-up")),
    ("init_lr", dict(type=float, default=1e-4, help="Noam schedule: rate at step 0")),
    ("final_lr", dict(type=float, default=1e-4, help="Noam schedule: rate at the last step")),
    ("lr", dict(type=float, default=3e-4, help="learning rate (peak rate under the Noam schedule)")),
    ("weight_decay", dict(type=float, default=0, help="decoupled weight decay")),
    ("momentum", dict(type=float, default=0.9, help="SGD momentum")),
    ("dilation", dict(type=int, default=1, choices=[1, 2, 4],
                      help="trade the stride of the last 1 (2) / 2 (4) encoder stages for dilation")),
]


class _JsonLines:
    """dllogger-shaped log: one 'DLLL {json}' line per record plus an echo on stdout (model/plt.py:35-40)"""

    def __init__(self, path):
        self.path = path

    def log(self, step, data):
        record = {"type": "LOG", "step": [] if step == () else step, "data": data}
        try:
            with open(self.path, "a") as fh:
                fh.write("DLLL %s\n" % json.dumps(record))
        except OSError:
            pass
        print("Epoch: %s %s" % (step, data))

    def flush(self):
        pass


def _is_rank0():
    return int(os.getenv("LOCAL_RANK", "0")) == 0


class Model(_Base):
    tta_flips = ([2], [3], [2, 3])

    def __init__(self, args):
        super().__init__()
        self.save_hyperparameters()
        self.args = args
        self.lr = args.lr
        self.n_class = 2 if args.type == "pre" else 5
        self.model = networks.UNetLoc(args) if args.type == "pre" else networks.get_dmg_unet(args)
        self.loss = criterion.Loss(args)
        self.softmax = nn.Softmax(dim=1)
        self.f1_score = F1(args)
        self.best_f1, self.best_epoch = torch.tensor(0), 0
        self.test_idx = 0
        log_dir, log_name = getattr(args, "results", "."), getattr(args, "logname", "logs")
        self.dllogger = _JsonLines(os.path.join(log_dir, log_name + ".json"))

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu"):
        """the argparse Namespace rides in the checkpoint as its hyper-parameters, Lightning style (model/plt.py:23)"""
        blob = torch.load(path, map_location=map_location, weights_only=False)
        hyper = blob["hyper_parameters"]
        model = cls(hyper["args"] if isinstance(hyper, dict) and "args" in hyper else hyper)
        model.load_state_dict(blob["state_dict"])
        return model

    # ---- forward / steps ----------------------------------------------------------------------------------------
    @staticmethod
    def flip(data, axis):
        if isinstance(data, ops.DeviceImage):       # uint8 tiles on the device: the flip rides in the normalise launch
            return data.flip(axis)
        return torch.flip(data, dims=axis)

    def forward(self, img):
        """logits; with --tta the mean over the identity and the three flips (model/plt.py:42-48)"""
        total = self.model(img)
        if not getattr(self.args, "tta", False):
            return total
        for dims in self.tta_flips:
            total = total + self.flip(self.model(self.flip(img, list(dims))), list(dims))
        return total / (1 + len(self.tta_flips))

    def compute_loss(self, preds, label):
        """deep-supervision weighting of model/plt.py:69-77 (criterion.compute_loss)"""
        return criterion.compute_loss(self.loss, preds, label, self.args.deep_supervision)

    def training_step(self, batch, _):
        return self.compute_loss(self.model(batch["image"]), batch["mask"])

    def _evaluate(self, batch):
        logits = self.forward(batch["image"])
        self.f1_score.update(logits, batch["mask"])
        return logits

    def validation_step(self, batch, _):
        logits = self._evaluate(batch)
        return {"val_loss": self.loss(logits, batch["mask"])}

    def test_step(self, batch, batch_idx):
        self.save(self._evaluate(batch), batch["mask"])

    # ---- epoch hooks --------------------------------------------------------------------------------------------
    def on_validation_epoch_start(self):
        self.f1_score.reset()

    on_test_epoch_start = on_validation_epoch_start

    @staticmethod
    def metric_mean(name, outputs):
        return torch.stack([o[name] for o in outputs]).mean(dim=0)

    @staticmethod
    def update_damage_scores(metrics, dmgs_f1):
        """per-class damage F1 as D1..D4 (model/plt.py:100-103)"""
        if dmgs_f1 is None:
            return
        for k in range(4):
            metrics["D%d" % (k + 1)] = round(dmgs_f1[k].item(), 3)

    def _finish_epoch(self):
        f1, per_class = self.f1_score.compute()
        self.f1_score.reset()
        return f1, per_class

    def _report(self, step, metrics, per_class):
        if not _is_rank0():
            return
        self.update_damage_scores(metrics, per_class)
        self.dllogger.log(step=step, data=metrics)
        self.dllogger.flush()

    def validation_epoch_end(self, outputs):
        val_loss = self.metric_mean("val_loss", outputs)
        f1, per_class = self._finish_epoch()
        if f1 >= self.best_f1:
            self.best_f1, self.best_epoch = f1, self.current_epoch
        self._report(self.current_epoch, {"f1": round(f1.item(), 3), "val_loss": round(val_loss.item(), 3),
                                          "top_f1": round(self.best_f1.item(), 3)}, per_class)
        self.log("f1_score", f1.cpu())
        self.log("val_loss", val_loss.cpu())

    def test_epoch_end(self, _):
        f1, per_class = self._finish_epoch()
        self._report((), {"f1": round(f1.item(), 3)}, per_class)

    # ---- evaluation artefacts -----------------------------------------------------------------------------------
    def _decode(self, preds):
        """what utils/post_process.py expects per tile (model/plt.py:126-135): building probability for the
        localization task, class probabilities (or the decoded ordinal / regression label) for the damage task"""
        if self.args.type == "pre":
            return torch.sigmoid(preds[:, 1])
        if self.args.loss_str == "coral":
            return (torch.sigmoid(preds) > 0.5).sum(dim=1) + 1
        if self.args.loss_str == "mse":
            return torch.round(torch.relu(preds[:, 0])) + 1
        return self.softmax(preds)

    def save(self, preds, targets):
        task = "localization" if self.args.type == "pre" else "damage"
        probs = self._decode(preds).detach().cpu().numpy()
        masks = targets.detach().cpu().numpy().astype(np.uint8)
        for prob, mask in zip(probs, masks):
            stem = "test_%s_%05d" % (task, self.test_idx)
            self.test_idx += 1
            np.save(os.path.join(self.args.results, "probs", stem), prob)
            target_base = os.path.join(self.args.results, "targets", stem + "_target")
            try:
                from PIL import Image
                Image.fromarray(mask).save(target_base + ".png")
            except ImportError:  # pragma: no cover
                np.save(target_base, mask)

    # ---- optimizer / schedule -----------------------------------------------------------------------------------
    def configure_optimizers(self):
        """model/plt.py:150-179.  AdamW (the default) is the flat fused HIP optimizer; sgd / adam / radam map to
        torch.optim; the apex / torch_optimizer-only choices are rejected with a clear message."""
        name, wd = self.args.optimizer.lower(), self.args.weight_decay
        if name == "adamw":
            optimizer = FlatAdamW(self.parameters(), lr=self.lr, weight_decay=wd)
        else:
            builders = {"sgd": lambda p: torch.optim.SGD(p, lr=self.lr, momentum=self.args.momentum),
                        "adam": lambda p: torch.optim.Adam(p, lr=self.lr, weight_decay=wd),
                        "radam": lambda p: torch.optim.RAdam(p, lr=self.lr, weight_decay=wd)}
            if name not in builders:
                raise NotImplementedError("--optimizer %s needs apex/torch_optimizer, which are CUDA-only / absent; "
                                          "use adamw (default), adam, sgd or radam" % name)
            optimizer = builders[name](self.parameters())
        if not self.args.use_scheduler:
            return optimizer
        # model/plt.py:170 uses len(self.train_dataloader()) // gpus on the UNSHARDED loader; here the trainer hands
        # over the length of the loader this rank really iterates (already one shard per rank: no second division)
        per_rank_steps = getattr(self, "steps_per_epoch_hint", None)
        if per_rank_steps is None:     # configure_optimizers() called outside a Trainer: the synthetic-data knob
            per_rank_steps = getattr(self.args, "steps_per_epoch", 1)
        per_rank_steps = max(1, int(per_rank_steps))
        noam = NoamLR(optimizer=optimizer, warmup_epochs=self.args.warmup, total_epochs=self.args.epochs,
                      steps_per_epoch=per_rank_steps, init_lr=self.args.init_lr, max_lr=self.args.lr,
                      final_lr=self.args.final_lr)
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": noam, "interval": "step", "frequency": 1}}

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        for flag, kw in _VALUED:
            parser.add_argument("--" + flag, **kw)
        for flag, text in _SWITCHES.items():
            parser.add_argument("--" + flag, action="store_true", help=text)
        return parser
