"""``Model``: the PyTorch-Lightning module surface of the reference's model/plt.py re-hosted on the HIP path.

Same constructor (``Model(args)``), methods and CLI flags (``add_model_specific_args`` reproduces every flag,
choice and default of model/plt.py:181-234).  pytorch_lightning is not installed in this image: when it is
importable ``Model`` derives from ``pl.LightningModule`` and can be handed to ``pl.Trainer`` unchanged,
otherwise from a minimal stand-in and is driven by xview2_amd.trainer.  The optimizers that exist only in
apex / torch_optimizer map to their torch.optim equivalents when there is one.
"""
import json
import os
from argparse import ArgumentParser

import numpy as np
import torch
from torch import nn

from . import criterion, networks
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


class _JsonLogger:
    """dllogger-shaped JSON-lines logger (model/plt.py:35-40,105-112)"""

    def __init__(self, path):
        self.path = path

    def log(self, step, data):
        line = {"type": "LOG", "step": step if step != () else [], "data": data}
        try:
            with open(self.path, "a") as f:
                f.write("DLLL " + json.dumps(line) + "\n")
        except OSError:
            pass
        print("Epoch: %s %s" % (step, data))

    def flush(self):
        pass


class Model(_Base):
    def __init__(self, args):
        super().__init__()
        self.save_hyperparameters()
        self.args = args
        self.f1_score = F1(args)
        self.model = networks.UNetLoc(args) if args.type == "pre" else networks.get_dmg_unet(args)
        self.loss = criterion.Loss(args)
        self.best_f1 = torch.tensor(0)
        self.best_epoch = 0
        self.tta_flips = [[2], [3], [2, 3]]
        self.lr = args.lr
        self.n_class = 2 if self.args.type == "pre" else 5
        self.softmax = nn.Softmax(dim=1)
        self.test_idx = 0
        results = getattr(args, "results", ".")
        self.dllogger = _JsonLogger(os.path.join(results, "%s.json" % getattr(args, "logname", "logs")))

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu"):
        """PL-style: hyper-parameters (the argparse Namespace) travel inside the checkpoint (model/plt.py:23)"""
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
        hp = ckpt["hyper_parameters"]
        model = cls(hp["args"] if isinstance(hp, dict) and "args" in hp else hp)
        model.load_state_dict(ckpt["state_dict"])
        return model

    def forward(self, img):  # model/plt.py:42-48
        pred = self.model(img)
        if getattr(self.args, "tta", False):
            for flip_idx in self.tta_flips:
                pred = pred + self.flip(self.model(self.flip(img, flip_idx)), flip_idx)
            pred = pred / (len(self.tta_flips) + 1)
        return pred

    def training_step(self, batch, _):  # model/plt.py:50-54
        img, lbl = batch["image"], batch["mask"]
        pred = self.model(img)
        return self.compute_loss(pred, lbl)

    def validation_step(self, batch, _):
        img, lbl = batch["image"], batch["mask"]
        pred = self.forward(img)
        loss = self.loss(pred, lbl)
        self.f1_score.update(pred, lbl)
        return {"val_loss": loss}

    def test_step(self, batch, batch_idx):
        img, lbl = batch["image"], batch["mask"]
        pred = self.forward(img)
        self.f1_score.update(pred, lbl)
        self.save(pred, lbl)

    def compute_loss(self, preds, label):  # model/plt.py:69-77
        return criterion.compute_loss(self.loss, preds, label, self.args.deep_supervision)

    @staticmethod
    def metric_mean(name, outputs):
        return torch.stack([out[name] for out in outputs]).mean(dim=0)

    @staticmethod
    def update_damage_scores(metrics, dmgs_f1):
        if dmgs_f1 is not None:
            for i in range(4):
                metrics.update({"D%d" % (i + 1): round(dmgs_f1[i].item(), 3)})

    def on_validation_epoch_start(self):
        self.f1_score.reset()

    def on_test_epoch_start(self):
        self.f1_score.reset()

    def validation_epoch_end(self, outputs):
        loss = self.metric_mean("val_loss", outputs)
        f1_score, dmgs_f1 = self.f1_score.compute()
        self.f1_score.reset()
        if f1_score >= self.best_f1:
            self.best_f1 = f1_score
            self.best_epoch = self.current_epoch
        if int(os.getenv("LOCAL_RANK", "0")) == 0:
            metrics = {"f1": round(f1_score.item(), 3), "val_loss": round(loss.item(), 3),
                       "top_f1": round(self.best_f1.item(), 3)}
            self.update_damage_scores(metrics, dmgs_f1)
            self.dllogger.log(step=self.current_epoch, data=metrics)
            self.dllogger.flush()
        self.log("f1_score", f1_score.cpu())
        self.log("val_loss", loss.cpu())

    def test_epoch_end(self, _):
        f1_score, dmgs_f1 = self.f1_score.compute()
        self.f1_score.reset()
        if int(os.getenv("LOCAL_RANK", "0")) == 0:
            metrics = {"f1": round(f1_score.item(), 3)}
            self.update_damage_scores(metrics, dmgs_f1)
            self.dllogger.log(step=(), data=metrics)
            self.dllogger.flush()

    def save(self, preds, targets):  # model/plt.py:126-144 (.npy probabilities + target PNGs)
        if self.args.type == "pre":
            probs = torch.sigmoid(preds[:, 1])
        elif self.args.loss_str == "coral":
            probs = torch.sum(torch.sigmoid(preds) > 0.5, dim=1) + 1
        elif self.args.loss_str == "mse":
            probs = torch.round(torch.relu(preds[:, 0])) + 1
        else:
            probs = self.softmax(preds)
        probs = probs.cpu().detach().numpy()
        targets = targets.cpu().detach().numpy().astype(np.uint8)
        for prob, target in zip(probs, targets):
            task = "localization" if self.args.type == "pre" else "damage"
            fname = os.path.join(self.args.results, "probs", "test_%s_%05d" % (task, self.test_idx))
            self.test_idx += 1
            np.save(fname, prob)
            try:
                from PIL import Image
                Image.fromarray(target).save(fname.replace("probs", "targets") + "_target.png")
            except ImportError:  # pragma: no cover
                np.save(fname.replace("probs", "targets") + "_target", target)

    @staticmethod
    def flip(data, axis):
        return torch.flip(data, dims=axis)

    def configure_optimizers(self):  # model/plt.py:150-179
        name = self.args.optimizer.lower()
        wd = self.args.weight_decay
        if name == "adamw":
            optimizer = FlatAdamW(self.parameters(), lr=self.lr, weight_decay=wd)
        else:
            table = {"sgd": lambda p: torch.optim.SGD(p, lr=self.lr, momentum=self.args.momentum),
                     "adam": lambda p: torch.optim.Adam(p, lr=self.lr, weight_decay=wd),
                     "radam": lambda p: torch.optim.RAdam(p, lr=self.lr, weight_decay=wd)}
            if name not in table:
                raise NotImplementedError("--optimizer %s needs apex/torch_optimizer, which are CUDA-only / absent; "
                                          "use adamw (default), adam, sgd or radam" % name)
            optimizer = table[name](self.parameters())
        if not self.args.use_scheduler:
            return optimizer
        steps = max(1, getattr(self.args, "steps_per_epoch", 1) // max(1, getattr(self.args, "gpus", 1)))
        scheduler = {"scheduler": NoamLR(optimizer=optimizer, warmup_epochs=self.args.warmup,
                                         total_epochs=self.args.epochs, steps_per_epoch=steps,
                                         init_lr=self.args.init_lr, max_lr=self.args.lr, final_lr=self.args.final_lr),
                     "interval": "step", "frequency": 1}
        return {"optimizer": optimizer, "lr_scheduler": scheduler}

    @staticmethod
    def add_model_specific_args(parent_parser):  # model/plt.py:181-234 (flags, choices, defaults verbatim)
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        arg = parser.add_argument
        arg("--optimizer", type=str, default="adamw",
            choices=["sgd", "adam", "adamw", "radam", "adabelief", "adabound", "adamp", "novograd"])
        arg("--dmg_model", type=str, default="siamese",
            choices=["siamese", "siameseEnc", "fused", "fusedEnc", "parallel", "parallelEnc", "diff", "cat"],
            help="U-Net variant for damage assessment task")
        arg("--encoder", type=str, default="resnest200",
            choices=["resnest50", "resnest101", "resnest200", "resnest269", "resnet50", "resnet101", "resnet152"],
            help="U-Net encoder")
        arg("--loss_str", type=str, default="focal+dice",
            help="Combination of: dice, focal, ce, ohem, mse, coral, e.g focal+dice creates the loss function as sum of focal and dice")
        arg("--use_scheduler", action="store_true", help="Enable Noam learning rate scheduler")
        arg("--warmup", type=int, default=1, help="Warmup epochs for Noam learning rate scheduler")
        arg("--init_lr", type=float, default=1e-4, help="Initial learning rate for Noam scheduler")
        arg("--final_lr", type=float, default=1e-4, help="Final learning rate for Noam scheduler")
        arg("--lr", type=float, default=3e-4, help="Learning rate, or a target learning rate for Noam scheduler")
        arg("--weight_decay", type=float, default=0, help="Weight decay (L2 penalty)")
        arg("--momentum", type=float, default=0.9, help="Momentum for SGD optimizer")
        arg("--dilation", type=int, choices=[1, 2, 4], default=1,
            help="Dilation rate for a encoder, e.g dilation=2 uses dilation instead of stride in the last encoder block")
        arg("--tta", action="store_true", help="Enable test time augmentation")
        arg("--ppm", action="store_true", help="Use pyramid pooling module")
        arg("--aspp", action="store_true", help="Use atrous spatial pyramid pooling")
        arg("--no_skip", action="store_true", help="Disable skip connections in UNet")
        arg("--deep_supervision", action="store_true", help="Enable deep supervision")
        arg("--attention", action="store_true", help="Enable attention module at the decoder")
        arg("--autoaugment", action="store_true", help="Use imageNet autoaugment pipeline")
        arg("--interpolate", action="store_true", help="Interpolate feature map from encoder without a decoder")
        arg("--dec_interp", action="store_true", help="Use interpolation instead of transposed convolution in a decoder")
        return parser
