"""ORACLE tooling (build container only): stand-in packages that let the REAL reference files
/root/reference/model/{unet,layers,loss,plt}.py and utils/f1.py be imported where torchvision, resnest,
monai, pytorch_lightning, apex, dllogger and torch_optimizer are not installed.  The stand-ins are
backed by oracle.backbones / oracle.torch_ref (our restatements of those third-party pieces), so what
gets pinned by the golden vectors is the reference's OWN wiring code.  Never used on the GPU box."""
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"


def install():
    from . import backbones, torch_ref

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tvm = mod("torchvision.models", resnet50=backbones.resnet50, resnet101=backbones.resnet101,
              resnet152=backbones.resnet152)
    mod("torchvision", models=tvm)
    rt = mod("resnest.torch", resnest50=backbones.resnest50, resnest101=backbones.resnest101,
             resnest200=backbones.resnest200, resnest269=backbones.resnest269)
    mod("resnest", torch=rt)

    class DiceLoss(torch_ref.DiceLoss):
        def __init__(self, include_background=True, softmax=False, to_onehot_y=False, batch=False, **kw):
            assert not kw, kw
            super().__init__(include_background, softmax, to_onehot_y, batch)

    class FocalLoss(torch_ref.FocalLoss):
        def __init__(self, gamma=2.0, **kw):
            assert not kw, kw
            super().__init__(gamma)

    ml = mod("monai.losses", DiceLoss=DiceLoss, FocalLoss=FocalLoss)
    mod("monai", losses=ml)

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    class Metric(nn.Module):
        def __init__(self, dist_sync_on_step=False, **kw):
            super().__init__()
            self._defaults = {}

        def add_state(self, name, default, dist_reduce_fx=None):
            self._defaults[name] = default
            setattr(self, name, default.clone())

        def reset(self):
            for k, v in self._defaults.items():
                setattr(self, k, v.clone())

    plm = mod("pytorch_lightning.metrics", Metric=Metric)
    mod("pytorch_lightning", LightningModule=LightningModule, metrics=plm)

    class _Opt(torch.optim.SGD):
        def __init__(self, params, lr=1e-3, **kw):
            super().__init__(params, lr=lr)

    mod("apex.optimizers", FusedAdam=_Opt, FusedNovoGrad=_Opt, FusedSGD=_Opt)
    mod("apex")
    mod("torch_optimizer", AdaBelief=_Opt, AdaBound=_Opt, AdamP=_Opt, RAdam=_Opt)

    class _Logger:
        def __init__(self, backends=None):
            pass

        def log(self, *a, **k):
            pass

        def flush(self):
            pass

    mod("dllogger", JSONStreamBackend=lambda *a, **k: None, StdOutBackend=lambda *a, **k: None, Logger=_Logger,
        Verbosity=types.SimpleNamespace(VERBOSE=0))
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
