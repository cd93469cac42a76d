#!/usr/bin/env python
"""Command line of the reference's main.py (main.py:26-125: same flags, choices and defaults) driving the HIP path.

    python main.py --exec_mode train --type pre --encoder resnet50 --loss_str dice --precision 32 --data synthetic
    torchrun --nproc-per-node 8 main.py --gpus 8 --type post --dmg_model siamese ...       (one process per GPU)

``--data <dir>`` reads xBD tiles laid out like the reference expects (<dir>/{train,test,holdout}/{images,targets}/*.png,
data_loading/data_module.py:12-14) through a PIL/numpy port of its loader; ``--data synthetic`` generates tiles of the
same contract.  ``--gpus N`` expects to be launched once per GPU (torchrun) instead of PL's self re-exec.  CPU affinity
follows main.py:62 (socket_unique_interleaved) with the GPU's NUMA-local cores read from sysfs instead of NVML."""
import os
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

import torch

from xview2_amd.data import SyntheticDataModule
from xview2_amd.data_loading.data_module import DataModule
from xview2_amd.lightning import Model
from xview2_amd.trainer import Trainer
from xview2_amd.utils.gpu_affinity import set_affinity


def set_hip_devices(gpus):
    """main.py:20-23 (CUDA_VISIBLE_DEVICES) for ROCm; a torchrun launch already scopes each rank through LOCAL_RANK"""
    if torch.cuda.is_available():
        assert gpus <= torch.cuda.device_count(), "Requested %d gpus, available %d." % (gpus, torch.cuda.device_count())
    os.environ.setdefault("HIP_VISIBLE_DEVICES", ",".join(str(i) for i in range(gpus)))


def transplant_encoder(model, pretrained_sd, dmg_model):
    """--ckpt_pre (main.py:76-94): copy every tensor whose key contains "enc" from a localization checkpoint into the
    damage model (the reference's `model.state_dict()[keys]` typo for the parallel variants is not reproduced)."""
    sd = model.state_dict()
    copied = 0
    for name, tensor in pretrained_sd.items():
        if "enc" not in name:
            continue
        if "parallel" in dmg_model:
            targets = [name.replace("unet", "unet_pre"), name.replace("unet", "unet_post")]
        elif dmg_model == "siameseEnc":
            targets = [name.replace(".unet", "")]
        else:
            targets = [name]
        for key in targets:
            if key in sd and sd[key].shape == tensor.shape:
                sd[key].copy_(tensor)
                copied += 1
    return copied


# Launcher flags of the reference (main.py:29-53): same names, types, choices and defaults; help strings are ours.
_LAUNCH_FLAGS = [
    ("exec_mode", dict(type=str, choices=["train", "eval"], default="train", help="train a model or evaluate a checkpoint")),
    ("data", dict(type=str, default="/data", help="xBD directory with train/ test/ holdout/, or 'synthetic'")),
    ("results", dict(type=str, default="/results", help="where checkpoints, logs and predictions go")),
    ("gpus", dict(type=int, default=1, help="GPUs of this node (one process each, e.g. under torchrun)")),
    ("num_workers", dict(type=int, default=8, help="loader worker processes")),
    ("batch_size", dict(type=int, default=16, help="samples per training step and GPU")),
    ("val_batch_size", dict(type=int, default=13, help="samples per evaluation step and GPU")),
    ("precision", dict(type=int, default=16, choices=[16, 32], help="32: exact fp32; 16: bf16 matrix math")),
    ("epochs", dict(type=int, default=250, help="training epochs")),
    ("patience", dict(type=int, default=100, help="early-stopping patience (epochs)")),
    ("ckpt", dict(type=str, default=None, help="checkpoint to resume from / to evaluate")),
    ("logname", dict(type=str, default="logs", help="stem of the JSON-lines log file")),
    ("ckpt_pre", dict(type=str, default=None,
                      help="localization checkpoint whose encoder initialises the damage model")),
    ("type", dict(type=str, choices=["pre", "post"], help="pre: building localization; post: damage assessment")),
    ("seed", dict(type=int, default=1)),
    # synthetic-data knobs (not in the reference)
    ("train_size", dict(type=int, default=512)),
    ("eval_size", dict(type=int, default=1024)),
    ("steps_per_epoch", dict(type=int, default=8)),
]


def build_parser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter)
    for flag, kw in _LAUNCH_FLAGS:
        parser.add_argument("--" + flag, **kw)
    return Model.add_model_specific_args(parser)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.interpolate:
        args.deep_supervision = False
        args.dec_interp = False
    set_hip_devices(args.gpus)
    try:
        set_affinity(os.getenv("LOCAL_RANK", "0"), "socket_unique_interleaved")     # main.py:62
    except (OSError, RuntimeError):
        pass                                                                        # restricted cpusets: keep the default
    torch.manual_seed(args.seed)
    os.makedirs(args.results, exist_ok=True)
    checkpoint = args.ckpt if args.ckpt is not None and os.path.exists(args.ckpt) else None
    if args.exec_mode == "train":
        model = Model(args)
    else:
        assert args.ckpt is not None, "No checkpoint found for evaluation"
        model = Model.load_from_checkpoint(args.ckpt)
        # the architecture comes from the checkpoint's hyper-parameters; run-time options come from this command
        for k in ("results", "logname", "tta", "val_batch_size", "seed"):
            setattr(model.args, k, getattr(args, k))
        model.dllogger.path = os.path.join(args.results, "%s.json" % args.logname)
    if args.type == "post" and args.ckpt_pre is not None:
        pre = torch.load(args.ckpt_pre, map_location="cpu", weights_only=False)["state_dict"]
        transplant_encoder(model, pre, args.dmg_model)
    trainer = Trainer(gpus=args.gpus, precision=args.precision, max_epochs=args.epochs, min_epochs=args.epochs,
                      sync_batchnorm=args.gpus > 1, accelerator="ddp" if args.gpus > 1 else None,
                      default_root_dir=args.results, checkpoint_callback=args.exec_mode == "train",
                      resume_from_checkpoint=checkpoint)
    if args.data == "synthetic":
        dm = SyntheticDataModule(args, device=trainer.device, rank=trainer.rank, train_size=args.train_size,
                                 eval_size=args.eval_size, steps_per_epoch=args.steps_per_epoch)
    else:
        dm = DataModule(args, device=trainer.device, rank=trainer.rank, world_size=trainer.world)
    if args.exec_mode == "train":
        trainer.fit(model, dm)
    else:
        for sub in ("probs", "targets"):
            os.makedirs(os.path.join(args.results, sub), exist_ok=True)
        trainer.test(model, test_dataloaders=dm.test_dataloader())
    return model


if __name__ == "__main__":
    main()
