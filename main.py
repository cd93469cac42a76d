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


def build_parser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter)
    arg = parser.add_argument
    arg("--exec_mode", type=str, choices=["train", "eval"], default="train", help="Execution mode of main script")
    arg("--data", type=str, default="/data", help="Path to the data directory ('synthetic' for generated tiles)")
    arg("--results", type=str, default="/results", help="Path to the results directory")
    arg("--gpus", type=int, default=1, help="Number of gpus to use")
    arg("--num_workers", type=int, default=8, help="Number of subprocesses to use for data loading")
    arg("--batch_size", type=int, default=16, help="Training batch size")
    arg("--val_batch_size", type=int, default=13, help="Evaluation batch size")
    arg("--precision", type=int, default=16, choices=[16, 32], help="Numerical precision")
    arg("--epochs", type=int, default=250, help="Max number of epochs")
    arg("--patience", type=int, default=100, help="Early stopping patience")
    arg("--ckpt", type=str, default=None, help="Path to pretrained checkpoint")
    arg("--logname", type=str, default="logs", help="Name of logging file")
    arg("--ckpt_pre", type=str, default=None,
        help="Path to pretrained checkpoint of localization model used to initialize network for damage assesment")
    arg("--type", type=str, choices=["pre", "post"],
        help="Type of task to run; pre - localization, post - damage assesment")
    arg("--seed", type=int, default=1)
    # synthetic-data knobs (not in the reference)
    arg("--train_size", type=int, default=512)
    arg("--eval_size", type=int, default=1024)
    arg("--steps_per_epoch", type=int, default=8)
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
