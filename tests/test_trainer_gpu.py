"""End-to-end CLI run on tiny synthetic tiles: train (with Noam schedule, deep supervision), checkpoint, resume-free
eval from the checkpoint with TTA, --ckpt_pre encoder transplant into a damage model."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cli_train_eval_and_transplant(tmp_path):
    import main as cli
    res = str(tmp_path / "pre")
    common = ["--data", "synthetic", "--encoder", "resnet50", "--precision", "32", "--batch_size", "2",
              "--val_batch_size", "2", "--train_size", "64", "--eval_size", "64", "--steps_per_epoch", "3"]
    m = cli.main(["--exec_mode", "train", "--type", "pre", "--loss_str", "dice", "--epochs", "2", "--results", res,
                  "--deep_supervision", "--use_scheduler", "--warmup", "1"] + common)
    ck = os.path.join(res, "checkpoints", "last.ckpt")
    assert os.path.exists(ck) and os.path.exists(os.path.join(res, "checkpoints", "best.ckpt"))
    sd = torch.load(ck, map_location="cpu", weights_only=False)["state_dict"]
    assert "model.unet.enc_l1.0.weight" in sd and "model.unet.dec_l1.conv_tranpose.conv.weight" in sd
    assert int(sd["model.unet.enc_l1.1.num_batches_tracked"]) == 6          # 2 epochs x 3 steps
    # eval from the checkpoint with TTA flips; writes the .npy probabilities the reference's post-processing reads
    res2 = str(tmp_path / "eval")
    cli.main(["--exec_mode", "eval", "--type", "pre", "--ckpt", ck, "--results", res2, "--tta"] + common)
    assert len(os.listdir(os.path.join(res2, "probs"))) == 4
    # damage model initialised from the localization checkpoint
    res3 = str(tmp_path / "post")
    mp = cli.main(["--exec_mode", "train", "--type", "post", "--dmg_model", "siamese", "--loss_str", "focal+dice",
                   "--epochs", "1", "--results", res3, "--ckpt_pre", ck] + common)
    assert os.path.exists(os.path.join(res3, "checkpoints", "last.ckpt"))
    assert mp.model.unet.enc_l1[0].weight.shape == m.model.unet.enc_l1[0].weight.shape


def test_cli_trains_and_evaluates_on_png_tiles(tmp_path, monkeypatch):
    """the PIL/numpy port of data_loading/pytorch_loader.py feeding the HIP path: 512x512 training crops from 640x640
    PNG tiles (B,G,R order, A.Normalize statistics), full-tile evaluation, .npy probabilities written"""
    import main as cli
    from tests.test_data_cpu import _tile_tree
    from xview2_amd.data_loading import pytorch_loader as pl
    root = str(tmp_path / "xbd")
    os.makedirs(root)
    monkeypatch.setattr(pl, "DEFAULT_INDEX", _tile_tree(root, n=4, S=640))
    res = str(tmp_path / "run")
    common = ["--data", root, "--type", "post", "--dmg_model", "siamese", "--encoder", "resnet50", "--precision", "32",
              "--batch_size", "2", "--val_batch_size", "2", "--num_workers", "0", "--loss_str", "focal+dice"]
    m = cli.main(["--exec_mode", "train", "--epochs", "1", "--results", res] + common)
    ck = os.path.join(res, "checkpoints", "last.ckpt")
    assert os.path.exists(ck)
    assert all(torch.isfinite(p).all() for p in m.parameters())
    res2 = str(tmp_path / "eval")
    cli.main(["--exec_mode", "eval", "--ckpt", ck, "--results", res2] + common)
    assert len(os.listdir(os.path.join(res2, "probs"))) == 4


def test_resume_continues_the_uninterrupted_run_bit_for_bit(tmp_path):
    """checkpoint + --ckpt resume: optimizer moments, the DEVICE step counter of the fused AdamW (bias correction) and
    the Noam schedule position are restored, so 1 epoch + resume + 1 epoch ends on exactly the parameters of an
    uninterrupted 2-epoch run (the synthetic loader replays the same batches each epoch; every kernel is deterministic)"""
    import shutil
    import main as cli
    common = ["--data", "synthetic", "--encoder", "resnet50", "--precision", "32", "--batch_size", "2",
              "--val_batch_size", "2", "--train_size", "64", "--eval_size", "64", "--steps_per_epoch", "3",
              "--exec_mode", "train", "--type", "pre", "--loss_str", "dice", "--use_scheduler", "--warmup", "1",
              "--final_lr", "1e-5"]
    full = cli.main(common + ["--epochs", "2", "--results", str(tmp_path / "full")])
    cli.main(common + ["--epochs", "1", "--results", str(tmp_path / "half")])
    ck = str(tmp_path / "half.ckpt")
    shutil.copy(os.path.join(str(tmp_path / "half"), "checkpoints", "last.ckpt"), ck)
    # (the 1-epoch run only covers the warm-up phase of the Noam schedule, which does not depend on --epochs)
    blob = torch.load(ck, map_location="cpu", weights_only=False)
    assert blob["global_step"] == 3 and blob["optimizer_states"][0]["step"] == 3
    resumed = cli.main(common + ["--epochs", "2", "--results", str(tmp_path / "res"), "--ckpt", ck])
    a, b = full.state_dict(), resumed.state_dict()
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
