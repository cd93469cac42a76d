"""Loss composition on the fused HIP loss kernels: drop-in for the reference's model/loss.py ``Loss``.

``--loss_str`` is a '+'-separated sum of dice, focal, ce, ohem (model/loss.py:68-83).  dice/focal are the
monai 0.4.0 losses with the constructor arguments of model/loss.py:11-13; ``mse`` / ``coral`` (model/loss.py:54-65,92-94) have their own kernels; ``ohem`` is numerically the mean
cross-entropy (the reference slices the (values, indices) tuple of ``sort`` at model/loss.py:45, so no
negative is ever dropped).  All terms of one call share a single softmax pass; for ``--type post`` the
building mask of model/loss.py:86-90 is applied inside the kernels (no compaction, order-independent sums).
"""
import torch
from torch import nn

from . import ops

_TERM_BITS = {"dice": ops.LOSS_DICE, "focal": ops.LOSS_FOCAL, "ce": ops.LOSS_CE, "ohem": ops.LOSS_CE}


class Loss(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.loss_str = args.loss_str
        self.post = args.type == "post"
        self.names = self.loss_str.split("+")
        for n in self.names:
            if n not in ("dice", "focal", "ce", "ohem", "mse", "coral"):
                raise KeyError(n)
        # The reference special-cases loss_str == "mse" (float targets, model/loss.py:92-94) and builds the 3-logit coral head
        # only for loss_str == "coral" (model/unet.py:21-26): combined with other terms both fail in the reference's FIRST
        # forward with a RuntimeError (nn.MSELoss on [M, 4] logits against [M] long labels: "The size of tensor a (4) must match
        # the size of tensor b (M)"; CORAL's [M, 4] logits against its [M, 3] levels: "... tensor a (4) ... tensor b (3) ...").
        # Same here: construction succeeds, the first forward raises RuntimeError.
        self.unsupported_combo = ("mse" in self.names or "coral" in self.names) and len(self.names) > 1

    def forward(self, y_pred, y_true, label_stride=1):
        """y_pred NCHW logits; y_true [N, H*label_stride, W*label_stride] uint8/long labels."""
        if self.unsupported_combo:
            bad = "mse" if "mse" in self.names else "coral"
            raise RuntimeError("--loss_str %s: the size of tensor a (%d) must match the size of tensor b (%s) at non-singleton "
                               "dimension 1 (%s does not compose with other loss terms: model/loss.py:92-99 fails the same way)"
                               % (self.loss_str, y_pred.shape[1], "3" if bad == "coral" else "M", bad))
        if self.names[0] in ("mse", "coral"):
            bits = ops.LOSS_MSE if self.names[0] == "mse" else ops.LOSS_CORAL
            return ops.LossFn.apply(y_pred, y_true, bits, self.post, label_stride)
        # the reference sums the terms one by one; duplicated names count twice
        total = None
        counts = {}
        for n in self.names:
            counts[_TERM_BITS[n]] = counts.get(_TERM_BITS[n], 0) + 1
        while counts:
            bits = 0
            for b in list(counts):
                bits |= b
                counts[b] -= 1
                if counts[b] == 0:
                    del counts[b]
            part = ops.LossFn.apply(y_pred, y_true, bits, self.post, label_stride)
            total = part if total is None else total + part
        return total


def compute_loss(loss_fn, preds, label, deep_supervision):
    """model/plt.py:69-77: L(out) + 1/2 L(out_dec4, lbl[::2, ::2]) + 1/4 L(out_dec3, lbl[::4, ::4]), times
    1/(2 - 2^-3); the nearest-neighbour label down-sampling is index arithmetic inside the loss kernel."""
    if not deep_supervision:
        return loss_fn(preds, label)
    loss = loss_fn(preds[0], label)
    for i, pred in enumerate(preds[1:]):
        stride = label.shape[-1] // pred.shape[-1]
        loss = loss + 0.5 ** (i + 1) * loss_fn(pred, label, label_stride=stride)
    c_norm = 1 / (2 - 2 ** (-len(preds)))
    return c_norm * loss
