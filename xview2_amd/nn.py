"""Glue between parameter containers and the HIP ops.

``torch.nn.Conv2d`` / ``BatchNorm2d`` / ``ConvTranspose2d`` objects are used purely as PARAMETER CONTAINERS
(same constructor arguments, default init and ``state_dict`` keys as the reference's modules); their
``forward`` is never called - the functions below route them through the fused HIP autograd nodes.
Activations are NHWC.
"""
import torch
from torch import nn

from . import ops

# set by xview2_amd.dist when training data-parallel (reference: Trainer(sync_batchnorm=gpus > 1), main.py:106)
SYNC_BN = False


def _cfg(conv):
    g = getattr(conv, "_xv2_cfg", None)
    if g is None:
        kh, kw = conv.kernel_size
        assert conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
        assert conv.dilation[0] == conv.dilation[1]
        g = ops.conv_cfg(kh, kw, conv.stride[0], conv.padding[0], conv.dilation[0], conv.groups)
        conv._xv2_cfg = g
    return g


def _flush_bn_counter(bn, *_):
    n = getattr(bn, "_xv2_pending", 0)
    if n and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(n)
        bn._xv2_pending = 0


def bump_bn_counter(bn):
    """num_batches_tracked += 1 without a device launch per layer per step: counted on the host and written
    back whenever the module's state_dict is taken (checkpoint surface stays exact)."""
    # (parameters and buffers are read from the module's own dicts on the hot path: Module.__getattr__ is the slow fallback of the
    #  attribute lookup - 0.36 us a piece, 7700 of them per cfg5 step)
    if not (bn.training and bn._buffers.get("num_batches_tracked") is not None):
        return
    if not hasattr(bn, "_xv2_pending"):
        bn._xv2_pending = 0
        bn.register_state_dict_pre_hook(_flush_bn_counter)
    bn._xv2_pending += ops.BN_SPLIT          # a split batch counts as BN_SPLIT consecutive batches


class bn_split:
    """`with bn_split(2):` - the batch entering the network holds 2 independent BatchNorm batches back to back
    (ops.BN_SPLIT): SiameseUNet's shared-weight passes over the pre and the post image as ONE batch."""

    def __init__(self, parts):
        self.parts = parts

    def __enter__(self):
        self.old, ops.BN_SPLIT = ops.BN_SPLIT, self.parts

    def __exit__(self, *exc):
        ops.BN_SPLIT = self.old


FUSED_INFERENCE = True     # eval + no_grad: one launch per conv layer (tests switch it off to compare both forms)


def conv_bn_act(conv, bn, x0, x1=None, act=ops.ACT_NONE, residual=None, passthrough=False):
    """act(BN(conv(cat(x0, x1))) [+ residual]) - one fused autograd node.  passthrough=True (or 1) returns (out, x0 alias):
    hand the alias to x0's other consumer and the two gradients are summed inside the backward-data kernel; 2 = alias of x1,
    3 = (out, x0 alias, x1 alias)."""
    passthrough = int(passthrough) & (3 if x1 is not None else 1)
    if not bn.training and not torch.is_grad_enabled() and x0.is_cuda and FUSED_INFERENCE:
        # inference: BatchNorm folded into the convolution epilogue (xv2_conv2d_forward_fused), no autograd node
        z = ops.conv_bn_act_infer(x0, x1, conv.weight, residual, _cfg(conv), ops.BnState(bn, False), act)
        return _with_aliases(z, x0, x1, passthrough)
    bump_bn_counter(bn)
    bs = ops.BnState(bn, SYNC_BN)
    out = ops.ConvBnActFn.apply(x0, x1, conv._parameters["weight"], bs.weight, bs.bias, residual, _cfg(conv),
                                bs, act, bn.training, passthrough)
    if passthrough == 3:
        ops.carry_amax(x0, out[1])
        ops.carry_amax(x1, out[2])
    elif passthrough:
        ops.carry_amax(x0 if passthrough == 1 else x1, out[1])
    return out


def _with_aliases(z, x0, x1, passthrough):
    if passthrough == 3:
        return z, x0, x1
    if passthrough == 2:
        return z, x1
    return (z, x0) if passthrough else z


def want_aliases(*xs):
    """pass-through aliases only matter (and only work) when gradients flow: every tensor on the device and differentiable"""
    return torch.is_grad_enabled() and all(x is not None and x.is_cuda and x.requires_grad for x in xs)


def conv(conv_m, x0, x1=None):
    return ops.ConvFn.apply(x0, x1, conv_m.weight, conv_m.bias, _cfg(conv_m))


def bn_act(bn, y, act=ops.ACT_NONE, residual=None):
    bump_bn_counter(bn)
    bs = ops.BnState(bn, SYNC_BN)
    return ops.BnActFn.apply(y, bs.weight, bs.bias, residual, bs, act, bn.training)


def head_conv(conv_m, x, nchw_out=True):
    return ops.HeadConvFn.apply(x, conv_m.weight, conv_m.bias, nchw_out)


def cat_channels(*xs):
    return ops.CatChannelsFn.apply(*xs)


class Numbered(nn.Module):
    """Container whose children are registered under '0', '1', ... like nn.Sequential (so checkpoints keep the
    reference's keys), but with an explicit forward supplied by subclasses."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m if m is not None else nn.Identity())

    def __getitem__(self, i):
        return getattr(self, str(i))

    def __len__(self):
        return len(self._modules)


class Chain(Numbered):
    """nn.Sequential equivalent for NHWC HIP blocks."""

    def forward(self, x):
        for m in self._modules.values():
            x = m(x)
        return x


# A tensor with TWO consumers (an encoder stage's output: the next stage and the decoder's skip connection,
# model/unet.py:150-170) costs autograd an elementwise sum of the two gradients - unless the consumer that runs its backward
# LAST adds its contribution onto the other one's gradient inside its own kernel.  The first layers of a stage can do that
# (ops.ConvBnActFn / MaxPool3x3s2Fn / AvgPoolFn with passthrough): asked to, the stage's first block publishes an ALIAS of its
# input; the other consumers read the alias, their gradient reaches the block's first layers as `dpass` and is summed in the
# backward-data epilogue / pooling backward.  Blocks keep returning a single tensor (forward hooks, nn.Sequential semantics):
# the alias travels through two attributes of the block.
def stage_with_input_alias(stage, x, *more):
    """-> (stage(x, *more), tensor that x's OTHER consumers should read): the alias published by the stage's first block, or x"""
    first = stage
    while isinstance(first, Chain) and len(first) > 0:
        first = first[0]
    if not (want_aliases(x) and hasattr(first, "alias_request")):
        return stage(x, *more), x
    first.alias_request = True
    try:
        y = stage(x, *more)
    finally:
        first.alias_request = False
    alias, first.alias_out = first.alias_out, None
    return y, (alias if alias is not None else x)


def to_nhwc_image(x_nchw):
    """[N,3,H,W] image (possibly a channel slice of the 6-channel pre/post pair) -> NHWC padded to 4 channels."""
    return ops.nchw_to_nhwc(x_nchw, 4)
