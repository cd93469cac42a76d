"""Deterministic, construction-order-independent parameter initialisation.

The reference downloads ImageNet weights at construction (model/unet.py:45 `pretrained=True`), which is
impossible offline.  For benchmarks and parity tests every tensor of a ``state_dict`` is instead filled
from a generator seeded by (seed, crc32(key)), so the reference, the CPU oracle and the HIP model get
bit-identical weights regardless of module registration order.
"""
import math
import zlib

import torch


def fill_by_key_(state_dict, seed=1):
    for key in sorted(state_dict.keys()):
        t = state_dict[key]
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t.zero_()
            continue
        shape = tuple(t.shape)
        if leaf == "running_mean":
            v = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            v = torch.rand(shape, generator=g) + 0.5
        elif t.dim() == 4 and shape[-1] * shape[-2] * shape[1] > 1:
            fan_in = shape[1] * shape[2] * shape[3]
            v = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif t.dim() == 2:
            v = torch.randn(shape, generator=g) * math.sqrt(1.0 / shape[1])
        elif leaf == "weight" and t.dim() == 1:
            v = torch.rand(shape, generator=g) + 0.5        # BatchNorm gamma
        elif leaf == "bias":
            v = torch.randn(shape, generator=g) * 0.1
        else:
            v = torch.randn(shape, generator=g) * 0.5
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return state_dict


def deterministic_init_(module, seed=1):
    """Fill every parameter and buffer of `module` in place (works for aliased FusedUNet entries too)."""
    fill_by_key_(module.state_dict(keep_vars=False), seed)
    return module
