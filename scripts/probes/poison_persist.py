"""Run ON the GPU box: do the results of a training step depend on what the grow-only workspaces hold?  (ops._persist_bufs: statistics
partials, split-K slabs, BatchNorm-backward partials; the side stream's weight-gradient workspace; the statistics-reduction scratch.)
They are never freed, so the allocator-poisoning run of scripts/crowd_diag.py does not reach them; what they hold at the start of a
step is whatever the previous step - or the previous TEST in the same worker process - left there.  Two runs of the full-size step
with the buffers filled with a pattern in between must agree bit for bit.   usage: poison_persist.py [pattern] [key=value overrides]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_model_gpu import _cfg2_step  # noqa: E402
from xview2_amd import ops  # noqa: E402

pat = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x7fc07fc0
over = dict(kv.split("=", 1) for kv in sys.argv[2:])
l1, z1, g1, p1, _, _ = _cfg2_step(**over)
n = 0
bufs = list(ops._persist_bufs.values()) + list(ops._side_ws.values()) + list(ops._scratch.values())
for t in bufs:
    if t.dtype == torch.float64:
        t.view(torch.int32).fill_(pat - (1 << 32) if pat >= (1 << 31) else pat)
    else:
        t.view(torch.int32).fill_(pat - (1 << 32) if pat >= (1 << 31) else pat)
    n += t.numel() * t.element_size()
torch.cuda.synchronize()
l2, z2, g2, p2, _, _ = _cfg2_step(**over)
print("%s pattern %#x over %d buffers (%.1f MB): loss %s logits %s gradients %s parameters %s" % (
    over or "resnet50", pat, len(bufs), n / 1e6, l1 == l2, torch.equal(z1, z2), torch.equal(g1, g2), torch.equal(p1, p2)))
print("  losses", l1, l2)
