"""Whole-step hipGraph capture: forward + loss + backward (+ RCCL all-reduce) + AdamW as ONE graph launch.

A training step of the U-Net issues ~1500 small-to-medium kernels; replaying them from a captured hipGraph removes
the per-launch host cost and the inter-kernel gaps.  Requirements met by the HIP path: no allocation or
synchronisation inside the C ABI, the optimizer's learning rate / step counter live in device memory
(``FlatAdamW.capturable``), inputs are copied into static buffers before each replay.
"""
import torch

from . import nn as xnn


class GraphedStep:
    def __init__(self, step_fn, optimizer, static_inputs, warmup=2):
        """step_fn(*static_inputs) -> loss tensor; it must zero grads, run backward and the optimizer step."""
        self.opt = optimizer
        self.inputs = static_inputs
        optimizer.sync_lr()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                      # first-call work (attribute setting, allocator growth)
                step_fn(*static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._bumped = []
        orig = xnn.bump_bn_counter

        def recording_bump(bn):
            orig(bn)
            if bn.training and bn.num_batches_tracked is not None:
                self._bumped.append(bn)
        xnn.bump_bn_counter = recording_bump
        from . import ops
        # F16X2: the replays zero the operand-maximum slots they re-use (a pool owned by this graph)
        self._amax = ops.amax_begin_capture(optimizer.flat_p.device)
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                ops.amax_capture_started(self._amax)
                self.loss = step_fn(*static_inputs)
        finally:
            ops.amax_end_capture(self._amax)
            xnn.bump_bn_counter = orig
        # capture records the launches WITHOUT executing them: undo the host-side bookkeeping of that pass
        optimizer.step_count -= 1
        for bn in self._bumped:
            bn._xv2_pending -= 1

    def __call__(self, *new_inputs):
        for dst, src in zip(self.inputs, new_inputs):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        self.opt.sync_lr()
        self.graph.replay()
        # the replay rewrote parameters and BatchNorm running statistics without passing through Python: cached
        # derived data (folded inference coefficients, packed weights seen by later EAGER calls) is stale
        from . import ops
        ops.weights_changed()
        ops.bn_stats_changed()
        self.opt.step_count += 1
        for bn in self._bumped:                          # host-side num_batches_tracked bookkeeping
            bn._xv2_pending += 1
        return self.loss
