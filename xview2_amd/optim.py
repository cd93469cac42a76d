"""AdamW on one flat parameter/gradient buffer: one HIP launch per step (model/plt.py:154 uses
torch.optim.AdamW as the default --optimizer; same update rule, decoupled weight decay).

All parameters are re-pointed at views of a single fp32 buffer and their ``.grad`` at views of a second one,
so (a) the optimizer step is a single streaming kernel over 4 arrays, (b) the data-parallel reducer
(xview2_amd.dist) all-reduces contiguous slices without packing copies.
"""
import torch

from . import ops


class FlatAdamW:
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        seen, plist = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:      # FusedUNet registers every stage twice
                seen.add(id(p))
                plist.append(p)
        self.params = plist
        dev = plist[0].device
        sizes = [p.numel() for p in plist]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 3) // 4 * 4                      # keep every view 16-byte aligned
        self.offsets, self.total = offs, total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(plist, offs):
                view = self.flat_p[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = None
                # the HIP backward kernels write a parameter's gradient straight into its slice of flat_g
                # (ops.grad_slot): autograd then adopts that view as .grad without an accumulation kernel
                p._xv2_slot = (self, o)
                p._xv2_epoch = -1
        ops.clear_pack_cache()         # the parameters just moved to new storage
        self.epoch = 0
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.param_groups = [{"lr": lr, "params": plist}]  # what utils/scheduler.py NoamLR touches
        # device-resident copies for hipGraph capture (see xview2_amd.graph.GraphedStep)
        self.capturable = dev.type == "cuda"   # always the device-state kernel on the GPU: one code path, graph-safe
        self.lr_dev = torch.tensor([lr], dtype=torch.float32, device=dev) if dev.type == "cuda" else None
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev) if dev.type == "cuda" else None
        self._lr_on_dev = lr

    def zero_grad(self):
        ops.begin_step()               # (a backward pass that raised leaves its side-stream bookkeeping behind)
        self.flat_g.zero_()            # one memset; parameters that get no gradient this step stay at zero
        self.epoch += 1
        for p in self.params:
            p.grad = None

    def _gather_foreign_grads(self):
        """gradients that did not come from a HIP backward kernel (torch-side ops) are copied into their slice"""
        base = self.flat_g.data_ptr()
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is not None and g.data_ptr() != base + 4 * o:
                self.flat_g[o:o + p.numel()].view_as(p).copy_(g)
                p.grad = self.flat_g[o:o + p.numel()].view_as(p)

    def sync_lr(self):
        """push the host-side learning rate to the device copy (call OUTSIDE a captured region)"""
        lr = self.param_groups[0]["lr"]
        if self.lr_dev is not None and lr != self._lr_on_dev:
            self.lr_dev.fill_(lr)
            self._lr_on_dev = lr

    def step(self, grad_scale=1.0):
        ops.join_wgrad_stream()      # weight-gradient kernels run on a side stream (ops.ASYNC_WGRAD)
        self._gather_foreign_grads()
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        if self.capturable:
            from ._capi import call
            self.sync_lr()
            call("xv2_adamw_step_dev", self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.flat_p.numel(),
                 self.lr_dev, float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                 self.step_dev, float(grad_scale))
            ops.weights_changed()
            ops.repack_all()       # the packed conv-weight layouts, refreshed in one launch
            return
        ops.adamw_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, lr, self.betas[0], self.betas[1],
                       self.eps, self.weight_decay, self.step_count, grad_scale)
        ops.weights_changed()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0]["lr"] = sd["lr"]
        if self.step_dev is not None:
            # the GPU kernel takes its bias correction from the device-resident counter
            self.step_dev.fill_(int(sd["step"]))
            self.sync_lr()
