"""F1 bookkeeping of the reference's utils/f1.py on device: label maps come from the HIP argmax kernel
(bit-exact torch.argmax: first maximum wins), tp/fp/fn are summed per class and all-reduced across ranks
(``dist_reduce_fx="sum"`` in the reference, utils/f1.py:24-26)."""
import torch
import torch.distributed as dist

from .. import ops


def convert_to_labels(loss_str, logits):  # utils/f1.py:7-15
    if loss_str == "mse":
        preds = torch.round(torch.relu(logits[:, 0])) + 1
        preds[preds > 4] = 4
        return preds
    if loss_str == "coral":
        return torch.sum(torch.sigmoid(logits) > 0.5, dim=1) + 1
    if not logits.is_cuda:      # host-side bookkeeping of already-downloaded logits
        return torch.argmax(logits, 1) + 1
    return ops.argmax_labels(logits, add=1).long()


class F1:
    def __init__(self, args):
        self.loss_str = args.loss_str
        self.n_class = 2 if args.type == "pre" else 5
        self.reset()

    def reset(self):
        self.tp = torch.zeros(self.n_class - 1, dtype=torch.float64)
        self.fp = torch.zeros(self.n_class - 1, dtype=torch.float64)
        self.fn = torch.zeros(self.n_class - 1, dtype=torch.float64)
        self.counts = None      # device int64 [(n_class-1)*3] (tp, fn, fp per class), filled by xv2_f1_counts

    def update(self, preds, targets):
        # softmax is monotone per pixel, so argmax(softmax(x)) == argmax(x) (utils/f1.py:29,36)
        if preds.is_cuda:
            # one counting launch on the label maps, no host round trip per class
            if self.n_class == 5:
                lab = convert_to_labels(self.loss_str, preds)
            else:
                lab = ops.argmax_labels(preds)
            lab = lab.to(torch.uint8).contiguous()
            tgt = targets.to(torch.uint8).contiguous()
            if self.counts is None:
                self.counts = torch.zeros((self.n_class - 1) * 3, dtype=torch.int64, device=preds.device)
            ops.f1_counts(lab, tgt, self.n_class, self.n_class == 5, self.counts)
            return
        targets = targets.long()
        if self.n_class == 5:
            lab = convert_to_labels(self.loss_str, preds)
            mask = targets > 0
            targets, lab = targets[mask], lab[mask]
        else:
            lab = torch.argmax(preds, 1)
        for i in range(self.n_class - 1):
            c = i + 1
            self.tp[i] += float(((lab == c) & (targets == c)).sum())
            self.fn[i] += float(((lab != c) & (targets == c)).sum())
            self.fp[i] += float(((lab == c) & (targets != c)).sum())

    def compute(self):
        tp, fp, fn = self.tp.clone(), self.fp.clone(), self.fn.clone()
        if self.counts is not None:
            c = self.counts.cpu().double().view(-1, 3)
            tp, fn, fp = tp + c[:, 0], fn + c[:, 1], fp + c[:, 2]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            buf = torch.stack([tp, fp, fn])
            if torch.cuda.is_available() and dist.get_backend() == "nccl":
                buf = buf.cuda()
            dist.all_reduce(buf)
            tp, fp, fn = buf.cpu()
        f1_score = (200 * tp / (2 * tp + fp + fn)).float()
        if self.n_class == 5:
            f1 = 4 / sum((f + 1e-6) ** -1 for f in f1_score)
            return f1, f1_score
        return f1_score, None
