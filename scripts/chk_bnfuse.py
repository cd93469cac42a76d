"""fused reduce+finalize vs the two-call path: every output must agree bit for bit"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xview2_amd import ops
torch.manual_seed(0)
dev = "cuda:0"
for (N, H, W, C0, Co, k) in [(4, 64, 64, 64, 128, 1), (2, 128, 128, 32, 32, 3), (4, 32, 32, 256, 512, 3)]:
    x = torch.randn(N, H, W, C0, device=dev)
    w = torch.randn(Co, C0, k, k, device=dev) * 0.1
    g = ops.conv_cfg(k, k, 1, k // 2)
    outs = []
    for fused in (True, False):
        bnm = torch.nn.BatchNorm2d(Co).to(dev)
        with torch.no_grad():
            bnm.weight.uniform_(0.5, 1.5); bnm.bias.normal_()
            torch.manual_seed(1); bnm.weight.uniform_(0.5, 1.5); bnm.bias.normal_()
        bn = ops.BnState(bnm)
        if fused:
            y, sums, co = ops._conv_forward(x, None, w, g, None, True, None, bn)
            z, st = ops._bn_forward(y, None, ops.ACT_RELU, bn, sums, True, co)
        else:
            y, sums = ops._conv_forward(x, None, w, g, None, True, None)
            z, st = ops._bn_forward(y, None, ops.ACT_RELU, bn, sums, True)
        torch.cuda.synchronize()
        outs.append((sums.clone(), st[0].clone(), st[1].clone(), st[3].clone(), st[4].clone(), bnm.running_mean.clone(), bnm.running_var.clone(), z.clone()))
    names = ["sums", "mean", "invstd", "scale", "shift", "rmean", "rvar", "z"]
    for n, a, b in zip(names, outs[0], outs[1]):
        print(n, torch.equal(a, b), (a.double() - b.double()).abs().max().item())
