import os, sys, socket, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.multiprocessing as mp
from tests.test_dist_gpu import _two_rank_worker


def main():
    from tests.golden.cases import ARGS, labels, model_input
    from xview2_amd import criterion, networks
    from xview2_amd.optim import FlatAdamW
    from xview2_amd.weights import deterministic_init_
    tmp = tempfile.mkdtemp()
    ctx = mp.get_context("spawn")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, tmp)) for r in range(2)]
    [p.start() for p in procs]; [p.join() for p in procs]
    res = [torch.load(os.path.join(tmp, "rank%d.pt" % r), weights_only=False) for r in range(2)]
    a = ARGS(encoder="resnet50", loss_str="ce", type="pre")
    x, y = model_input(a, batch=4).cuda(), labels(a, batch=4).cuda()
    torch.manual_seed(0)
    m = networks.UNetLoc(a); deterministic_init_(m, 1); m.cuda().train()
    opt = FlatAdamW(m.parameters(), lr=1e-3)
    opt.zero_grad()
    loss = criterion.Loss(a)(m(x), y); loss.backward()
    g = opt.flat_g.cpu().clone(); gr = res[0][3]
    names = [n for n, _ in m.named_parameters()]
    rows = []
    for (n, p), o in zip(m.named_parameters(), opt.offsets):
        a_, b_ = gr[o:o + p.numel()].double(), g[o:o + p.numel()].double()
        rows.append(((a_ - b_).abs().max().item(), b_.abs().max().item(), a_.abs().max().item(), n))
    rows.sort(reverse=True)
    for r in rows[:25]:
        print("absdiff %.3e  ref max %.3e  2rank max %.3e  %s" % r)
    print("params", len(rows), "with rel>1e-2:", sum(1 for r in rows if r[0] > 1e-2 * max(r[1], 1e-12)))


if __name__ == "__main__":
    main()
