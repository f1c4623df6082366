"""What a user of INTEGRATION.md recipe A gets: the reference's loop body (train_camus_echo.py:205-303: two FPN calls, seg
loss, score maps, GModule, four Discriminators, one backward) over modules imported under the reference's names, with
STOCK torch.optim.Adam / SGD -- next to the same step with this package's flat fused optimizers and with its trainer.
usage: bench_dropin_loop.py [frames per domain, default 8]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphecho_amd
graphecho_amd.install_as_reference_modules()
from models.fpnseg import FPN, Discriminator          # the reference's import lines
from models.graph_matching import GModule
from utils.losses import DiceLoss
from graphecho_amd.optim import FlatAdam, FlatSGD
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
xs, masks = synthetic_batch(nb, 3, 4, 256, dev, 1)
xt, _ = synthetic_batch(nb, 3, 4, 256, dev, 2)


def build(flat):
    torch.manual_seed(0)
    net = FPN([2, 4, 23, 3], 4, 3, back_bone="resnet").to(dev).train()
    gm = GModule(256, 4, dev).to(dev).train()
    dis = {n: Discriminator(grad_reverse_lambda=0.02).to(dev).train() for n in ("p2", "p3", "p4", "p5")}
    if flat:
        opts = [FlatAdam(net, lr=1e-4, weight_decay=1e-4)] + [FlatSGD(m, lr=8e-4, momentum=0.9, weight_decay=1e-4)
                                                             for m in [gm] + list(dis.values())]
    else:
        opts = [torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-4)]
        opts += [torch.optim.SGD(m.parameters(), lr=8e-4, momentum=0.9, weight_decay=1e-4) for m in [gm] + list(dis.values())]
    dice, bce = DiceLoss(), torch.nn.BCEWithLogitsLoss()
    losses = {}

    def step():
        pred_s, feat_s = net(xs)
        losses["seg_loss"] = 0.1 * (dice(pred_s, masks) + bce(pred_s, masks)) / 2
        pred_t, feat_t = net(xt)
        score = torch.where(torch.sigmoid(pred_t) > 0.5, 1, 0)
        (f_s, f_t), _n, mh = gm((xs, xt), (feat_s, feat_t), targets=masks, score_maps=score)
        losses.update(mh)
        for lvl, name in enumerate(("p2", "p3", "p4", "p5")):
            losses["loss_adv_" + name] = 0.1 * dis[name]((f_s[lvl], f_t[lvl]))
        for o in opts:
            o.zero_grad()
        sum(losses.values()).backward()
        for o in opts:
            o.step()
    return step


def timeit(step, n=20):
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print(f"{nb}+{nb} frames @256x256, ms per step:")
print(f"  reference loop body, stock torch.optim.Adam / SGD (recipe A as is): {timeit(build(False)):7.2f}")
print(f"  same loop, graphecho_amd.optim.FlatAdam / FlatSGD:                 {timeit(build(True)):7.2f}")
for g in (False, "auto"):
    tr = GraphEchoTrainer(dev, workload="full", seed=0, graphs=g)
    print(f"  graphecho_amd.trainer.GraphEchoTrainer(graphs={g!r}):{'':14s}{timeit(lambda: tr.step(xs, masks, xt)):7.2f}")
