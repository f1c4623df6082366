"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference, authoring container only).

Every fixture stores small probes of the reference's outputs for inputs/weights that are regenerated from
seeds (oracle.weights), so nothing of the reference itself is stored -- only numbers it computed.
Run:  PYTHONPATH=/root/repo python tools/gen_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import refimport  # noqa: E402

refimport.setup()
from oracle.weights import det_tensor, fill_state_dict, rect_masks  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v.detach() if torch.is_tensor(v) else v).shape) for k, v in arrs.items()})


def no_dropout(m):
    for s in m.modules():
        if isinstance(s, (nn.Dropout, nn.Dropout2d)):
            s.p = 0.0
    return m


def fpn_case(tag, bb, cin, nc, hw):
    from models.fpnseg import FPN
    from utils.losses import DiceLoss

    net = FPN([2, 4, 23, 3], nc, cin, back_bone=bb)
    net.load_state_dict(fill_state_dict(net.state_dict(), seed=1))
    net.train()
    x = det_tensor(f"{tag}.x", (2, cin, hw, hw), "uniform").requires_grad_(True)
    t = (det_tensor(f"{tag}.t", (2, nc, hw, hw), "uniform") > 0.6).float()
    logits, pyr = net(x)
    loss = DiceLoss()(logits, t) + nn.BCEWithLogitsLoss()(logits, t)
    loss.backward()
    sd = net.state_dict()
    bnkey = next(k for k in sd if k.endswith("running_mean"))
    save(f"fpn_{tag}", logits=logits[:, :, ::8, ::8], p5=pyr[3], p2_mean=pyr[0].mean((0, 2, 3)),
         p3_mean=pyr[1].mean((0, 2, 3)), p4_std=pyr[2].std((0, 2, 3)), loss=loss,
         g_smooth3=net.smooth3.weight.grad[:8, :8], g_conv3=net.conv3.weight.grad, g_x=x.grad[:, :, ::16, ::16],
         running_mean0=sd[bnkey], keys=np.array(list(sd.keys())),
         shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))


def discriminator_case():
    from models.fpnseg import Discriminator

    dis = Discriminator(grad_reverse_lambda=0.02)
    dis.load_state_dict(fill_state_dict(dis.state_dict(), seed=2))
    fs = det_tensor("dis.fs", (2, 256, 16, 16)).requires_grad_(True)
    ft = det_tensor("dis.ft", (2, 256, 16, 16)).requires_grad_(True)
    loss = dis((fs, ft))
    loss.backward()
    save("discriminator", loss=loss, g_fs=fs.grad[:, ::32, ::4, ::4], g_ft=ft.grad[:, ::32, ::4, ::4],
         g_cls=dis.cls_logits.weight.grad[0, :16], keys=np.array(list(dis.state_dict().keys())))


def knn_case():
    from models.vig import DenseDilatedKnnGraph

    out = {}
    for tag, (B, C, N, M, d) in {"n64_m64": (2, 256, 64, 64, 1), "self256": (2, 64, 256, None, 1),
                                 "n1024_m256_d2": (1, 128, 1024, 256, 2),
                                 "n4096_m256": (1, 256, 4096, 256, 1)}.items():     # Grapher r=4 on p2 (survey F3)
        x = det_tensor(f"knn.{tag}.x", (B, C, N, 1))
        y = None if M is None else det_tensor(f"knn.{tag}.y", (B, C, M, 1))
        idx = DenseDilatedKnnGraph(9, d)(x, y)
        # stable rows: every consecutive gap among the top-(k*d+1) distances exceeds 1e-5
        xn = F.normalize(x, dim=1)[..., 0].transpose(1, 2).double()
        yn = xn if y is None else F.normalize(y, dim=1)[..., 0].transpose(1, 2).double()
        dist = (xn * xn).sum(-1, keepdim=True) - 2 * xn @ yn.transpose(1, 2) + (yn * yn).sum(-1)[:, None, :]
        top = dist.topk(9 * d + 1, largest=False)[0]
        stable = (top[..., 1:] - top[..., :-1]).min(-1)[0] > 1e-5
        out[tag + "_idx"] = idx.numpy().astype(np.int16 if N > 2048 else np.int32)
        out[tag + "_stable"] = stable.numpy()
    save("knn", **out)


def grapher_case(only=None):
    from models.vig import Grapher

    for tag, (C, hw, r) in {"c64_r2": (64, 16, 2), "c256_r1": (256, 8, 1), "c256_r4_64": (256, 64, 4)}.items():
        if only and tag not in only:
            continue
        g = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.0, r, n=hw * hw)
        g.load_state_dict(fill_state_dict(g.state_dict(), seed=3))
        g.train()
        x = det_tensor(f"grapher.{tag}.x", (2, C, hw, hw)).requires_grad_(True)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            y = g(x)
        (y * det_tensor(f"grapher.{tag}.g", tuple(y.shape))).sum().backward()
        sp = 4 if hw >= 64 else 1     # the config-2 p2 block (survey F4): spatial stride keeps the fixture small
        save(f"grapher_{tag}", y=y[:, ::8, ::sp, ::sp], g_x=x.grad[:, ::8, ::sp, ::sp], g_fc1=g.fc1[0].weight.grad[:8, :8, 0, 0],
             g_gconv=g.graph_conv.gconv.nn[0].weight.grad[:8, :8, 0, 0], keys=np.array(list(g.state_dict().keys())))


def pvig_case():
    """Pyramid ViG (tiny) at its design size 224x224, B=2, 10 classes: logits + gradient probes."""
    import contextlib
    import io

    from models.vig import pvig_ti_224_gelu

    with contextlib.redirect_stdout(io.StringIO()):
        net = pvig_ti_224_gelu(num_classes=10)
    sd = net.state_dict()
    filled = fill_state_dict(sd, seed=5)
    for k_ in sd:                       # relative_pos is derived (frozen sin-cos table), not a learned weight
        if "relative_pos" in k_:
            filled[k_] = sd[k_].clone()
    net.load_state_dict(filled)
    net.train()
    x = det_tensor("pvig.x", (2, 3, 224, 224), "uniform").requires_grad_(True)
    with contextlib.redirect_stdout(io.StringIO()):
        y = net(x)
    (y * det_tensor("pvig.g", tuple(y.shape))).sum().backward()
    b0, bl = net.backbone[0], net.backbone[-1]
    save("pvig_ti", y=y, g_x=x.grad[:, :, ::16, ::16], g_stem=net.stem.convs[0].weight.grad[:8],
         g_pos=net.pos_embed.grad[0, :8, ::8, ::8], g_fc1_first=b0[0].fc1[0].weight.grad[:8, :8, 0, 0],
         g_gconv_last=bl[0].graph_conv.gconv.nn[0].weight.grad[:8, :8, 0, 0],
         rel_first=b0[0].relative_pos[0, ::64, ::16], rel_last=bl[0].relative_pos[0, ::7, ::7],
         keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))


GRAPHCONV_CASES = {  # tag: (conv, act, norm, C_in, C_out, N, M or None)
    "edge_relu_batch": ("edge", "relu", "batch", 32, 64, 49, None),
    "edge_leaky_batch_xy": ("edge", "leakyrelu", "batch", 32, 48, 50, 16),   # norm="instance" cannot be built by
    # the reference itself: BasicConv.reset_parameters (vig.py:497-500) dereferences the affine=False norm's weight
    "sage_prelu_batch_xy": ("sage", "prelu", "batch", 32, 64, 50, 16),
    "gin_hswish_none": ("gin", "hswish", None, 32, 64, 49, None),
    "mr_relu_none_xy": ("mr", "relu", None, 32, 64, 50, 16),
}


def graphconv_case():
    """GraphConv2d with every aggregator / activation / norm the reference offers (vig.py:88-181,433-500)."""
    from models.vig import DenseDilatedKnnGraph, GraphConv2d

    out = {}
    for tag, (conv, act, norm, ci, co, N, M) in GRAPHCONV_CASES.items():
        g = GraphConv2d(ci, co, conv, act, norm, True)
        g.load_state_dict(fill_state_dict(g.state_dict(), seed=7))
        g.train()
        x = det_tensor(f"gconv.{tag}.x", (2, ci, N, 1)).requires_grad_(True)
        y = det_tensor(f"gconv.{tag}.y", (2, ci, M, 1)).requires_grad_(True) if M else None
        with torch.no_grad():
            edge = DenseDilatedKnnGraph(9, 1)(x, y)
        o = g(x, edge, y)
        (o * det_tensor(f"gconv.{tag}.g", tuple(o.shape))).sum().backward()
        first = [k for k in g.state_dict() if k.endswith("0.weight")][0]
        mod = g.gconv.nn1[0] if conv == "sage" else g.gconv.nn[0]
        out.update({f"{tag}.edge": edge, f"{tag}.out": o, f"{tag}.g_x": x.grad, f"{tag}.g_w": mod.weight.grad,
                    f"{tag}.keys": np.array(list(g.state_dict().keys()))})
        if y is not None:
            out[f"{tag}.g_y"] = y.grad
        if conv == "gin":
            out[f"{tag}.g_eps"] = g.gconv.eps.grad
    save("graphconv", **out)


def small_ops_case():
    from models.transformer import MultiHeadAttention
    from models.affinity_layer import Affinity
    from models.graph_matching import GModule
    from utils.sinkhorn_distance import SinkhornDistance

    mha = MultiHeadAttention(256, 1, dropout=0.0, version="v2")
    mha.load_state_dict(fill_state_dict(mha.state_dict(), seed=4))
    kv, q = det_tensor("mha.kv", (70, 256)), det_tensor("mha.q", (50, 256))
    o, a = mha(kv, kv, q)
    aff = Affinity(256)
    aff.load_state_dict(fill_state_dict(aff.state_dict(), seed=5))
    X, Y = det_tensor("aff.x", (37, 256)), det_tensor("aff.y", (45, 256))
    M = aff(X, Y)
    gm = GModule(256, 4, "cpu")
    la = det_tensor("rpm.a", (1, 60, 75))
    rpm = gm.sinkhorn_rpm(la, n_iters=20).exp()
    sd = SinkhornDistance(eps=0.1, max_iter=5, reduction="mean")
    x3, y3 = det_tensor("sd.x", (2, 64, 256), "uniform"), det_tensor("sd.y", (2, 64, 256), "uniform")
    c3, p3, C3 = sd(x3, y3)
    x2, y2 = det_tensor("sd2.x", (64, 32), "uniform"), det_tensor("sd2.y", (50, 32), "uniform")
    c2, p2, C2 = sd(x2, y2)
    save("small_ops", mha_out=o, mha_att=a[::5, ::5], aff_M=M, rpm=rpm[0, ::3, ::3], sd3_cost=c3, sd3_pi=p3[:, ::4, ::4],
         sd3_C=C3[:, ::4, ::4], sd2_cost=c2, sd2_pi=p2[::4, ::4],
         mha_keys=np.array(list(mha.state_dict().keys())), aff_keys=np.array(list(aff.state_dict().keys())))


def gmodule_case():
    from models.graph_matching import GModule

    import contextlib, io
    for cluster in (False, True):
        with contextlib.redirect_stdout(io.StringIO()):
            gm = GModule(256, 4, "cpu")
        gm.load_state_dict(fill_state_dict(gm.state_dict(), seed=6))
        no_dropout(gm)
        gm.train()
        gm.with_cluster_update = cluster
        sizes = (64, 32, 16, 8)
        fs = [det_tensor(f"gm.fs{l}", (2, 256, s, s)).requires_grad_(True) for l, s in enumerate(sizes)]
        ft = [det_tensor(f"gm.ft{l}", (2, 256, s, s)).requires_grad_(True) for l, s in enumerate(sizes)]
        tgt, sm = rect_masks(2, 4, 256, 256, seed=1), rect_masks(2, 4, 256, 256, seed=2)
        _, (n1, n2), losses = gm(None, (fs, ft), targets=tgt, score_maps=sm)
        sum(losses.values()).backward()
        save(f"gmodule_cluster{int(cluster)}", n1=n1[::7, ::16], n2=n2[::7, ::16], n_nodes=np.array([len(n1), len(n2)]),
             **{k: v for k, v in losses.items()}, sr_seed=gm.sr_seed, tg_seed=gm.tg_seed,
             g_fs0=fs[0].grad[:, ::32, ::8, ::8], g_aff=gm.node_affinity.fc_M[0].weight.grad[:8, :8],
             keys=np.array(list(gm.state_dict().keys())))


def tgcn_case():
    from models.TGCN import TGCN
    from utils.sinkhorn_distance import SinkhornDistance

    for method in ("node_discriminate", "sinkhorn_distance"):
        m = TGCN(256, 256, (3, 8, 8), 10, 10, transport_method=method)
        m.load_state_dict(fill_state_dict(m.state_dict(), seed=7))
        no_dropout(m)
        m.train()
        b, t = 2, 3
        feats = [det_tensor(f"tgcn.f{l}", (b, t, 256, s, s)) for l, s in enumerate((64, 32, 16, 8))]
        nodes = (det_tensor("tgcn.ns", (33, 256)), det_tensor("tgcn.nt", (34, 256)))
        sk = SinkhornDistance(eps=0.1, max_iter=5, reduction="mean")
        upd = (torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long))
        # capture current_graph through a forward hook on the grapher
        graphs = []
        h = m.grapher.register_forward_hook(lambda mod, i, o: graphs.append(o[0].detach()))
        losses = m(feats, nodes, sk, nn.CrossEntropyLoss(), upd, r=[8, 4, 2, 1])
        h.remove()
        sum(losses.values()).backward()
        save(f"tgcn_{method}", graph=graphs[-1][:, ::16, ::4], graph0=graphs[0][:, ::16, ::4],
             **{k: v for k, v in losses.items()}, g_pos=m.pos_embed.grad[:, 0, ::32], 
             g_mlp=m.grapher.MLP[0].weight.grad[:8, :8, 0, 0], keys=np.array(list(m.state_dict().keys())))


def step_case(tag, nb, hw, seg="cardiac"):
    """Survey F8: optimisation steps of the full GraphEcho loop composed from the reference's own modules exactly as
    train_camus_echo.py:205-303 / train_cardiac_uda.py:222-320 compose them (the scripts themselves import MONAI / cv2
    at module top and cannot be imported): FPN on source and target frames, seg loss, score maps, GModule, four
    Discriminators x 0.1, one backward, Adam(FPN) / SGD-momentum(others) at the reference's effective lr / 3, twice."""
    import contextlib
    import io

    from models.fpnseg import FPN, Discriminator
    from models.graph_matching import GModule
    from utils.losses import DiceLoss

    nc = 4
    net = FPN([2, 4, 23, 3], nc, 3, back_bone="resnet")
    net.load_state_dict(fill_state_dict(net.state_dict(), seed=1))
    with contextlib.redirect_stdout(io.StringIO()):
        gm = GModule(256, nc, "cpu")
    gm.load_state_dict(fill_state_dict(gm.state_dict(), seed=6))
    no_dropout(gm)
    dis = {}
    for i, name in enumerate(("p2", "p3", "p4", "p5")):
        dis[name] = Discriminator(grad_reverse_lambda=0.02)
        dis[name].load_state_dict(fill_state_dict(dis[name].state_dict(), seed=20 + i))
    for m in [net, gm] + list(dis.values()):
        m.train()
    xs = det_tensor(f"step.{tag}.xs", (nb, 3, hw, hw), "uniform")
    xt = det_tensor(f"step.{tag}.xt", (nb, 3, hw, hw), "uniform")
    masks = rect_masks(nb, nc, hw, hw, seed=3)
    opts = [torch.optim.Adam(net.parameters(), lr=3e-4 / 3, weight_decay=1e-4)]
    opts += [torch.optim.SGD(m.parameters(), lr=0.0025 / 3, momentum=0.9, weight_decay=1e-4) for m in [gm] + list(dis.values())]
    dice, bce = DiceLoss(), nn.BCEWithLogitsLoss(reduction="mean")
    # Pseudo-labels of a hash-filled network rarely contain every class, so the hallucination branch
    # (graph_matching.py:432-472) runs.  Its torch.normal draws are replaced by a named deterministic stream
    # (standard-normal tensor number i = det_tensor(f"noise.{i}")), which the HIP module consumes through
    # GModule.noise_fn: same code path of the reference, reproducible noise.
    noise_calls = []
    real_normal = torch.normal

    def det_normal(mean=0.0, std=1.0, size=None, **kw):
        shape = tuple(size) if size is not None else tuple(mean.shape if torch.is_tensor(mean) else std.shape)
        eps = det_tensor(f"noise.{len(noise_calls)}", shape)
        noise_calls.append(shape)
        return mean + std * eps

    torch.normal = det_normal
    out, losses = {}, {}
    try:
        for step in range(2):
            pred_s, feat_s = net(xs)
            d, b = dice(pred_s, masks), bce(pred_s, masks)
            losses["seg_loss"] = d + b if seg == "cardiac" else 0.1 * (d + b) / 2
            pred_t, feat_t = net(xt)
            score = torch.where(nn.Sigmoid()(pred_t) > 0.5, 1, 0)
            with contextlib.redirect_stdout(io.StringIO()):
                (f_s, f_t), (n1, n2), mh = gm((xs, xt), (feat_s, feat_t), targets=masks, score_maps=score)
            losses.update(mh)
            for l, name in enumerate(("p2", "p3", "p4", "p5")):
                losses["loss_adv_" + name] = 0.1 * dis[name]((f_s[l], f_t[l]))
            for o in opts:
                o.zero_grad()
            total = sum(losses.values())
            total.backward()
            if step == 0:
                out.update(g_conv3=net.conv3.weight.grad.clone(), g_top=net.toplayer.weight.grad[:8, :8, 0, 0].clone(),
                           g_dis_p3=dis["p3"].cls_logits.weight.grad[0, :16].clone(),
                           g_gm=gm.node_affinity.fc_M[0].weight.grad[:8, :8].clone(),
                           score_frac=score.float().mean((0, 2, 3)), logits_t=pred_t[:, :, ::16, ::16].detach().clone(),
                           n_nodes=np.array([len(n1), len(n2)]))
            for o in opts:
                o.step()
            out.update({f"s{step}.{k}": v.detach().clone() for k, v in losses.items()})
            out[f"s{step}.total"] = total.detach().clone()
            out[f"s{step}.sr_seed"], out[f"s{step}.tg_seed"] = gm.sr_seed.clone(), gm.tg_seed.clone()
    finally:
        torch.normal = real_normal
    print("hallucination draws:", noise_calls)
    sd = net.state_dict()
    bnkey = next(k for k in sd if k.endswith("running_mean"))
    save(f"step_c3_{tag}", **out, conv3_after=sd["conv3.weight"], running_mean0=sd[bnkey], sr_seed=gm.sr_seed, tg_seed=gm.tg_seed,
         loss_keys=np.array(list(losses.keys())), noise_draws=np.array(len(noise_calls)))


def temporal_case(tag="c5", nb=2, hw=256, t=16):
    """BASELINE config 5 as the reference runs it (train_cardiac_uda.py:73,222-320): FPN(in_channel=1, back_bone="VGG16"),
    seg loss Dice + BCE over all channels, GModule + four Discriminators on a source / target frame batch, then the
    temporal branch -- one source + one target clip of `t` frames folded into the batch, FPN, GModule on the clip
    features, TGCN(transport_method='sinkhorn_distance') with SinkhornDistance(eps 0.1, 5 iterations) -- ONE backward,
    Adam(FPN) / SGD(others).  Dropout 0, hallucination noise from the named deterministic stream (as step_case).
    Stored: every loss term of the step (incl. the TGCN's and the second GModule call's own terms), the Sinkhorn cost /
    plan / cost-matrix probes and stopping iteration, logits, gradient probes, weights after the step."""
    import contextlib
    import io

    from models.fpnseg import FPN, Discriminator
    from models.graph_matching import GModule
    from models.TGCN import TGCN
    from utils.losses import DiceLoss
    from utils.sinkhorn_distance import SinkhornDistance

    nc = 4
    net = FPN([2, 4, 23, 3], nc, 1, back_bone="VGG16")
    net.load_state_dict(fill_state_dict(net.state_dict(), seed=1))
    with contextlib.redirect_stdout(io.StringIO()):
        gm = GModule(256, nc, "cpu")
    gm.load_state_dict(fill_state_dict(gm.state_dict(), seed=6))
    no_dropout(gm)
    dis = {}
    for i, name in enumerate(("p2", "p3", "p4", "p5")):
        dis[name] = Discriminator(grad_reverse_lambda=0.02)
        dis[name].load_state_dict(fill_state_dict(dis[name].state_dict(), seed=20 + i))
    tg = TGCN(256, 256, (t, hw // 32, hw // 32), 10, 10, transport_method="sinkhorn_distance")
    tg.load_state_dict(fill_state_dict(tg.state_dict(), seed=7))
    no_dropout(tg)
    for m in [net, gm, tg] + list(dis.values()):
        m.train()
    xs = det_tensor(f"temporal.{tag}.xs", (nb, 1, hw, hw), "uniform")
    xt = det_tensor(f"temporal.{tag}.xt", (nb, 1, hw, hw), "uniform")
    masks = rect_masks(nb, nc, hw, hw, seed=3)
    cs = det_tensor(f"temporal.{tag}.cs", (1, 1, hw, hw, t), "uniform")
    ct = det_tensor(f"temporal.{tag}.ct", (1, 1, hw, hw, t), "uniform")
    cm = rect_masks(t, nc, hw, hw, seed=5).permute(1, 2, 3, 0).unsqueeze(0).contiguous()      # (1, nc, H, W, T)
    cm[..., 1::4] = 0          # sparsely annotated clip: three of four frames carry labels, the others hand the prediction on
    opts = [torch.optim.Adam(net.parameters(), lr=3e-4 / 3, weight_decay=1e-4)]
    opts += [torch.optim.SGD(m.parameters(), lr=0.0025 / 3, momentum=0.9, weight_decay=1e-4)
             for m in [gm, tg] + list(dis.values())]
    dice, bce, ce = DiceLoss(), nn.BCEWithLogitsLoss(reduction="mean"), nn.CrossEntropyLoss()
    sk = SinkhornDistance(eps=0.1, max_iter=5, reduction="mean")
    sk_out = []
    real_fwd = sk.forward

    def rec_fwd(x, y):
        out = real_fwd(x, y)
        sk_out.append([o.detach().clone() for o in out])
        return out

    sk.forward = rec_fwd
    noise_calls = []
    real_normal = torch.normal

    def det_normal(mean=0.0, std=1.0, size=None, **kw):
        shape = tuple(size) if size is not None else tuple(mean.shape if torch.is_tensor(mean) else std.shape)
        eps = det_tensor(f"noise.{len(noise_calls)}", shape)
        noise_calls.append(shape)
        return mean + std * eps

    torch.normal = det_normal
    losses, out = {}, {}
    try:
        pred_s, feat_s = net(xs)
        losses["seg_loss"] = dice(pred_s, masks) + bce(pred_s, masks)
        pred_t, feat_t = net(xt)
        score = torch.where(nn.Sigmoid()(pred_t) > 0.5, 1, 0)
        with contextlib.redirect_stdout(io.StringIO()):
            (f_s, f_t), _, mh = gm((xs, xt), (feat_s, feat_t), targets=masks, score_maps=score)
        losses.update(mh)
        for l, name in enumerate(("p2", "p3", "p4", "p5")):
            losses["loss_adv_" + name] = 0.1 * dis[name]((f_s[l], f_t[l]))
        for o in opts:
            o.zero_grad()
        # ---- temporal branch (train_cardiac_uda.py:258-312) ----
        imgs_temp = torch.cat([cs, ct], dim=0)
        b, c, h, w, tt = imgs_temp.shape
        imgs_temp = imgs_temp.permute(0, 4, 1, 2, 3).reshape(-1, c, h, w)
        src_masks = cm.permute(0, 4, 1, 2, 3).reshape(b * tt // 2, -1, h, w) / 1.0
        sel = torch.where(torch.sum(src_masks, dim=(1, 2, 3)) > 100, 1, 0)
        preds_, features_ = net(imgs_temp)
        pst = preds_[:b * tt // 2]
        sm = torch.cat([src_masks[i].unsqueeze(0) if ok else pst[i].unsqueeze(0) for i, ok in enumerate(sel)], dim=0)
        sf = [f[:f.shape[0] // 2] for f in features_]
        tf = [f[f.shape[0] // 2:] for f in features_]
        with contextlib.redirect_stdout(io.StringIO()):
            (_, _), (sn, tn), tmh = gm((imgs_temp[:b * tt // 2], imgs_temp[b * tt // 2:]), (sf, tf), targets=sm,
                                       score_maps=preds_[b * tt // 2:])
        gfeat = [f.reshape(b, -1, f.shape[1], f.shape[2], f.shape[3]) for f in features_]
        upd = (torch.zeros(b // 2, dtype=torch.long), torch.zeros(b // 2, dtype=torch.long))
        tgl = tg(gfeat, (sn.clone().detach(), tn.clone().detach()), sk, ce, upd, r=[8, 4, 2, 1])
        losses["temporal_graph_loss"] = sum(tgl.values()) + sum(tmh.values())
        total = sum(losses.values())
        total.backward()
        out.update(g_conv3=net.conv3.weight.grad.clone(), g_top=net.toplayer.weight.grad[:8, :8, 0, 0].clone(),
                   g_vgg0=net.back_bone.block_1[0].weight.grad[:8, 0].clone(),
                   g_tgcn_mlp=tg.grapher.MLP[0].weight.grad[:8, :8, 0, 0].clone(),
                   g_gm=gm.node_affinity.fc_M[0].weight.grad[:8, :8].clone(),
                   logits_s=pred_s[:, :, ::16, ::16].detach().clone(), logits_clip=preds_[::4, :, ::16, ::16].detach().clone(),
                   labelled=sel.clone(), n_nodes=np.array([len(sn), len(tn)]))
        for o in opts:
            o.step()
    finally:
        torch.normal = real_normal
    print("hallucination draws:", noise_calls, "sinkhorn calls:", len(sk_out))
    out.update({k: v.detach().clone() for k, v in losses.items()})
    out.update({"tgcn." + k: v.detach().clone() for k, v in tgl.items()})
    out.update({"clipgm." + k: v.detach().clone() for k, v in tmh.items()})
    cost, pi, C = sk_out[-1]
    sd = net.state_dict()
    bnkey = next(k for k in sd if k.endswith("running_mean"))
    save(f"temporal_{tag}", **out, total=total.detach(), sk_cost=cost, sk_pi=pi[..., ::4, ::4], sk_C=C[..., ::4, ::4],
         sk_pi_sum=pi.sum(), conv3_after=sd["conv3.weight"], running_mean0=sd[bnkey], sr_seed=gm.sr_seed, tg_seed=gm.tg_seed,
         loss_keys=np.array(list(losses.keys())), tgcn_keys=np.array(list(tgl.keys())), clipgm_keys=np.array(list(tmh.keys())),
         noise_draws=np.array(len(noise_calls)), sk_calls=np.array(len(sk_out)))


def edge_case():
    """Reference branches no trainer configuration reaches by default: MultiHeadAttention v1 / v2
    with 4 heads (transformer.py:25-110), CrossGraph (:115-160), stochastic dilation under a fixed torch RNG seed
    (vig.py:332-354), the chunked > 10 000-point k-NN (vig.py:277-309), GModule's `< 6 source nodes` early return
    (graph_matching.py:258-260)."""
    import contextlib
    import io

    from models.graph_matching import GModule
    from models.transformer import CrossGraph, MultiHeadAttention
    from models.vig import DenseDilatedKnnGraph

    out = {}
    kv, q = det_tensor("edge.mha.kv", (70, 256)), det_tensor("edge.mha.q", (50, 256))
    for ver in ("v1", "v2"):
        mha = MultiHeadAttention(256, 4, dropout=0.0, version=ver)
        mha.load_state_dict(fill_state_dict(mha.state_dict(), seed=11))
        kk, qq = kv.clone().requires_grad_(True), q.clone().requires_grad_(True)
        # v1 reshapes (1, N, 256) buffers into 4 "heads" of N rows: it needs key and query of equal length
        o, a = mha(kk, kk, kk if ver == "v1" else qq)
        (o * det_tensor(f"edge.mha.g.{ver}", tuple(o.shape))).sum().backward()
        out.update({f"mha_{ver}_out": o, f"mha_{ver}_att": a[:, ::5, ::5], f"mha_{ver}_g_kv": kk.grad[::5, ::16],
                    f"mha_{ver}_g_wq": mha.linear_q.weight.grad[:8, :8]})
    cg = CrossGraph(256, 0.0)
    cg.load_state_dict(fill_state_dict(cg.state_dict(), seed=12))
    n1, n2 = det_tensor("edge.cg.n1", (37, 256)), det_tensor("edge.cg.n2", (45, 256))
    o1, o2 = cg(n1, n2)
    out.update(cg_o1=o1, cg_o2=o2, cg_keys=np.array(list(cg.state_dict().keys())))
    # stochastic dilation: epsilon = 1 -> the random branch is always taken in train mode; same seed, same picks
    g = DenseDilatedKnnGraph(9, 2, stochastic=True, epsilon=1.0).train()
    x = det_tensor("edge.sto.x", (2, 64, 196, 1))
    torch.manual_seed(777)
    out["sto_idx"] = g(x).numpy().astype(np.int16)
    g.eval()
    out["sto_idx_eval"] = g(x).numpy().astype(np.int16)
    # chunked k-NN: 10 050 points (> n_part = 10 000 -> two chunks), self graph
    xb = det_tensor("edge.big.x", (1, 16, 10050, 1))
    idx = DenseDilatedKnnGraph(9, 1)(xb)
    xn = F.normalize(xb, dim=1)[0, :, :, 0].t().double()
    dist = (xn * xn).sum(-1, keepdim=True) - 2 * xn @ xn.t() + (xn * xn).sum(-1)[None]
    top = dist.topk(10, largest=False)[0]
    out["big_idx"] = idx[0, 0].numpy().astype(np.int16)
    out["big_stable"] = ((top[:, 1:] - top[:, :-1]).min(-1)[0] > 1e-5).numpy()
    # < 6 source nodes: 2 x 2 pixel masks contain no sampling location
    with contextlib.redirect_stdout(io.StringIO()):
        gm = GModule(256, 4, "cpu")
    gm.load_state_dict(fill_state_dict(gm.state_dict(), seed=6))
    gm.train()
    sizes = (64, 32, 16, 8)
    fs = [det_tensor(f"gm.fs{l}", (2, 256, s, s)) for l, s in enumerate(sizes)]
    ft = [det_tensor(f"gm.ft{l}", (2, 256, s, s)) for l, s in enumerate(sizes)]
    tgt = torch.zeros(2, 4, 256, 256)
    for c in range(4):
        tgt[:, c, 10 + c * 20:12 + c * 20, 10:12] = 1
    feats, (e1, e2), losses = gm(None, (fs, ft), targets=tgt, score_maps=rect_masks(2, 4, 256, 256, seed=2))
    assert len(losses) == 0 and feats[0][0] is fs[0]
    out.update(few_n=np.array([e1.shape[0], e2.shape[0]]), few_n2=e2[::7, ::16], few_sr_seed=gm.sr_seed)
    save("edge_branches", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fpn", "dis", "knn", "grapher", "pvig", "graphconv", "small", "gmodule", "tgcn", "step", "edge"]
    if "fpn" in which:
        fpn_case("resnet_c3_n4_128", "resnet", 3, 4, 128)
        fpn_case("vgg_c1_n1_128", "VGG16", 1, 1, 128)
        fpn_case("resnet_c1_n3_256", "resnet", 1, 3, 256)
    if "fpn" in which or "fpn256" in which:
        fpn_case("resnet_c3_n4_256", "resnet", 3, 4, 256)     # BASELINE config 1 exactly: 2 x 3 x 256 x 256, 4 classes
    if "dis" in which:
        discriminator_case()
    if "knn" in which:
        knn_case()
    if "grapher" in which:
        grapher_case()
    if "grapher64" in which:
        grapher_case(only=("c256_r4_64",))
    if "edge" in which:
        edge_case()
    if "step" in which or "step128" in which:
        step_case("128", 2, 128)
    if "step" in which or "step256" in which:
        step_case("256", 8, 256)                               # BASELINE config 3 exactly: source 8 + target 8 @256
    if "pvig" in which:
        pvig_case()
    if "graphconv" in which:
        graphconv_case()
    if "small" in which:
        small_ops_case()
    if "gmodule" in which:
        gmodule_case()
    if "tgcn" in which:
        tgcn_case()
    if "temporal" in which:
        temporal_case()                                        # BASELINE config 5 as train_cardiac_uda.py runs it (VGG16, 1 ch)
