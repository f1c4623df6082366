"""Oracle: whole training steps on the CPU (torch ops), used (a) to check the HIP trainer's loss trajectory and
(b) as bench.py's cpu_baseline ("port": the reference's algorithm restated, timed on the host cores).
TEST INFRASTRUCTURE ONLY.
"""
import torch
import torch.nn.functional as F

from . import vig as ovig
from .fpn import fpn_forward
from .misc import seg_loss_camus, seg_loss_cardiac


def knn_torch(x, y, k, dilation=1):
    """Reference-style k-NN (vig.py:262-329, 369-381): normalise, ||x||^2 - 2xy + ||y||^2, topk.  Fast (MKL),
    tie order unspecified -- used only where speed matters (CPU baseline timing)."""
    with torch.no_grad():
        xn = F.normalize(x, p=2.0, dim=1)
        yn = xn if y is None else F.normalize(y, p=2.0, dim=1)
        a, b = xn.transpose(2, 1).squeeze(-1), yn.transpose(2, 1).squeeze(-1)
        dist = (a * a).sum(-1, keepdim=True) + (-2 * a @ b.transpose(2, 1)) + (b * b).sum(-1, keepdim=True).transpose(2, 1)
        idx = torch.topk(-dist, k=k * dilation)[1]
        ctr = torch.arange(a.shape[1]).repeat(a.shape[0], k * dilation, 1).transpose(2, 1)
        return torch.stack((idx, ctr), dim=0)[:, :, :, ::dilation]


def fpn_grapher_loss(fpn_params, grapher_params, x, masks, seg="camus", exact_knn=True, ratios=(4, 2, 1, 1)):
    """Loss of the config-2 harness: seg loss + 0.01 * sum_l mean(Grapher_l(p_l)^2)."""
    logits, pyr = fpn_forward(fpn_params, x, True)
    loss = (seg_loss_camus if seg == "camus" else seg_loss_cardiac)(logits, masks)
    if grapher_params is not None:
        saved = ovig.edge_index
        if not exact_knn:
            ovig.edge_index = lambda xx, yy, k, d, rp=None: knn_torch(xx, yy, k, d)
        try:
            for l, (p, r) in enumerate(zip(pyr, ratios)):
                sd = {k[len(f"blocks.{l}."):]: v for k, v in grapher_params.items() if k.startswith(f"blocks.{l}.")}
                out = ovig.grapher_forward(sd, "", p, 9, 1, r, "gelu", True, True)
                loss = loss + 0.01 * (out * out).mean()
        finally:
            ovig.edge_index = saved
    return loss, logits


class CpuTrainer:
    """fwd + loss + bwd + Adam(FPN) / SGD(Graphers), mirroring graphecho_amd.trainer for 'fpn' / 'fpn_grapher'."""

    def __init__(self, fpn_sd, grapher_sd=None, seg="camus", exact_knn=False):
        self.seg, self.exact_knn = seg, exact_knn
        self.fpn = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
                    for k, v in fpn_sd.items()}
        self.gr = None
        if grapher_sd is not None:
            self.gr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
                           else v.clone()) for k, v in grapher_sd.items()}
        # effective LRs of the reference: WarmupMultiStepLR('constant', factor 1/3, 1000 "iters") is stepped per
        # epoch, so every optimizer runs at base_lr / 3 (train_camus_echo.py:312-313,565-626)
        self.opt = torch.optim.Adam([p for p in self.fpn.values() if p.requires_grad], lr=3e-4 / 3, weight_decay=1e-4)
        self.opt2 = None
        if self.gr is not None:
            self.opt2 = torch.optim.SGD([p for p in self.gr.values() if p.requires_grad], lr=0.0025 / 3, momentum=0.9,
                                        weight_decay=1e-4)

    def step(self, x, masks):
        self.opt.zero_grad()
        if self.opt2:
            self.opt2.zero_grad()
        loss, logits = fpn_grapher_loss(self.fpn, self.gr, x, masks, self.seg, self.exact_knn)
        loss.backward()
        self.opt.step()
        if self.opt2:
            self.opt2.step()
        return loss.detach(), logits.detach()


class FullCpuTrainer:
    """The full GraphEcho step (BASELINE config 3) on the CPU: FPN on source and target frames, seg loss, score maps,
    GModule, four Discriminators x 0.1, one backward, Adam(FPN) / SGD-momentum 0.9 (others) at lr / 3, weight decay
    1e-4 -- train_camus_echo.py:205-303, 425-445 (train_cardiac_uda.py:222-320 for the 'cardiac' seg form).  Pinned
    by tests/golden/step_c3_*.npz, which the reference's own modules produced (tools/gen_golden.py step_case)."""

    def __init__(self, fpn_sd, gm_sd, dis_sds, seg="cardiac", num_class=4, with_cluster=True, noise_fn=None):
        grad = lambda sd, skip=(): {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
                                        and k not in skip else v.clone()) for k, v in sd.items()}
        self.fpn = grad(fpn_sd)
        self.gm = grad(gm_sd, ("sr_seed", "tg_seed"))
        self.dis = {k: grad(v) for k, v in dis_sds.items()}
        self.seg, self.nc, self.with_cluster, self.noise_fn = seg, num_class, with_cluster, noise_fn
        req = lambda d: [p for p in d.values() if p.requires_grad]
        self.opts = [torch.optim.Adam(req(self.fpn), lr=3e-4 / 3, weight_decay=1e-4)]
        self.opts += [torch.optim.SGD(req(d), lr=0.0025 / 3, momentum=0.9, weight_decay=1e-4)
                      for d in [self.gm] + list(self.dis.values())]
        self.losses = {}      # persists across steps like the reference's dict (train_camus_echo.py:185)

    def step(self, xs, masks, xt):
        from . import fpn as ofpn
        from .fpn import discriminator_forward
        from .gmodule import gmodule_forward

        losses = self.losses
        ofpn.UPDATE_RUNNING = True
        try:
            pred_s, feat_s = fpn_forward(self.fpn, xs, True)
            pred_t, feat_t = fpn_forward(self.fpn, xt, True)
        finally:
            ofpn.UPDATE_RUNNING = False
        losses["seg_loss"] = (seg_loss_camus if self.seg == "camus" else seg_loss_cardiac)(pred_s, masks)
        score = (torch.sigmoid(pred_t) > 0.5).float()
        n1, n2, gl, seeds, counts = gmodule_forward(self.gm, (feat_s, feat_t), masks, score, self.nc, self.with_cluster,
                                                    self.noise_fn)
        losses.update(gl)
        with torch.no_grad():
            self.gm["sr_seed"], self.gm["tg_seed"] = seeds[0].clone(), seeds[1].clone()
        for l, name in enumerate(("p2", "p3", "p4", "p5")):
            losses["loss_adv_" + name] = 0.1 * discriminator_forward(self.dis[name], (feat_s[l], feat_t[l]), 0.02)
        for o in self.opts:
            o.zero_grad()
        total = sum(losses.values())
        total.backward()
        for o in self.opts:
            o.step()
        return total.detach(), {k: v.detach() for k, v in losses.items()}, pred_t.detach(), counts


class TemporalCpuTrainer(FullCpuTrainer):
    """BASELINE config 5 as the reference runs it (train_cardiac_uda.py:222-320): the full step above plus the temporal
    branch -- source + target clips folded into the batch, FPN, GModule on the clip features (frames whose label map has
    <= 100 pixels hand their prediction on as the target, :279-290), TGCN with the SinkhornDistance transport loss --
    one backward, Adam(FPN) / SGD(others).  Pinned by tests/golden/temporal_c5.npz (tools/gen_golden.py temporal_case)."""

    def __init__(self, fpn_sd, gm_sd, dis_sds, tgcn_sd, seg="cardiac", num_class=4, with_cluster=True, noise_fn=None):
        super().__init__(fpn_sd, gm_sd, dis_sds, seg, num_class, with_cluster, noise_fn)
        self.tg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
                   for k, v in tgcn_sd.items()}
        self.opts.append(torch.optim.SGD([p for p in self.tg.values() if p.requires_grad], lr=0.0025 / 3, momentum=0.9,
                                         weight_decay=1e-4))

    def step(self, xs, masks, xt, clips):
        from . import fpn as ofpn
        from .fpn import discriminator_forward
        from .gmodule import gmodule_forward
        from .tgcn import tgcn_forward

        losses = self.losses
        ofpn.UPDATE_RUNNING = True
        try:
            pred_s, feat_s = fpn_forward(self.fpn, xs, True)
            pred_t, feat_t = fpn_forward(self.fpn, xt, True)
            losses["seg_loss"] = (seg_loss_camus if self.seg == "camus" else seg_loss_cardiac)(pred_s, masks)
            score = (torch.sigmoid(pred_t) > 0.5).float()
            _, _, gl, seeds, _ = gmodule_forward(self.gm, (feat_s, feat_t), masks, score, self.nc, self.with_cluster,
                                                 self.noise_fn)
            losses.update(gl)
            with torch.no_grad():
                self.gm["sr_seed"], self.gm["tg_seed"] = seeds[0].clone(), seeds[1].clone()
            for l, name in enumerate(("p2", "p3", "p4", "p5")):
                losses["loss_adv_" + name] = 0.1 * discriminator_forward(self.dis[name], (feat_s[l], feat_t[l]), 0.02)
            x = torch.cat([clips["source"], clips["target"]], dim=0)
            b, c, h, w, t = x.shape
            x = x.permute(0, 4, 1, 2, 3).reshape(-1, c, h, w)
            cm = clips["masks"].permute(0, 4, 1, 2, 3).reshape(b * t // 2, -1, h, w).float()
            preds, feats = fpn_forward(self.fpn, x, True)
        finally:
            ofpn.UPDATE_RUNNING = False
        half = b * t // 2
        labelled = cm.sum(dim=(1, 2, 3)) > 100
        src_masks = torch.where(labelled.view(-1, 1, 1, 1), cm, preds[:half])
        sf, tf = [f[:half] for f in feats], [f[half:] for f in feats]
        n1, n2, tgl, seeds, _ = gmodule_forward(self.gm, (sf, tf), src_masks, preds[half:], self.nc, self.with_cluster,
                                                self.noise_fn)
        with torch.no_grad():
            self.gm["sr_seed"], self.gm["tg_seed"] = seeds[0].clone(), seeds[1].clone()
        gfeat = [f.reshape(b, -1, f.shape[1], f.shape[2], f.shape[3]) for f in feats]
        tl, _ = tgcn_forward(self.tg, gfeat, (n1.detach().clone(), n2.detach().clone()), [8, 4, 2, 1],
                             "sinkhorn_distance", True)
        losses["temporal_graph_loss"] = sum(tl.values()) + sum(tgl.values())
        for o in self.opts:
            o.zero_grad()
        total = sum(losses.values())
        total.backward()
        for o in self.opts:
            o.step()
        return total.detach(), {k: v.detach() for k, v in losses.items()}, {k: v.detach() for k, v in tl.items()}, \
            {k: v.detach() for k, v in tgl.items()}
