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
