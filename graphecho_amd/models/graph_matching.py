"""Graph-matching domain-adaptation module (reference models/graph_matching.py) on the gfx950 kernels.

``GModule(in_channels, num_classes, device)`` / ``forward(images, features, targets=None, score_maps=None)`` keep
the reference contract: train -> ``(features, (nodes_1, nodes_2), loss_dict)`` with keys ``dis_loss``,
``node_loss``, ``mat_loss_aff``, ``mat_loss_qu``; ``targets=None`` -> ``(features, None)``.  Parameter / buffer
names match (``sr_seed``, ``tg_seed``, ``head_in_ln.{0,3}``, ``node_cls_middle.{0,2}``, ``seed_project_left``,
``{cross,intra}_domain_graph.*``, ``node_affinity.*``, ``node_dis_2.{0,3,6,9}``).

Compute runs on the HIP GEMM / LayerNorm / softmax / fused-Affinity / Sinkhorn kernels.  The data-dependent
node sampling is restated around ONE device->host read per call: class boxes and the byte label of every pyramid
location come from two HIP kernels (ge_mask_boxes, ge_fcos_labels), the labels go to the host in one small pinned
copy, and counts, sampling ranks, class histograms, the class-first order and the label vectors of the regrouped
node sets are planned there (``PrototypeComputation.plan_rows``); the sampled rows are gathered straight from the
NCHW levels (ge_gather_nodes_*).  That replaces the reference's dozens of implicit ``.item()``/boolean-mask
synchronisations.  ``label_maps`` / ``sample`` keep the same steps as batched torch ops (any device): the CPU suite
pins them to the oracle and the GPU suite pins the kernels to them.  Quirks kept on purpose:
``compute_locations`` strides (8,16,32,64) on maps whose true strides are (4,8,16,32) (graph_matching.py:611);
both domains use the box-based sampler (:250-256); class channel 0 doubles as background label 0 (:953-954);
the focal matching loss is divided a second time by len(TP) / sum(FP) (:587-588); seed-bank update uses
scikit-learn SpectralClustering on the host (:539-567), exactly as the reference does.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import functional as GF
from .. import nn as gnn
from .affinity_layer import Affinity
from .gradient_reversal import GradientReversal
from .transformer import MultiHeadAttention

# GE_FUSED_O2O=0: the reference's dozen element-wise ops for the one-to-one matching loss instead of ge_match_o2o_*
FUSED_O2O_LOSS = os.environ.get("GE_FUSED_O2O", "1") != "0"
# GE_FUSED_SEED=0: the seed-bank momentum update as the reference's chain of torch ops instead of ge_seed_bank_update
FUSED_SEED_UPDATE = os.environ.get("GE_FUSED_SEED", "1") != "0"

INF = 100000000


class BCEFocalLoss(torch.nn.Module):
    """-alpha (1-p)^gamma t log p - (1-alpha) p^gamma (1-t) log(1-p)  (graph_matching.py:23-45)."""

    def __init__(self, gamma=2, alpha=0.25, reduction="elementwise_mean"):
        super().__init__()
        self.gamma = gamma
        self.alpha = alpha
        self.reduction = reduction

    def forward(self, _input, target):
        pt = _input
        loss = -self.alpha * (1 - pt) ** self.gamma * target * torch.log(pt) - \
            (1 - self.alpha) * pt ** self.gamma * (1 - target) * torch.log(1 - pt)
        if self.reduction == "elementwise_mean":
            return torch.mean(loss)
        if self.reduction == "sum":
            return torch.sum(loss)
        return loss


def _h2d(values, dtype, device):
    """Small host list / array -> device tensor through pinned memory, without blocking the host: a pageable copy
    waits for everything already queued on the stream, and GModule makes a dozen of these per call."""
    host = torch.as_tensor(values, dtype=dtype)
    if device.type != "cuda":
        return host.to(device)
    return host.pin_memory().to(device, non_blocking=True)


def _first_true(mask, dim):
    """Index of the first True along `dim` (0 if none)."""
    return mask.to(torch.uint8).argmax(dim=dim)


class PrototypeComputation(object):
    """FCOS-style location -> class assignment and per-level node sampling (graph_matching.py:861-1013)."""

    SIZES_OF_INTEREST = ((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF))

    def __init__(self, num_class):
        self.num_class = num_class
        self.class_threshold = (0.5, 1.0)
        self.num_nodes_per_class = 100
        self.num_nodes_per_lvl = 100
        self.bg_ratio = 8
        self.sample_bg_nodes = True

    def label_maps(self, locations, boxes):
        """locations: list of (L_l, 2); boxes (B, nc, 4) -> per-level int64 labels (B*L_l,), image-major.

        graph_matching.py:874-959: a location takes the class whose box contains it (strictly), whose
        max(l,t,r,b) lies in the level's size range, and has minimal area (first class on ties); else 0."""
        pts = torch.cat(locations, dim=0)                                   # (L, 2)
        lo = torch.cat([pts.new_full((len(p),), float(self.SIZES_OF_INTEREST[l][0])) for l, p in enumerate(locations)])
        hi = torch.cat([pts.new_full((len(p),), float(self.SIZES_OF_INTEREST[l][1])) for l, p in enumerate(locations)])
        xs, ys = pts[:, 0][None, :, None], pts[:, 1][None, :, None]          # (1, L, 1)
        bx = boxes[:, None, :, :]                                           # (B, 1, nc, 4)
        l_, t_ = xs - bx[..., 0], ys - bx[..., 1]
        r_, b_ = bx[..., 2] - xs, bx[..., 3] - ys
        reg = torch.stack([l_, t_, r_, b_], dim=-1)                         # (B, L, nc, 4)
        inside = reg.min(dim=-1)[0] > 0
        mx = reg.max(dim=-1)[0]
        cared = (mx >= lo[None, :, None]) & (mx <= hi[None, :, None])
        area = ((boxes[..., 3] - boxes[..., 1]) * (boxes[..., 2] - boxes[..., 0]))[:, None, :]
        area = area.expand(-1, pts.shape[0], -1).clone()
        area[~(inside & cared)] = INF
        min_area, inds = area.min(dim=2)
        labels = torch.where(min_area == INF, torch.zeros_like(inds), inds)  # (B, L)
        out, off = [], 0
        for p in locations:
            out.append(labels[:, off:off + len(p)].reshape(-1))
            off += len(p)
        return out

    @staticmethod
    def _take_ranked(mask, ranks):
        """Row indices of the ranks-th True entries of a boolean vector, without a host sync."""
        csum = torch.cumsum(mask.to(torch.int32), dim=0)
        return torch.searchsorted(csum, (ranks + 1).to(torch.int32))

    def plan(self, counts):
        """Per level: (positive ranks, negative ranks) as python ranges, from the (n_pos, n_neg) counts."""
        out = []
        for n_pos, n_neg in counts:
            step = n_pos // self.num_nodes_per_class
            pos = list(range(0, n_pos, step)) if step > 1 else list(range(n_pos))
            if n_pos > n_neg:
                neg = list(range(n_neg))
            else:
                k = len(pos) // self.bg_ratio
                neg = [int(v) for v in np.floor(np.linspace(0, n_neg - 2, k))] if k > 0 else []
                neg = [v % n_neg if n_neg > 0 else 0 for v in neg] if n_neg > 0 else []
            out.append((pos, neg))
        return out

    def sample(self, features, labels, plan):
        """Gather the planned rows: returns (nodes (N, C), labels (N,)) ordered [bg p2..p5, fg p2..p5]."""
        dev = features[0].device
        C = features[0].shape[1]
        pos_pts, pos_lab, neg_pts = [], [], []
        for feat, lab, (pos, neg) in zip(features, labels, plan):
            hw = feat.shape[2] * feat.shape[3]
            planes = feat.reshape(feat.shape[0], C, hw)

            def rows(idx):      # (n, C) feature rows of flat (b, y, x) positions, gathered straight from NCHW
                b = torch.div(idx, hw, rounding_mode="floor")      # (no NHWC copy of the whole level for ~200 rows)
                return planes[b, :, idx - b * hw]

            if pos:
                idx = self._take_ranked(lab > 0, _h2d(pos, torch.int64, dev))
                pos_pts.append(rows(idx))
                pos_lab.append(lab[idx])
            if neg:
                idx = self._take_ranked(lab == 0, _h2d(neg, torch.int64, dev))
                neg_pts.append(rows(idx))
        empty = features[0].new_zeros((0, C))
        pos_pts = torch.cat(pos_pts, dim=0) if pos_pts else empty
        pos_lab = torch.cat(pos_lab) if pos_lab else torch.zeros(0, dtype=torch.int64, device=dev)
        neg_pts = torch.cat(neg_pts, dim=0) if neg_pts else empty
        nodes = torch.cat([neg_pts, pos_pts], dim=0)
        lab = torch.cat([pos_lab.new_zeros(neg_pts.shape[0]), pos_lab])
        return nodes, lab


    def plan_rows(self, labels, level_sizes):
        """Host-side sampling plan of one domain from its byte label maps: labels (B, sum L_l) uint8 numpy, levels
        concatenated as `ge_fcos_labels` writes them.  Returns (level, index, node_labels, unique, present):
        row i of the node set is location index[i] (= b * L_l + position, image-major as in `label_maps`) of pyramid level
        level[i]; rows are ordered [background p2..p5, foreground p2..p5] exactly as `sample` orders them; `unique` is
        False if a background rank repeats; `present[l]` says whether level l contributes a row."""
        neg_i, neg_l, pos_i, pos_l, pos_lab = [], [], [], [], []
        unique, off = True, 0
        present = [False] * len(level_sizes)
        for lvl, L in enumerate(level_sizes):
            lab = labels[:, off:off + L].reshape(-1)
            off += L
            pos_all, neg_all = np.flatnonzero(lab > 0), np.flatnonzero(lab == 0)
            (pos, neg), = self.plan([(int(pos_all.size), int(neg_all.size))])
            if pos:
                idx = pos_all[np.asarray(pos, dtype=np.int64)]
                pos_i.append(idx)
                pos_l.append(np.full(idx.size, lvl, dtype=np.int64))
                pos_lab.append(lab[idx].astype(np.int64))
                present[lvl] = True
            if neg:
                ranks = np.asarray(neg, dtype=np.int64)
                unique = unique and np.unique(ranks).size == ranks.size
                idx = neg_all[ranks]
                neg_i.append(idx)
                neg_l.append(np.full(idx.size, lvl, dtype=np.int64))
                present[lvl] = True
        empty = np.zeros(0, dtype=np.int64)
        n_neg = sum(a.size for a in neg_i)
        index = np.concatenate(neg_i + pos_i) if (neg_i or pos_i) else empty
        level = np.concatenate(neg_l + pos_l) if (neg_l or pos_l) else empty
        node_labels = np.concatenate([np.zeros(n_neg, dtype=np.int64)] + pos_lab) if (n_neg or pos_lab) else empty
        return level, index.astype(np.int64), node_labels, unique, present


class GModule(torch.nn.Module):
    LOSS_KEYS = ("dis_loss", "node_loss", "mat_loss_aff", "mat_loss_qu")   # every key the training forward may return

    def __init__(self, in_channels, num_classes, device):
        super().__init__()
        self.device = device
        self.fpn_strides = [8, 16, 32, 64, 128]
        self.num_classes = num_classes
        self.matching_loss_type = "FL"
        self.matching_cfg = "o2o"
        self.with_cluster_update = True
        self.with_semantic_completion = True
        self.with_quadratic_matching = True
        self.weight_matching = 0.1
        self.weight_nodes = 1.0
        self.weight_dis = 0.1
        self.lambda_dis = 0.02
        self.with_domain_interaction = True
        self.with_complete_graph = True
        self.with_node_dis = True
        self.with_global_graph = False
        self.node_dis_place = "feat"
        self.with_cond_cls = False
        self.with_score_weight = False
        self.graph_generator = PrototypeComputation(num_classes)
        self.head_in_cfg = "LN"
        self.head_in_ln = nn.Sequential(
            gnn.Linear(256, 256), gnn.LayerNorm(256, elementwise_affine=False), gnn.ReLU(),
            gnn.Linear(256, 256), gnn.LayerNorm(256, elementwise_affine=False))
        self.node_cls_middle = nn.Sequential(gnn.Linear(256, 512), gnn.ReLU(), gnn.Linear(512, self.num_classes))
        self.seed_project_left = gnn.Linear(256, 256)
        self.register_buffer("sr_seed", torch.randn(self.num_classes, 256))
        self.register_buffer("tg_seed", torch.randn(self.num_classes, 256))
        self.cross_domain_graph = MultiHeadAttention(256, 1, dropout=0.1, version="v2")
        self.intra_domain_graph = MultiHeadAttention(256, 1, dropout=0.1, version="v2")
        self.node_affinity = Affinity(d=256)
        self.InstNorm_layer = gnn.InstanceNormMatrix()
        self.matching_loss = BCEFocalLoss()
        self.quadratic_loss = torch.nn.L1Loss(reduction="mean")
        self.grad_reverse = GradientReversal(self.lambda_dis)
        self.node_dis_2 = nn.Sequential(
            gnn.Linear(256, 256), gnn.LayerNorm(256, elementwise_affine=False), gnn.ReLU(),
            gnn.Linear(256, 256), gnn.LayerNorm(256, elementwise_affine=False), gnn.ReLU(),
            gnn.Linear(256, 256), gnn.LayerNorm(256, elementwise_affine=False), gnn.ReLU(),
            gnn.Linear(256, 1))
        self._init_weight()
        self._loc_cache = {}

    def _init_weight(self, init_item=None):
        nn.init.normal_(self.seed_project_left.weight, std=0.01)
        nn.init.constant_(self.seed_project_left.bias, 0)
        for seq in (self.node_dis_2, self.node_cls_middle, self.head_in_ln):
            for layer in seq:
                if isinstance(layer, nn.Linear):
                    nn.init.normal_(layer.weight, std=0.01)
                    nn.init.constant_(layer.bias, 0)

    # ---- public entry ---------------------------------------------------------------------------------------
    def forward(self, images, features, targets=None, score_maps=None, prepared=None):
        if targets is not None:
            return self._forward_train(images, features, targets, score_maps, prepared)
        return self._forward_inference(images, features), None

    def _forward_inference(self, images, features):
        return features

    # ---- geometry -------------------------------------------------------------------------------------------
    def compute_locations(self, features):
        return [self.compute_locations_per_level(f.shape[-2], f.shape[-1], self.fpn_strides[lvl], f.device)
                for lvl, f in enumerate(features)]

    def compute_locations_per_level(self, h, w, stride, device):
        key = (h, w, stride, str(device))
        if key not in self._loc_cache:
            sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32, device=device)
            sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32, device=device)
            gy, gx = torch.meshgrid(sy, sx, indexing="ij")
            self._loc_cache[key] = torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1) + stride // 2
        return self._loc_cache[key]

    def masks_to_boxes(self, masks):
        """(N, H, W) -> (N, 4) tight (x1, y1, x2, y2) of the non-zero pixels; (0, 0, W, H) for an empty mask
        (graph_matching.py:702-740).  Batched, no host reads."""
        N, H, W = masks.shape
        if masks.numel() == 0:
            return torch.zeros((0, 4), device=masks.device, dtype=torch.float)
        nz = masks != 0
        cols, rows = nz.any(dim=1), nz.any(dim=2)                           # (N, W), (N, H)
        x1, y1 = _first_true(cols, 1), _first_true(rows, 1)
        x2 = W - 1 - _first_true(cols.flip(1), 1)
        y2 = H - 1 - _first_true(rows.flip(1), 1)
        box = torch.stack([x1, y1, x2, y2], dim=1).to(torch.float)
        key = ("full_box", W, H, box.device)
        if key not in self._loc_cache:
            self._loc_cache[key] = box.new_tensor([0, 0, W, H])
        full = self._loc_cache[key].expand(N, 4)
        return torch.where(cols.any(dim=1, keepdim=True), box, full)

    def find_bbox(self, masks):
        B, nc, H, W = masks.shape
        return self.masks_to_boxes(masks.reshape(B * nc, H, W)).reshape(B, nc, 4)

    def one_hot(self, x):
        return torch.eye(self.num_classes, device=x.device)[x.long(), :]

    # ---- training forward -----------------------------------------------------------------------------------
    def prepare(self, features, targets, score_maps):
        """First stage of the training forward, split off so that a caller can enqueue independent device work behind it:
        class boxes of both domains (ge_mask_boxes), the byte label of every pyramid location (ge_fcos_labels), the copy
        of those labels to the host (pinned buffer, not waited for) and the event that marks it.  Pass the result as
        ``prepared=`` to ``forward``; whatever was enqueued in between keeps the device busy while the host plans the
        node sampling from the labels."""
        features_s, features_t = features
        levels = [(f.shape[-2], f.shape[-1], self.fpn_strides[l]) for l, f in enumerate(features_s)]
        if [tuple(f.shape[-2:]) for f in features_t] != [lv[:2] for lv in levels]:
            raise NotImplementedError("GModule: source and target pyramids of different geometry")
        ranges = PrototypeComputation.SIZES_OF_INTEREST[:len(levels)]
        if not features_s[0].is_cuda:     # host-logic tests: the torch restatement of the two kernels
            gen = self.graph_generator
            lab = [torch.cat([l.reshape(m.shape[0], -1) for l in
                              gen.label_maps(self.compute_locations(f), self.find_bbox(m))], dim=1)
                   for f, m in ((features_s, targets), (features_t, score_maps))]
            return torch.cat(lab).to(torch.uint8), None, targets.shape[0]
        boxes = torch.cat([GF.mask_boxes(m.reshape(-1, m.shape[-2], m.shape[-1])).view(m.shape[0], m.shape[1], 4)
                           for m in (targets, score_maps)])
        labels = GF.fcos_labels(boxes, levels, ranges)
        host = torch.empty(labels.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(labels, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev, targets.shape[0]

    def _forward_train(self, images, features, targets=None, score_maps=None, prepared=None):
        features_s, features_t = features
        losses = {}
        gen = self.graph_generator
        host, ev, n_src = prepared if prepared is not None else self.prepare(features, targets, score_maps)
        # the one host read of the sampling: byte labels of every location of both domains
        if ev is not None:
            ev.synchronize()
        lab = host.numpy()
        sizes = [f.shape[-2] * f.shape[-1] for f in features_s]
        lvl_s, idx_s, nl_s, uniq_s, pres_s = gen.plan_rows(lab[:n_src], sizes)
        lvl_t, idx_t, nl_t, uniq_t, pres_t = gen.plan_rows(lab[n_src:], sizes)
        if idx_s.size < 6:                              # graph_matching.py:258-260 (only the source count is checked)
            tab = _h2d(np.concatenate([lvl_s, idx_s, nl_s, lvl_t, idx_t, nl_t]), torch.int64, features_s[0].device)
            a, b = idx_s.size, idx_t.size
            nodes_1 = GF.gather_nodes(features_s, tab[:a], tab[a:2 * a], uniq_s, pres_s)
            nodes_2 = GF.gather_nodes(features_t, tab[3 * a:3 * a + b], tab[3 * a + b:3 * a + 2 * b], uniq_t, pres_t)
            return features, (nodes_1, nodes_2), losses
        # everything the class-first regrouping needs is known on the host already: stable class order of both node
        # sets, the class histograms, the labels of the regrouped (and completed) sets
        so, to = np.argsort(nl_s, kind="stable"), np.argsort(nl_t, kind="stable")
        nc = self.num_classes
        hist = [np.bincount(nl_s, minlength=nc)[:nc].tolist(), np.bincount(nl_t, minlength=nc)[:nc].tolist()]
        out_s, out_t = [], []
        for c in range(nc):
            ns, nt = hist[0][c], hist[1][c]
            if ns or nt:
                out_s += [c] * (ns if ns else nt)       # a class missing on one side is completed with as many
                out_t += [c] * (nt if nt else ns)       # hallucinated nodes as the other side has (:432-472)
        a, b = idx_s.size, idx_t.size
        dev = features_s[0].device
        tab = _h2d(np.concatenate([lvl_s, idx_s, nl_s, so, lvl_t, idx_t, nl_t, to]), torch.int64, dev)
        flab = _h2d(np.asarray(out_s + out_t, dtype=np.float32), torch.float32, dev)
        nodes_1 = GF.gather_nodes(features_s, tab[:a], tab[a:2 * a], uniq_s, pres_s)
        t0 = 4 * a
        nodes_2 = GF.gather_nodes(features_t, tab[t0:t0 + b], tab[t0 + b:t0 + 2 * b], uniq_t, pres_t)
        labels_1, labels_2 = tab[2 * a:3 * a], tab[t0 + 2 * b:t0 + 3 * b]
        order = (tab[3 * a:4 * a], tab[t0 + 3 * b:t0 + 4 * b])
        final = (flab[:len(out_s)], flab[len(out_s):])
        host_final = (np.asarray(out_s, dtype=np.int64), np.asarray(out_t, dtype=np.int64))

        if self.with_node_dis and self.node_dis_place == "feat":
            losses["dis_loss"] = self._node_dis(nodes_1, nodes_2)
        nodes_1 = self.head_in_ln(nodes_1)
        # An EMPTY target node set is a regular case early in training (pseudo-label boxes covering the whole image put
        # every location in the background, and background sampling is len(fg) // 8 = 0): the reference carries the
        # (0, 256) tensor through and completes every source class on the target side from the seed bank (:455-472).
        nodes_2 = self.head_in_ln(nodes_2) if nodes_2.size(0) > 0 else nodes_2

        (nodes_1, nodes_2), (labels_1, labels_2) = \
            self._forward_preprocessing_source_target((nodes_1, nodes_2), (labels_1, labels_2),
                                                      plan=(order, hist, final))
        if self.with_complete_graph:
            nodes_1, edges_1 = self._forward_intra_domain_graph(nodes_1)
            nodes_2, edges_2 = self._forward_intra_domain_graph(nodes_2)
        self.update_seed(nodes_1, labels_1, nodes_2, labels_2, host_labels=host_final)
        if self.with_node_dis and self.node_dis_place == "intra":
            losses["dis_loss"] = self._node_dis(nodes_1, nodes_2)
        if self.with_domain_interaction:
            nodes_1, nodes_2 = self._forward_cross_domain_graph(nodes_1, nodes_2)
        if self.with_node_dis and self.node_dis_place == "inter":
            losses["dis_loss"] = self._node_dis(nodes_1, nodes_2)
        node_loss = self._forward_node_loss(torch.cat([nodes_1, nodes_2], dim=0),
                                            torch.cat([labels_1, labels_2], dim=0))
        losses["node_loss"] = self.weight_nodes * node_loss
        if self.matching_cfg != "none":
            loss_aff, affinity = self._forward_aff(nodes_1, nodes_2, labels_1, labels_2)
            losses["mat_loss_aff"] = self.weight_matching * loss_aff
            if self.with_quadratic_matching:
                losses["mat_loss_qu"] = self._forward_qu(edges_1.detach(), edges_2.detach(), affinity)
        return features, (nodes_1, nodes_2), losses

    def _node_dis(self, nodes_1, nodes_2):
        """GRL + 4-layer node discriminator, BCE(source->1, target->0) * weight_dis (graph_matching.py:263-270)."""
        rev = self.node_dis_2(self.grad_reverse(torch.cat([nodes_1, nodes_2], dim=0)))
        tgt = torch.cat([torch.ones(nodes_1.size(0), device=rev.device), torch.zeros(nodes_2.size(0), device=rev.device)])
        return self.weight_dis * GF.bce_with_logits(rev.view(-1), tgt)

    # Source of the hallucination branch's standard-normal draws: None = the device RNG; tests set a callable
    # ``noise_fn(n, 256) -> tensor`` to feed the reference, the oracle and this module the same noise.
    noise_fn = None

    def _hallucinate(self, seed_row, like_nodes):
        """Nodes for a class missing on one side: seed + Gaussian noise (graph_matching.py:432-472)."""
        n = like_nodes.shape[0]
        base = seed_row.unsqueeze(0).expand(n, 256)
        eps = torch.randn(n, 256, device=like_nodes.device) if self.noise_fn is None else \
            self.noise_fn(n, 256).to(like_nodes.device)
        if not self.with_semantic_completion:
            out = eps * 0.01
        elif n < 5:
            out = eps * 0.01 + base
        else:
            # = torch.normal(mean=base, std=...) without its host-side "std >= 0" check (a device sync)
            out = eps * like_nodes.std(0).unsqueeze(0) + base
        return self.seed_project_left(out)

    def _forward_preprocessing_source_target(self, nodes, labels, weights=None, plan=None):
        """Regroup both node sets class-first (ascending class id) and complete classes missing on one side
        (graph_matching.py:381-483).  plan: (stable class orders, class histograms, final label vectors) of both sets as
        the caller derived them on the host from the label maps; None reads the labels back (one host read)."""
        sr_nodes, tg_nodes = nodes
        sr_lab, tg_lab = labels
        nc = self.num_classes
        if plan is None:
            hs, ht = sr_lab.cpu().numpy(), tg_lab.cpu().numpy()
            hist = [np.bincount(hs, minlength=nc)[:nc].tolist(), np.bincount(ht, minlength=nc)[:nc].tolist()]
            so = torch.as_tensor(np.argsort(hs, kind="stable"), device=sr_lab.device)
            to = torch.as_tensor(np.argsort(ht, kind="stable"), device=sr_lab.device)
            final = None
        else:
            (so, to), hist, final = plan
        sr_sorted, tg_sorted = sr_nodes[so], tg_nodes[to]
        if all((a > 0) == (b > 0) for a, b in zip(*hist)):
            return (sr_sorted, tg_sorted), (final if final is not None else (sr_lab[so].float(), tg_lab[to].float()))
        sr_parts, tg_parts, sl, tl = [], [], [], []
        so_off = to_off = 0
        for c in range(nc):
            ns, nt = hist[0][c], hist[1][c]
            if ns == 0 and nt == 0:
                continue
            s_c = sr_sorted[so_off:so_off + ns]
            t_c = tg_sorted[to_off:to_off + nt]
            so_off += ns
            to_off += nt
            if ns == 0:
                s_c = self._hallucinate(self.sr_seed[c], t_c)
            elif nt == 0:
                t_c = self._hallucinate(self.tg_seed[c], s_c)
            sr_parts.append(s_c)
            tg_parts.append(t_c)
            if final is None:
                sl.append(torch.full((s_c.shape[0],), float(c), device=s_c.device))
                tl.append(torch.full((t_c.shape[0],), float(c), device=s_c.device))
        return (torch.cat(sr_parts), torch.cat(tg_parts)), (final if final is not None else (torch.cat(sl), torch.cat(tl)))

    def _forward_preprocessing_source(self, sr_nodes, sr_nodes_label):
        """Source-only split (even rows / odd rows per class), graph_matching.py:354-379."""
        n1, n2, l1, l2 = [], [], [], []
        for c in sr_nodes_label.unique():
            rows = sr_nodes[sr_nodes_label == c]
            n1.append(rows[::2])
            n2.append(rows[1::2])
            l1.append(rows.new_ones(len(n1[-1])) * c)
            l2.append(rows.new_ones(len(n2[-1])) * c)
        return (torch.cat(n1), torch.cat(n2)), (torch.cat(l1), torch.cat(l2))

    def _forward_intra_domain_graph(self, nodes):
        return self.intra_domain_graph(nodes, nodes, nodes)

    def _forward_cross_domain_graph(self, nodes_1, nodes_2):
        if self.with_global_graph:
            n_1 = len(nodes_1)
            g = torch.cat([nodes_1, nodes_2], dim=0)
            g = self.cross_domain_graph(g, g, g)[0]
            return g[:n_1], g[n_1:]
        nodes2_enhanced = self.cross_domain_graph(nodes_1, nodes_1, nodes_2)[0]
        nodes1_enhanced = self.cross_domain_graph(nodes_2, nodes_2, nodes_1)[0]
        return nodes1_enhanced, nodes2_enhanced

    def _forward_node_loss(self, nodes, labels, weights=None):
        labels = labels.long()
        assert len(nodes) == len(labels)
        logits = self.node_cls_middle(nodes)
        return F.cross_entropy(logits, labels, reduction="mean")

    # ---- seed bank ------------------------------------------------------------------------------------------
    # The momentum update needs, per class and domain, a scikit-learn SpectralClustering fit on the host (as in the
    # reference).  Those fits do not feed anything else in the step, so they are submitted to worker processes
    # (cluster_pool) and the bank is brought up to date at its next read: the next update_seed, a hallucinated
    # class, any access to ``sr_seed`` / ``tg_seed``, or ``state_dict()``.  Same inputs, same order, same result as
    # running them inline (``async_seed_update = False``).
    async_seed_update = True

    def __getattr__(self, name):
        if name in ("sr_seed", "tg_seed") and self.__dict__.get("_pending_seed"):
            self._flush_seed_updates()
        return super().__getattr__(name)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_seed_updates()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._flush_seed_updates()
        super()._load_from_state_dict(*args, **kwargs)

    @torch.no_grad()
    def update_seed(self, sr_nodes, sr_labels, tg_nodes=None, tg_labels=None, k=20, host_labels=None):
        """Momentum update of the per-class seed bank from (spectral-cluster-filtered) class means
        (graph_matching.py:532-567).  Clustering runs in scikit-learn on the host, as in the reference."""
        self._flush_seed_updates()   # the previous update is an input of this one (seed row is clustered with the nodes)
        banks = [("sr_seed", sr_nodes.detach(), sr_labels)]
        if tg_nodes is not None:
            banks.append(("tg_seed", tg_nodes.detach(), tg_labels))
        pending = []
        cluster = self.with_cluster_update
        if cluster:   # one device->host read of everything the fits need
            packed = torch.cat([n for _, n, _ in banks] + [self._buffers[name] for name, _, _ in banks]).cpu().numpy()
        if host_labels is not None:       # the caller planned the node sets on the host: no read-back of the labels
            labels_h = np.concatenate([np.asarray(h, dtype=np.int64) for h in host_labels[:len(banks)]])
        else:
            labels_h = torch.cat([l.long() for _, _, l in banks]).cpu().numpy()
        off = 0
        seed_off = sum(n.shape[0] for _, n, _ in banks)
        pool = None
        for name, nodes, _ in banks:
            N = nodes.shape[0]
            lab = labels_h[off:off + N]
            entry = {"name": name, "nodes": nodes, "classes": []}
            for c in range(self.num_classes):
                idx = np.nonzero(lab == c)[0]
                if idx.size == 0:
                    continue
                ticket = None
                if idx.size > k and cluster:
                    rows = np.concatenate([packed[seed_off + c][None, :], packed[off + idx]])
                    if self.async_seed_update:
                        if pool is None:
                            from ..cluster_pool import get_pool

                            pool = get_pool()
                        ticket = (pool, pool.submit(rows, idx.size // 2))
                    else:
                        from .._cluster_worker import spectral_keep

                        ticket = (None, spectral_keep(rows, idx.size // 2))
                entry["classes"].append((c, idx, ticket))
            pending.append(entry)
            off += N
            seed_off += self.num_classes
        self.__dict__["_pending_seed"] = pending
        if not self.async_seed_update:
            self._flush_seed_updates()

    @torch.no_grad()
    def _flush_seed_updates(self):
        pending = self.__dict__.get("_pending_seed")
        if not pending:
            return
        self.__dict__["_pending_seed"] = None
        for entry in pending:
            bank, nodes = self._buffers[entry["name"]], entry["nodes"]
            sel = np.zeros((self.num_classes, nodes.shape[0]), dtype=np.float32)
            cnt = np.zeros((self.num_classes, 1), dtype=np.float32)
            has = np.zeros((self.num_classes, 1), dtype=bool)
            for c, idx, ticket in entry["classes"]:
                if ticket is not None:
                    keep = ticket[0].result(ticket[1]) if ticket[0] is not None else ticket[1]
                    idx = idx[np.asarray(keep, dtype=bool)]
                sel[c, idx] = 1.0
                cnt[c] = idx.size
                has[c] = True
            dev = nodes.device
            if FUSED_SEED_UPDATE and nodes.is_cuda and nodes.shape[0] > 0:
                # class means, cosine similarity with the bank rows and the blend in ONE launch (ge_seed_bank_update) fed by one
                # small host-to-device copy, instead of the ~13 launches and three copies of the lines below
                tab = np.full(nodes.shape[0] + self.num_classes, -1, dtype=np.int32)
                tab[nodes.shape[0]:] = 0
                for c, _idx, _t in entry["classes"]:
                    tab[np.nonzero(sel[c])[0]] = c
                    tab[nodes.shape[0] + c] = 1
                GF.seed_bank_update(bank, nodes, _h2d(tab, torch.int32, dev), self.num_classes)
                continue
            sums = GF.matmul(_h2d(sel, torch.float32, dev), nodes)            # (nc, N) x (N, 256): kept-row sums
            means = sums / _h2d(cnt, torch.float32, dev)                       # empty cluster -> NaN, as the reference
            momentum = F.cosine_similarity(means, bank, dim=1).unsqueeze(1)
            new = bank * momentum + means * (1.0 - momentum)
            bank.copy_(torch.where(_h2d(has, torch.bool, dev), new, bank))

    def _forward_aff(self, nodes_1, nodes_2, labels_side1, labels_side2):
        if self.matching_cfg == "o2o":
            M = self.node_affinity(nodes_1, nodes_2)
            M = self.InstNorm_layer(M[None, None, :, :])
            if FUSED_O2O_LOSS and M.is_cuda:
                # the focal TP / FP terms below in two launches (+ one backward) on the log plan: same values to rounding
                return GF.match_o2o_loss(self.sinkhorn_rpm(M[:, 0, :, :], n_iters=20).squeeze(0), labels_side1, labels_side2)
            target = (labels_side1.long()[:, None] == labels_side2.long()[None, :]).float()
            M = self.sinkhorn_rpm(M[:, 0, :, :], n_iters=20).squeeze(0).exp()
            # TP: per row, the best same-class entry; FP: every different-class entry (graph_matching.py:577-590)
            indx = (M * target).max(-1)[1]
            tp = M.gather(1, indx[:, None])
            tp_loss = (-0.25 * (1 - tp) ** 2 * torch.log(tp)).mean() / tp.shape[0]
            fp_mask = 1.0 - target
            fp_terms = -0.75 * M ** 2 * torch.log(1 - M) * fp_mask
            fp_loss = fp_terms.sum() / fp_mask.sum() / (M * fp_mask).sum().detach()
            return tp_loss + fp_loss, M
        if self.matching_cfg == "m2m":
            M = self.node_affinity(nodes_1, nodes_2)
            target = (labels_side1.long()[:, None] == labels_side2.long()[None, :]).float()
            return self.matching_loss(M.sigmoid(), target).mean(), M
        return 0, None

    def _forward_qu(self, edge_1, edge_2, affinity):
        R = GF.matmul(edge_1, affinity) - GF.matmul(affinity, edge_2)
        return R.abs().mean()

    def sinkhorn_rpm(self, log_alpha, n_iters=5, slack=True, eps=-1):
        """Log-domain Sinkhorn with a slack row/column (graph_matching.py:637-689) on the fused kernels."""
        if not slack or eps > 0:
            raise NotImplementedError("only the slack=True, eps<=0 configuration used by the reference is built")
        return GF.sinkhorn_rpm(log_alpha, n_iters)

    def dynamic_fc(self, features, kernel_par):
        return GF.matmul(features, kernel_par, False, True)
