"""Oracle: GModule._forward_train as a pure function of a state_dict, restated literally (per-image loops,
boolean masks) from reference models/graph_matching.py:244-352, 505-530, 569-635, 702-746, 874-1013.

Scope of the restatement: dropout p=0; the seed-bank update is returned separately (mean-only or with scikit-learn
clustering); the hallucination branch (:432-472: a class present on one side only is completed from the seed bank plus
Gaussian noise) takes its standard-normal draws from `noise_fn(n, 256)` (tests pass the same named deterministic
stream to the reference, the oracle and the HIP module); the `< 6 source nodes` early return (:258-260) gives an empty
loss dict.
TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .misc import affinity, grad_reverse, layer_norm, mha_v2, sinkhorn_rpm

INF = 100000000
STRIDES = [8, 16, 32, 64, 128]
SIZES = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, INF]]


def masks_to_boxes(masks):
    n, H, W = masks.shape
    out = torch.zeros((n, 4))
    for i, m in enumerate(masks):
        y, x = torch.where(m != 0)
        if x.numel() == 0:
            out[i] = torch.tensor([0.0, 0.0, W, H])
        else:
            out[i] = torch.tensor([x.min(), y.min(), x.max(), y.max()], dtype=torch.float)
    return out


def locations(h, w, stride):
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    gy, gx = torch.meshgrid(sy, sx, indexing="ij")
    return torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1) + stride // 2


def sample_nodes(features, masks, num_class):
    """PrototypeComputation.__call__ source branch -> (nodes, labels)."""
    boxes = [masks_to_boxes(m) for m in masks]
    locs = [locations(f.shape[2], f.shape[3], STRIDES[l]) for l, f in enumerate(features)]
    pts = torch.cat(locs)
    soi = torch.cat([torch.tensor(SIZES[l], dtype=torch.float32)[None].expand(len(p), -1) for l, p in enumerate(locs)])
    per_im = []
    for bx in boxes:
        area = torch.tensor([(b[3] - b[1]) * (b[2] - b[0]) for b in bx])
        xs, ys = pts[:, 0], pts[:, 1]
        reg = torch.stack([xs[:, None] - bx[:, 0][None], ys[:, None] - bx[:, 1][None],
                           bx[:, 2][None] - xs[:, None], bx[:, 3][None] - ys[:, None]], dim=2)
        inside = reg.min(dim=2)[0] > 0
        mx = reg.max(dim=2)[0]
        cared = (mx >= soi[:, [0]]) & (mx <= soi[:, [1]])
        a = area[None].repeat(len(pts), 1)
        a[inside == 0] = INF
        a[cared == 0] = INF
        mn, ind = a.min(dim=1)
        lab = torch.arange(num_class)[ind]
        lab[mn == INF] = 0
        per_im.append(torch.split(lab, [len(p) for p in locs]))
    labels = [torch.cat([im[l] for im in per_im]) for l in range(len(locs))]
    C = features[0].shape[1]
    pos_pts, pos_lab, neg_pts = [], [], []
    for l, f in enumerate(features):
        rows = f.permute(0, 2, 3, 1).reshape(-1, C)
        pi, ni = labels[l] > 0, labels[l] == 0
        pa, la = rows[pi], labels[l][pi]
        step = len(la) // 100
        if step > 1:
            pa, la = pa[::step], la[::step]
        pos_pts.append(pa)
        pos_lab.append(la)
        if int(pi.sum()) > int(ni.sum()):
            neg_pts.append(rows[ni])
        else:
            idx = [int(v) for v in np.floor(np.linspace(0, int(ni.sum()) - 2, len(la) // 8))]
            neg_pts.append(rows[ni][idx])
    pos_pts, pos_lab, neg_pts = torch.cat(pos_pts), torch.cat(pos_lab), torch.cat(neg_pts)
    return torch.cat([neg_pts, pos_pts]), torch.cat([pos_lab.new_zeros(len(neg_pts)), pos_lab])


def class_first(nodes, labels):
    parts, lab = [], []
    for c in labels.unique():
        parts.append(nodes[labels == c])
        lab.append(torch.full((len(parts[-1]),), float(c)))
    return torch.cat(parts), torch.cat(lab)


def class_first_complete(sd, nodes, labels, noise_fn):
    """_forward_preprocessing_source_target (graph_matching.py:381-483): both node sets regrouped class-first over the
    UNION of their labels; a class missing on one side gets seed + noise nodes pushed through seed_project_left
    (with_semantic_completion=True: 0.01-sigma noise below 5 nodes, else the other side's per-feature std)."""
    (sn, tn), (sl, tl) = nodes, labels
    sp, tp, slab, tlab = [], [], [], []

    def hallucinate(seed_row, like):
        n = len(like)
        base = seed_row[None].expand(n, 256)
        eps = noise_fn(n, 256)
        out = 0.01 * eps + base if n < 5 else base + like.std(0)[None].expand(n, 256) * eps
        return F.linear(out, sd["seed_project_left.weight"], sd["seed_project_left.bias"])

    for c in torch.cat([sl, tl]).unique():
        s_c, t_c = sn[sl == c], tn[tl == c]
        if len(s_c) == 0:
            s_c = hallucinate(sd["sr_seed"][int(c)], t_c)
        elif len(t_c) == 0:
            t_c = hallucinate(sd["tg_seed"][int(c)], s_c)
        sp.append(s_c)
        tp.append(t_c)
        slab.append(torch.full((len(s_c),), float(c)))
        tlab.append(torch.full((len(t_c),), float(c)))
    return (torch.cat(sp), torch.cat(tp)), (torch.cat(slab), torch.cat(tlab))


def seed_update(seed, nodes, labels, with_cluster):
    """update_seed for one bank (graph_matching.py:535-550); returns the new bank."""
    seed = seed.clone()
    for c in labels.unique().long():
        bs = nodes[labels == c].detach()
        if len(bs) > 20 and with_cluster:
            import sklearn.cluster as cluster

            sp = cluster.SpectralClustering(2, affinity="nearest_neighbors", n_jobs=-1, assign_labels="kmeans",
                                            random_state=1234, n_neighbors=len(bs) // 2)
            ind = sp.fit_predict(torch.cat([seed[c][None], bs]).numpy())
            bs = bs[torch.from_numpy((ind == ind[0])[1:])].mean(0)
        else:
            bs = bs.mean(0)
        mom = F.cosine_similarity(bs[None], seed[c][None])
        seed[c] = seed[c] * mom + bs * (1.0 - mom)
    return seed


def _node_dis(sd, n1, n2):
    z = grad_reverse(torch.cat([n1, n2]), 0.02)
    for i in (0, 3, 6):
        z = F.relu(layer_norm(F.linear(z, sd[f"node_dis_2.{i}.weight"], sd[f"node_dis_2.{i}.bias"])))
    z = F.linear(z, sd["node_dis_2.9.weight"], sd["node_dis_2.9.bias"]).view(-1)
    return 0.1 * F.binary_cross_entropy_with_logits(z, torch.cat([torch.ones(len(n1)), torch.zeros(len(n2))]))


def gmodule_forward(sd, features, targets, score_maps, num_class, with_cluster=False, noise_fn=None):
    """-> (nodes_1, nodes_2, losses dict, (new_sr_seed, new_tg_seed), raw node counts)."""
    fs, ft = features
    n1, l1 = sample_nodes(fs, targets, num_class)
    n2, l2 = sample_nodes(ft, score_maps, num_class)
    counts = (len(n1), len(n2))
    if len(n1) < 6:                       # graph_matching.py:258-260: no losses, seed banks untouched
        return n1, n2, {}, (sd["sr_seed"], sd["tg_seed"]), counts
    losses = {"dis_loss": _node_dis(sd, n1, n2)}

    def head(z):
        z = F.relu(layer_norm(F.linear(z, sd["head_in_ln.0.weight"], sd["head_in_ln.0.bias"])))
        return layer_norm(F.linear(z, sd["head_in_ln.3.weight"], sd["head_in_ln.3.bias"]))

    n1, n2 = head(n1), head(n2)
    if noise_fn is None and not torch.equal(l1.unique(), l2.unique()):
        raise ValueError("a class is present in one domain only: pass noise_fn for the hallucination branch")
    if noise_fn is None:
        n1, l1 = class_first(n1, l1)
        n2, l2 = class_first(n2, l2)
    else:
        (n1, n2), (l1, l2) = class_first_complete(sd, (n1, n2), (l1, l2), noise_fn)
    n1, e1 = mha_v2(sd, "intra_domain_graph", n1, n1, n1)
    n2, e2 = mha_v2(sd, "intra_domain_graph", n2, n2, n2)
    seeds = (seed_update(sd["sr_seed"], n1, l1, with_cluster), seed_update(sd["tg_seed"], n2, l2, with_cluster))
    n2e = mha_v2(sd, "cross_domain_graph", n1, n1, n2)[0]
    n1e = mha_v2(sd, "cross_domain_graph", n2, n2, n1)[0]
    n1, n2 = n1e, n2e
    z = torch.cat([n1, n2])
    logits = F.linear(F.relu(F.linear(z, sd["node_cls_middle.0.weight"], sd["node_cls_middle.0.bias"])),
                      sd["node_cls_middle.2.weight"], sd["node_cls_middle.2.bias"])
    losses["node_loss"] = F.cross_entropy(logits, torch.cat([l1, l2]).long())
    M = affinity(sd, "node_affinity", n1, n2)
    tgt = (torch.eye(num_class)[l1.long()] @ torch.eye(num_class)[l2.long()].t())
    M = F.instance_norm(M[None, None])
    M = sinkhorn_rpm(M[:, 0], 20).squeeze().exp()
    idx = (M * (tgt == 1).float()).max(-1)[1]
    tp = M[range(M.size(0)), idx].view(-1, 1)
    fp = M[tgt == 0].view(-1, 1)

    def focal(p, t):
        return torch.mean(-0.25 * (1 - p) ** 2 * t * torch.log(p) - 0.75 * p ** 2 * (1 - t) * torch.log(1 - p))

    aff = focal(tp, torch.ones_like(tp)) / len(tp) + focal(fp, torch.zeros_like(fp)) / fp.sum().detach()
    losses["mat_loss_aff"] = 0.1 * aff
    R = e1.detach() @ M - M @ e2.detach()
    losses["mat_loss_qu"] = R.abs().mean()
    return n1, n2, losses, seeds, counts
