"""Oracle: small pieces -- GRL, losses, attention, affinity, sinkhorn.  TEST INFRASTRUCTURE ONLY."""
import torch
import torch.nn.functional as F


class _GRL(torch.autograd.Function):
    # gradient_reversal.py:15-24
    @staticmethod
    def forward(ctx, x, lam):
        ctx.lam = lam
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return -ctx.lam * g, None


def grad_reverse(x, lam):
    return _GRL.apply(x, lam)


def dice_loss(logits, target, smooth=1.0):
    """utils/losses.py:44-61,81-95: softmax over C; per channel 1-(sum pt+1)/(sum p^2+t^2+1) per sample, mean, mean."""
    p = torch.softmax(logits, dim=1)
    n = logits.shape[0]
    total = 0.0
    for c in range(target.shape[1]):
        pc, tc = p[:, c].reshape(n, -1), target[:, c].reshape(n, -1)
        num = (pc * tc).sum(1) + smooth
        den = (pc.pow(2) + tc.pow(2)).sum(1) + smooth
        total = total + (1 - num / den).mean()
    return total / target.shape[1]


def seg_loss_camus(logits, masks):
    """train_camus_echo.py:212: 0.1 * (Dice + BCE) / 2."""
    return 0.1 * (dice_loss(logits, masks) + F.binary_cross_entropy_with_logits(logits, masks)) / 2


def seg_loss_cardiac(logits, masks):
    """train_cardiac_uda.py:228: Dice + BCE over all channels."""
    return dice_loss(logits, masks) + F.binary_cross_entropy_with_logits(logits, masks)


def overlap_metrics(gt, pred, eps=1e-5):
    """train_camus_echo.py:402-417 -> (pixel_acc, dice, precision, specificity, recall)."""
    o, t = pred.reshape(-1).float(), gt.reshape(-1).float()
    tp, fp = (o * t).sum(), (o * (1 - t)).sum()
    fn, tn = ((1 - o) * t).sum(), ((1 - o) * (1 - t)).sum()
    return ((tp + tn + eps) / (tp + tn + fp + fn + eps), (2 * tp + eps) / (2 * tp + fp + fn + eps),
            (tp + eps) / (tp + fp + eps), (tn + eps) / (tn + fp + eps), (tp + eps) / (tp + fn + eps))


def layer_norm(x, w=None, b=None):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def mha_v2(sd, pre, key, value, query, dropout_p=0.0):
    """MultiHeadAttention(version='v2', 1 head).forward(key, value, query) -- transformer.py:43-74,110.

    scale = (dim_per_head // num_heads) ** -0.5; returns (LayerNorm(query + out), attention)."""
    d = sd[pre + ".linear_k.weight"].shape[0]
    k = F.linear(key, sd[pre + ".linear_k.weight"], sd[pre + ".linear_k.bias"])
    v = F.linear(value, sd[pre + ".linear_v.weight"], sd[pre + ".linear_v.bias"])
    q = F.linear(query, sd[pre + ".linear_q.weight"], sd[pre + ".linear_q.bias"])
    att = torch.softmax((q @ k.t()) * (d ** -0.5), dim=-1)
    att = F.dropout(att, dropout_p, dropout_p > 0)
    ctx = att @ v
    out = F.linear(ctx, sd[pre + ".linear_final.weight"], sd[pre + ".linear_final.bias"])
    out = F.dropout(out, dropout_p, dropout_p > 0)
    out = F.layer_norm(query + out, (d,), sd[pre + ".layer_norm.weight"], sd[pre + ".layer_norm.bias"], 1e-5)
    return out, att


def affinity(sd, pre, X, Y):
    """Affinity.forward (affinity_layer.py:52-73), evaluated the reference's way: broadcast-concat + MLP."""
    X = F.linear(X, sd[pre + ".project_sr.weight"])
    Y = F.linear(Y, sd[pre + ".project_tg.weight"])
    n1, n2 = X.shape[0], Y.shape[0]
    M = torch.cat([X.unsqueeze(1).expand(n1, n2, -1), Y.unsqueeze(0).expand(n1, n2, -1)], dim=-1)
    M = F.relu(F.linear(M, sd[pre + ".fc_M.0.weight"], sd[pre + ".fc_M.0.bias"]))
    return F.linear(M, sd[pre + ".fc_M.2.weight"], sd[pre + ".fc_M.2.bias"]).squeeze(-1)


def sinkhorn_rpm(log_alpha, n_iters=5):
    """GModule.sinkhorn_rpm(slack=True) -- graph_matching.py:653-676, literal pad / normalise / slice form."""
    x = F.pad(log_alpha, (0, 1, 0, 1))
    for _ in range(n_iters):
        x = torch.cat((x[:, :-1, :] - torch.logsumexp(x[:, :-1, :], dim=2, keepdim=True), x[:, -1:, :]), dim=1)
        x = torch.cat((x[:, :, :-1] - torch.logsumexp(x[:, :, :-1], dim=1, keepdim=True), x[:, :, -1:]), dim=2)
    return x[:, :-1, :-1]


def sinkhorn_distance(x, y, eps, max_iter, reduction="none", thresh=0.1):
    """SinkhornDistance.forward -- sinkhorn_distance.py:27-86.  Returns (cost, pi, C, iterations_run)."""
    C = ((x.unsqueeze(-2) - y.unsqueeze(-3)).abs() ** 2).sum(-1)
    p1, p2 = x.shape[-2], y.shape[-2]
    bs = 1 if x.dim() == 2 else x.shape[0]
    mu = torch.full((bs, p1), 1.0 / p1).squeeze()
    nu = torch.full((bs, p2), 1.0 / p2).squeeze()
    u, v = torch.zeros_like(mu), torch.zeros_like(nu)

    def M(u, v):
        return (-C + u.unsqueeze(-1) + v.unsqueeze(-2)) / eps

    nits = 0
    for _ in range(max_iter):
        u1 = u
        u = eps * (torch.log(mu + 1e-8) - torch.logsumexp(M(u, v), dim=-1)) + u
        v = eps * (torch.log(nu + 1e-8) - torch.logsumexp(M(u, v).transpose(-2, -1), dim=-1)) + v
        nits += 1
        if (u - u1).abs().sum(-1).mean().item() < thresh:
            break
    pi = torch.exp(M(u, v))
    cost = (pi * C).sum((-2, -1))
    if reduction == "mean":
        cost = cost.mean()
    elif reduction == "sum":
        cost = cost.sum()
    return cost, pi, C, nits
