"""Oracle: Grapher / MRConv / dynamic graph conv as pure functions of a state_dict (torch CPU + the C k-NN).

Follows reference models/vig.py: MRConv2d :96-105, BasicConv :476-488 (groups=4), DyGraphConv2d :196-206,
Grapher :422-430, FFN :540-546; TGCN's DyGraphConv2d models/TGCN.py:62-78.  TEST INFRASTRUCTURE ONLY.
k-NN indices come from oracle/knn_ref.c (pinned arithmetic order, lowest-index tie-break).
"""
import torch
import torch.nn.functional as F

from .knn import knn_graph


def _bn(sd, pre, x, training):
    return F.batch_norm(x, sd[pre + ".running_mean"].clone(), sd[pre + ".running_var"].clone(), sd[pre + ".weight"],
                        sd[pre + ".bias"], training, 0.1, 1e-5)


def _act(x, act):
    return {"relu": F.relu, "gelu": F.gelu}[act](x)


def edge_index(x, y, k, dilation, relative_pos=None):
    e = knn_graph(x.detach().numpy(), None if y is None else y.detach().numpy(), k, dilation,
                  None if relative_pos is None else relative_pos.detach().numpy(), True)
    return torch.from_numpy(e)


def mr_conv(sd, pre, x, edge, y, act, norm_bn, training):
    """x (B,C,N,1); gather x_j from y (or x), x_i from x; max over k of (x_j - x_i); interleave; grouped 1x1 conv."""
    B, C, N, _ = x.shape
    src = x if y is None else y
    bi = torch.arange(B).view(B, 1, 1, 1)
    ci = torch.arange(C).view(1, C, 1, 1)
    xj = src[:, :, :, 0][bi, ci, edge[0].unsqueeze(1)]
    xi = x[:, :, :, 0][bi, ci, edge[1].unsqueeze(1)]
    m = (xj - xi).max(-1, keepdim=True)[0]
    z = torch.cat([x.unsqueeze(2), m.unsqueeze(2)], dim=2).reshape(B, 2 * C, N, 1)
    z = F.conv2d(z, sd[pre + ".nn.0.weight"], sd.get(pre + ".nn.0.bias"), groups=4)
    if norm_bn:
        z = _bn(sd, pre + ".nn.1", z, training)
    return _act(z, act)


def grapher_forward(sd, pre, x, k=9, dilation=1, r=1, act="gelu", norm_bn=True, training=True):
    """Grapher(C, k, dilation, 'mr', act, 'batch', r=r).forward(x) -- vig.py:422-430."""
    p = (pre + ".") if pre else ""
    short = x
    h = _bn(sd, p + "fc1.1", F.conv2d(x, sd[p + "fc1.0.weight"], sd[p + "fc1.0.bias"]), training)
    B, C, H, W = h.shape
    y = None
    if r > 1:
        y = F.avg_pool2d(h, r, r).reshape(B, C, -1, 1)
    hn = h.reshape(B, C, -1, 1)
    e = edge_index(hn, y, k, dilation)
    g = mr_conv(sd, p + "graph_conv.gconv", hn, e, y, act, norm_bn, training).reshape(B, -1, H, W)
    out = _bn(sd, p + "fc2.1", F.conv2d(g, sd[p + "fc2.0.weight"], sd[p + "fc2.0.bias"]), training)
    return out + short
