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


def _act(x, act, prelu_weight=None):
    """act_layer (vig.py:433-450): relu, gelu (erf), leakyrelu(0.2), prelu (one learned slope), hswish."""
    if act == "prelu":
        return F.prelu(x, prelu_weight)
    return {"relu": F.relu, "gelu": F.gelu, "leakyrelu": lambda t: F.leaky_relu(t, 0.2),
            "hswish": F.hardswish}[act](x)


def basic_conv(sd, pre, z, act, norm, training=True):
    """BasicConv([cin, cout], act, norm) (vig.py:476-488): grouped(4) 1x1 conv, optional BatchNorm, activation."""
    z = F.conv2d(z, sd[pre + ".0.weight"], sd.get(pre + ".0.bias"), groups=4)
    i = 1
    if norm == "batch":
        z = _bn(sd, f"{pre}.{i}", z, training)
        i += 1
    return _act(z, act, sd.get(f"{pre}.{i}.weight"))


def _gather(src, idx):
    """batched_index_select (vig.py:209-229): src (B,C,M,1), idx (B,N,K) -> (B,C,N,K)."""
    B, C = src.shape[:2]
    bi = torch.arange(B).view(B, 1, 1, 1)
    ci = torch.arange(C).view(1, C, 1, 1)
    return src[:, :, :, 0][bi, ci, idx.unsqueeze(1)]


def graph_conv(sd, pre, conv, x, edge, y, act, norm):
    """GraphConv2d.forward for conv in {edge, sage, gin, mr} (vig.py:88-181), train mode."""
    src = x if y is None else y
    p = pre + ".gconv"
    if conv == "mr":
        return mr_conv(sd, p, x, edge, y, act, norm == "batch", True)
    x_j = _gather(src, edge[0])
    if conv == "edge":
        x_i = _gather(x, edge[1])
        return basic_conv(sd, p + ".nn", torch.cat([x_i, x_j - x_i], dim=1), act, norm).max(-1, keepdim=True)[0]
    if conv == "sage":
        x_j = basic_conv(sd, p + ".nn1", x_j, act, norm).max(-1, keepdim=True)[0]
        return basic_conv(sd, p + ".nn2", torch.cat([x, x_j], dim=1), act, norm)
    if conv == "gin":
        return basic_conv(sd, p + ".nn", (1 + sd[p + ".eps"]) * x + x_j.sum(-1, keepdim=True), act, norm)
    raise NotImplementedError(conv)


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
    return _act(z, act, sd.get(pre + (".nn.2.weight" if norm_bn else ".nn.1.weight")))


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


def _grapher_rel(sd, pre, x, k, dilation, r, act):
    """Grapher with a relative-position bias on the distances (vig.py:405-430): `relative_pos` is a frozen
    parameter of the block, used as stored when H*W equals the block's design size (always true at 224x224)."""
    p = pre + "."
    short = x
    h = _bn(sd, p + "fc1.1", F.conv2d(x, sd[p + "fc1.0.weight"], sd[p + "fc1.0.bias"]), True)
    B, C, H, W = h.shape
    y = F.avg_pool2d(h, r, r).reshape(B, C, -1, 1) if r > 1 else None
    hn = h.reshape(B, C, -1, 1)
    rel = sd[p + "relative_pos"]
    assert rel.shape[1] == H * W, "oracle restates the design-size case only (vig.py:416-420 interpolates otherwise)"
    e = edge_index(hn, y, k, dilation, rel)
    g = mr_conv(sd, p + "graph_conv.gconv", hn, e, y, act, True, True).reshape(B, -1, H, W)
    return _bn(sd, p + "fc2.1", F.conv2d(g, sd[p + "fc2.0.weight"], sd[p + "fc2.0.bias"]), True) + short


def _ffn(sd, pre, x, act):
    """FFN (vig.py:540-546): 1x1 conv + BN, act, 1x1 conv + BN, + shortcut."""
    p = pre + "."
    h = _act(_bn(sd, p + "fc1.1", F.conv2d(x, sd[p + "fc1.0.weight"], sd[p + "fc1.0.bias"]), True), act)
    return _bn(sd, p + "fc2.1", F.conv2d(h, sd[p + "fc2.0.weight"], sd[p + "fc2.0.bias"]), True) + x


def deepgcn_stages(sd, blocks, k=9, act="gelu"):
    """The pyramid ViG classifier as an ordered list of (tag, fn) stages, train mode, drop_path = dropout = 0
    (vig.py:586-651, Stem :549-568, Downsample :571-583).  `blocks` e.g. [2, 2, 6, 2]; widths come from the weights.
    Tags name the position in the reference's `backbone` Sequential so tests can drive one stage at a time."""
    def conv_bn(pre, i, h, stride):
        h = F.conv2d(h, sd[f"{pre}.{i}.weight"], sd[f"{pre}.{i}.bias"], stride=stride, padding=1)
        return _bn(sd, f"{pre}.{i + 1}", h, True)

    def stem(x):
        h = _act(conv_bn("stem.convs", 0, x, 2), act)
        h = _act(conv_bn("stem.convs", 3, h, 2), act)
        return conv_bn("stem.convs", 6, h, 1) + sd["pos_embed"]

    def head(h):
        h = F.adaptive_avg_pool2d(h, 1)
        h = F.conv2d(h, sd["prediction.0.weight"], sd["prediction.0.bias"])
        h = _act(_bn(sd, "prediction.1", h, True), act)
        return F.conv2d(h, sd["prediction.4.weight"], sd["prediction.4.bias"]).squeeze(-1).squeeze(-1)

    stages = [("stem+pos", stem)]
    max_dilation = 49 // k
    reduce_ratios = [4, 2, 1, 1]
    idx = 0      # running Grapher index (sets the dilation)
    pos = 0      # position in the `backbone` Sequential
    for stage, nb in enumerate(blocks):
        if stage > 0:
            stages.append((f"backbone.{pos}", lambda h, pos=pos: conv_bn(f"backbone.{pos}.conv", 0, h, 2)))
            pos += 1
        for _ in range(nb):
            d, r = min(idx // 4 + 1, max_dilation), reduce_ratios[stage]
            stages.append((f"backbone.{pos}.0", lambda h, pos=pos, d=d, r=r: _grapher_rel(sd, f"backbone.{pos}.0", h, k, d, r, act)))
            stages.append((f"backbone.{pos}.1", lambda h, pos=pos: _ffn(sd, f"backbone.{pos}.1", h, act)))
            idx += 1
            pos += 1
    stages.append(("prediction", head))
    return stages


def deepgcn_forward(sd, x, blocks, k=9, act="gelu", taps=None):
    """DeepGCN.forward (vig.py:643-651); `taps`, if a list, receives (tag, stage input, stage output)."""
    h = x
    for tag, fn in deepgcn_stages(sd, blocks, k, act):
        out = fn(h)
        if taps is not None:
            taps.append((tag, h, out))
        h = out
    return h
