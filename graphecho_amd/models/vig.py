"""Vision-GNN building blocks (reference models/vig.py) on the gfx950 Grapher kernels.

Hot pieces -- the ones the reference's training path reaches through models/TGCN.py:7 and the C2 benchmark --
are single fused kernels here:
  * ``DenseDilatedKnnGraph``: L2-normalise + fp32-MFMA distance tiles + per-row wavefront top-k, never
    materialising the (B, N, M) distance matrix (vig.py:232-381);
  * ``MRConv2d``: neighbour gather + max-relative + channel-interleaved concat in one pass (vig.py:88-105),
    followed by the grouped 1x1 conv of ``BasicConv`` (vig.py:476-488) on the implicit-GEMM conv kernel.
Edge index contract is the reference's: int64 (2, B, N, k), [0] neighbour ids nearest-first, [1] centre ids.
Tie order (unspecified for torch.topk) is defined as lowest index first.

Deviations, on purpose: ``DyGraphConv2d.forward`` does not print the edge-index shape every call (vig.py:204);
``MLP`` (vig.py:464-473, references an undefined ``Lin``) is not provided; pretrained weights cannot be fetched.
"""

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Sequential as Seq

from .. import functional as GF
from .. import nn as gnn

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


class DropPath(nn.Module):
    """Stochastic depth per sample (the one timm symbol the reference instantiates, vig.py:404,538)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


# ---- relative position embedding (vig.py:21-85), init-time numpy only ----------------------------------------
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    assert embed_dim % 2 == 0
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0))
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed_from_grid(embed_dim, grid):
    assert embed_dim % 2 == 0
    return np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                           get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    axis = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(axis, axis), axis=0).reshape([2, 1, grid_size, grid_size])  # w first
    pos_embed = get_2d_sincos_pos_embed_from_grid(embed_dim, grid)
    if cls_token:
        pos_embed = np.concatenate([np.zeros([1, embed_dim]), pos_embed], axis=0)
    return pos_embed


def get_2d_relative_pos_embed(embed_dim, grid_size):
    pos_embed = get_2d_sincos_pos_embed(embed_dim, grid_size)
    return 2 * np.matmul(pos_embed, pos_embed.transpose()) / pos_embed.shape[1]


# ---- graph construction ---------------------------------------------------------------------------------------
def batched_index_select(x, idx):
    """(B, C, M, 1), (B, N, k) -> (B, C, N, k) neighbour features (vig.py:209-229): gather / scatter-add kernels.
    The non-MR graph convs use it; MRConv2d fuses gather + max into one kernel instead."""
    return GF.edge_gather(x, idx)


def _bmm_nt(a, b):
    return torch.stack([GF.matmul(a[i], b[i], False, True) for i in range(a.shape[0])])


def pairwise_distance(x):
    """(B, N, C) -> (B, N, N) squared distances, ||x||^2 - 2 x x^T + ||x||^2^T (vig.py:232-243)."""
    with torch.no_grad():
        sq = torch.sum(x * x, dim=-1, keepdim=True)
        return sq + (-2 * _bmm_nt(x, x)) + sq.transpose(2, 1)


def part_pairwise_distance(x, start_idx=0, end_idx=1):
    with torch.no_grad():
        part = x[:, start_idx:end_idx]
        sq_part = torch.sum(part * part, dim=-1, keepdim=True)
        sq = torch.sum(x * x, dim=-1, keepdim=True)
        return sq_part + (-2 * _bmm_nt(part.contiguous(), x)) + sq.transpose(2, 1)


def xy_pairwise_distance(x, y):
    with torch.no_grad():
        sx = torch.sum(x * x, dim=-1, keepdim=True)
        sy = torch.sum(y * y, dim=-1, keepdim=True)
        return sx + (-2 * _bmm_nt(x, y)) + sy.transpose(2, 1)


def dense_knn_matrix(x, k=16, relative_pos=None):
    """x (B, C, N, 1) -> edge_index (2, B, N, k) of the self-graph (vig.py:277-309)."""
    return GF.knn_graph(x, None, k, 1, relative_pos, normalize=False)


def xy_dense_knn_matrix(x, y, k=16, relative_pos=None):
    """x (B, C, N, 1), y (B, C, M, 1) -> edge_index (2, B, N, k) into y (vig.py:312-329)."""
    return GF.knn_graph(x, y, k, 1, relative_pos, normalize=False)


class DenseDilated(nn.Module):
    """Keep every `dilation`-th neighbour of a (2, B, N, k*dilation) list (vig.py:332-354)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation = dilation
        self.stochastic = stochastic
        self.epsilon = epsilon
        self.k = k

    def forward(self, edge_index):
        if self.stochastic and torch.rand(1) < self.epsilon and self.training:
            pick = torch.randperm(self.k * self.dilation)[:self.k]
            return edge_index[:, :, :, pick]
        return edge_index[:, :, :, ::self.dilation]


class DenseDilatedKnnGraph(nn.Module):
    """L2-normalise over channels, then dilated k-NN (vig.py:357-381) -- one fused kernel pair."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation = dilation
        self.stochastic = stochastic
        self.epsilon = epsilon
        self.k = k
        self._dilated = DenseDilated(k, dilation, stochastic, epsilon)

    def forward(self, x, y=None, relative_pos=None):
        if self.stochastic:
            full = GF.knn_graph(x, y, self.k * self.dilation, 1, relative_pos, normalize=True)
            return self._dilated(full)
        return GF.knn_graph(x, y, self.k, self.dilation, relative_pos, normalize=True)


# ---- layers ---------------------------------------------------------------------------------------------------
def act_layer(act, inplace=False, neg_slope=0.2, n_prelu=1):
    act = act.lower()
    if act == "relu":
        return gnn.ReLU(inplace)
    if act == "gelu":
        return gnn.GELU()
    if act == "leakyrelu":
        return gnn.LeakyReLU(neg_slope, inplace)
    if act == "prelu":
        return gnn.PReLU(num_parameters=n_prelu, init=neg_slope)
    if act == "hswish":
        return gnn.Hardswish(inplace)
    raise NotImplementedError("activation layer [%s] is not found" % act)


class _InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(nc, affine=False): per-(n, c) plane statistics = GroupNorm with one channel per group."""

    def __init__(self, nc):
        super().__init__()
        self.nc = nc

    def forward(self, x):
        return GF.group_norm(x, self.nc, None, None, 1e-5)


def norm_layer(norm, nc):
    norm = norm.lower()
    if norm == "batch":
        return gnn.BatchNorm2d(nc, affine=True)
    if norm == "instance":
        return _InstanceNorm2d(nc)
    raise NotImplementedError("normalization layer [%s] is not found" % norm)


_FUSE_BASICCONV = __import__("os").environ.get("GE_BASICCONV_FUSE", "1") != "0"


class BasicConv(Seq):
    """Grouped (groups=4) 1x1 conv [+ norm] [+ act] [+ Dropout2d] (vig.py:476-500)."""

    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0.0):
        m = []
        for i in range(1, len(channels)):
            m.append(gnn.Conv2d(channels[i - 1], channels[i], 1, bias=bias, groups=4))
            if norm is not None and norm.lower() != "none":
                m.append(norm_layer(norm, channels[-1]))
            if act is not None and act.lower() != "none":
                m.append(act_layer(act))
            if drop > 0:
                m.append(nn.Dropout2d(drop))
        super().__init__(*m)
        self.reset_parameters()

    def forward(self, x):
        """conv -> BatchNorm -> GELU/ReLU runs as ONE fused pair: moments from the conv epilogue, the activation inside
        the BatchNorm apply kernel (and its derivative recomputed inside the BatchNorm backward kernels) -- no separate
        statistics pass, no separate activation kernels.  Any other layer sequence runs layer by layer."""
        mods = list(self)
        if not _FUSE_BASICCONV:
            for m in mods:
                x = m(x)
            return x
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, gnn.Conv2d) and isinstance(nxt, gnn.BatchNorm2d):
                act = mods[i + 2] if i + 2 < len(mods) else None
                if isinstance(act, gnn.GELU):
                    x = gnn.conv_bn(m, nxt, x, relu="gelu")
                    i += 3
                    continue
                if isinstance(act, gnn.ReLU):
                    x = gnn.conv_bn(m, nxt, x, relu=True)
                    i += 3
                    continue
                x = gnn.conv_bn(m, nxt, x)
                i += 2
                continue
            x = m(x)
            i += 1
        return x

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


class MRConv2d(nn.Module):
    """Max-relative graph conv: cat-interleave(x, max_j(x_j - x_i)) -> grouped 1x1 conv (vig.py:88-105)."""

    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super().__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index, y=None):
        return self.nn(GF.mr_aggregate(x, edge_index, y))


class EdgeConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super().__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index, y=None):
        x_i = batched_index_select(x, edge_index[1])
        x_j = batched_index_select(x if y is None else y, edge_index[0])
        return GF.neighbour_max(self.nn(torch.cat([x_i, x_j - x_i], dim=1)))


class GraphSAGE(nn.Module):
    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super().__init__()
        self.nn1 = BasicConv([in_channels, in_channels], act, norm, bias)
        self.nn2 = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index, y=None):
        x_j = batched_index_select(x if y is None else y, edge_index[0])
        x_j = GF.neighbour_max(self.nn1(x_j))
        return self.nn2(torch.cat([x, x_j], dim=1))


class GINConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super().__init__()
        self.nn = BasicConv([in_channels, out_channels], act, norm, bias)
        self.eps = nn.Parameter(torch.Tensor([0.0]))

    def forward(self, x, edge_index, y=None):
        x_j = batched_index_select(x if y is None else y, edge_index[0])
        return self.nn((1 + self.eps) * x + GF.neighbour_sum(x_j))


class GraphConv2d(nn.Module):
    """Static graph convolution layer (vig.py:163-181)."""

    def __init__(self, in_channels, out_channels, conv="edge", act="relu", norm=None, bias=True):
        super().__init__()
        if conv == "edge":
            self.gconv = EdgeConv2d(in_channels, out_channels, act, norm, bias)
        elif conv == "mr":
            self.gconv = MRConv2d(in_channels, out_channels, act, norm, bias)
        elif conv == "sage":
            self.gconv = GraphSAGE(in_channels, out_channels, act, norm, bias)
        elif conv == "gin":
            self.gconv = GINConv2d(in_channels, out_channels, act, norm, bias)
        else:
            raise NotImplementedError("conv:{} is not supported".format(conv))

    def forward(self, x, edge_index, y=None):
        return self.gconv(x, edge_index, y)


class DyGraphConv2d(GraphConv2d):
    """Dynamic graph conv: (optionally r x r average-pooled) k-NN graph rebuilt every call (vig.py:184-206)."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None,
                 bias=True, stochastic=False, epsilon=0.0, r=1):
        super().__init__(in_channels, out_channels, conv, act, norm, bias)
        self.k = kernel_size
        self.d = dilation
        self.r = r
        self.dilated_knn_graph = DenseDilatedKnnGraph(kernel_size, dilation, stochastic, epsilon)

    def forward(self, x, relative_pos=None):
        B, C, H, W = x.shape
        y = None
        if self.r > 1:
            y = GF.avg_pool2d(x, self.r).reshape(B, C, -1, 1)
        x = x.reshape(B, C, -1, 1)
        edge_index = self.dilated_knn_graph(x, y, relative_pos)
        x = super().forward(x, edge_index, y)
        return x.reshape(B, -1, H, W)


class Grapher(nn.Module):
    """fc1 -> dynamic graph conv -> fc2 -> + residual (vig.py:384-430)."""

    def __init__(self, in_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None, bias=True,
                 stochastic=False, epsilon=0.0, r=1, n=196, drop_path=0.0, relative_pos=False):
        super().__init__()
        self.channels = in_channels
        self.n = n
        self.r = r
        self.fc1 = nn.Sequential(gnn.Conv2d(in_channels, in_channels, 1, stride=1, padding=0),
                                 gnn.BatchNorm2d(in_channels))
        self.graph_conv = DyGraphConv2d(in_channels, in_channels * 2, kernel_size, dilation, conv, act, norm, bias,
                                        stochastic, epsilon, r)
        self.fc2 = nn.Sequential(gnn.Conv2d(in_channels * 2, in_channels, 1, stride=1, padding=0),
                                 gnn.BatchNorm2d(in_channels))
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.relative_pos = None
        if relative_pos:
            rel = torch.from_numpy(np.float32(get_2d_relative_pos_embed(in_channels, int(n ** 0.5))))
            rel = F.interpolate(rel.unsqueeze(0).unsqueeze(1), size=(n, n // (r * r)), mode="bicubic",
                                align_corners=False)
            self.relative_pos = nn.Parameter(-rel.squeeze(1), requires_grad=False)

    def _get_relative_pos(self, relative_pos, H, W):
        if relative_pos is None or H * W == self.n:
            return relative_pos
        N = H * W
        return F.interpolate(relative_pos.unsqueeze(0), size=(N, N // (self.r * self.r)), mode="bicubic").squeeze(0)

    def forward(self, x):
        # x feeds fc1 and the residual: the residual reads the alias fc1 returns, so its gradient is added in fc1's
        # data-gradient epilogue instead of by a separate tensor add
        x, shortcut = gnn.conv_bn(self.fc1[0], self.fc1[1], x, with_skip=True)
        B, C, H, W = x.shape
        x = self.graph_conv(x, self._get_relative_pos(self.relative_pos, H, W))
        if isinstance(self.drop_path, nn.Identity):
            return gnn.conv_bn(self.fc2[0], self.fc2[1], x, residual=shortcut)  # BN + residual add in one pass
        return self.drop_path(gnn.conv_bn(self.fc2[0], self.fc2[1], x)) + shortcut


class FFN(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act="relu", drop_path=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Sequential(gnn.Conv2d(in_features, hidden_features, 1, stride=1, padding=0),
                                 gnn.BatchNorm2d(hidden_features))
        self.act = act_layer(act)
        self.fc2 = nn.Sequential(gnn.Conv2d(hidden_features, out_features, 1, stride=1, padding=0),
                                 gnn.BatchNorm2d(out_features))
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x):
        act = True if isinstance(self.act, gnn.ReLU) else ("gelu" if isinstance(self.act, gnn.GELU) else False)
        x, shortcut = gnn.conv_bn(self.fc1[0], self.fc1[1], x, relu=act, with_skip=True)
        if act is False:
            x = self.act(x)
        if isinstance(self.drop_path, nn.Identity):
            return gnn.conv_bn(self.fc2[0], self.fc2[1], x, residual=shortcut)
        return self.drop_path(gnn.conv_bn(self.fc2[0], self.fc2[1], x)) + shortcut


class _ConvBNAct(nn.Sequential):
    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            if isinstance(mods[i], gnn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], gnn.BatchNorm2d):
                nxt = mods[i + 2] if i + 2 < len(mods) else None
                act = True if isinstance(nxt, gnn.ReLU) else ("gelu" if isinstance(nxt, gnn.GELU) else False)
                x = gnn.conv_bn(mods[i], mods[i + 1], x, relu=act)     # moments from the conv epilogue, fused activation
                i += 3 if act else 2
            else:
                x = mods[i](x)
                i += 1
        return x


class Stem(nn.Module):
    """Image -> visual embedding: three 3x3 convs, overall stride 4 (vig.py:549-568)."""

    def __init__(self, img_size=224, in_dim=3, out_dim=768, act="relu"):
        super().__init__()
        self.convs = _ConvBNAct(
            gnn.Conv2d(in_dim, out_dim // 2, 3, stride=2, padding=1), gnn.BatchNorm2d(out_dim // 2), act_layer(act),
            gnn.Conv2d(out_dim // 2, out_dim, 3, stride=2, padding=1), gnn.BatchNorm2d(out_dim), act_layer(act),
            gnn.Conv2d(out_dim, out_dim, 3, stride=1, padding=1), gnn.BatchNorm2d(out_dim))

    def forward(self, x):
        return self.convs(x)


class Downsample(nn.Module):
    def __init__(self, in_dim=3, out_dim=768):
        super().__init__()
        self.conv = _ConvBNAct(gnn.Conv2d(in_dim, out_dim, 3, stride=2, padding=1), gnn.BatchNorm2d(out_dim))

    def forward(self, x):
        return self.conv(x)


class DeepGCN(nn.Module):
    """Pyramid ViG classifier (vig.py:586-651)."""

    def __init__(self, opt):
        super().__init__()
        k, act, norm, bias = opt.k, opt.act, opt.norm, opt.bias
        epsilon, stochastic, conv = opt.epsilon, opt.use_stochastic, opt.conv
        blocks, channels = opt.blocks, opt.channels
        self.n_blocks = sum(blocks)
        reduce_ratios = [4, 2, 1, 1]
        dpr = [x.item() for x in torch.linspace(0, opt.drop_path, self.n_blocks)]
        num_knn = [int(x.item()) for x in torch.linspace(k, k, self.n_blocks)]
        max_dilation = 49 // max(num_knn)
        self.stem = Stem(out_dim=channels[0], act=act)
        self.pos_embed = nn.Parameter(torch.zeros(1, channels[0], 224 // 4, 224 // 4))
        HW = 224 // 4 * 224 // 4
        backbone = []
        idx = 0
        for i in range(len(blocks)):
            if i > 0:
                backbone.append(Downsample(channels[i - 1], channels[i]))
                HW = HW // 4
            for _ in range(blocks[i]):
                backbone.append(Seq(
                    Grapher(channels[i], num_knn[idx], min(idx // 4 + 1, max_dilation), conv, act, norm, bias,
                            stochastic, epsilon, reduce_ratios[i], n=HW, drop_path=dpr[idx], relative_pos=True),
                    FFN(channels[i], channels[i] * 4, act=act, drop_path=dpr[idx])))
                idx += 1
        self.backbone = Seq(*backbone)
        self.prediction = Seq(gnn.Conv2d(channels[-1], 1024, 1, bias=True), gnn.BatchNorm2d(1024), act_layer(act),
                              nn.Dropout(opt.dropout), gnn.Conv2d(1024, opt.n_classes, 1, bias=True))
        self.model_init()

    def model_init(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                m.weight.requires_grad = True
                if m.bias is not None:
                    m.bias.data.zero_()
                    m.bias.requires_grad = True

    def forward(self, inputs):
        x = self.stem(inputs) + self.pos_embed
        for blk in self.backbone:
            x = blk(x)
        x = GF.adaptive_avg_pool2d_1(x)
        return self.prediction(x).squeeze(-1).squeeze(-1)


class _PvigOpt:
    def __init__(self, blocks, channels, num_classes=1000, drop_path_rate=0.0, **kwargs):
        self.k = 9
        self.conv = "mr"
        self.act = "gelu"
        self.norm = "batch"
        self.bias = True
        self.dropout = 0.0
        self.use_dilation = True
        self.epsilon = 0.2
        self.use_stochastic = False
        self.drop_path = drop_path_rate
        self.blocks = blocks
        self.channels = channels
        self.n_classes = num_classes
        self.emb_dims = 1024


def _cfg(**kwargs):
    cfg = {"url": "", "num_classes": 1000, "input_size": (3, 224, 224), "pool_size": None, "crop_pct": 0.9,
           "interpolation": "bicubic", "mean": IMAGENET_DEFAULT_MEAN, "std": IMAGENET_DEFAULT_STD,
           "first_conv": "patch_embed.proj", "classifier": "head"}
    cfg.update(kwargs)
    return cfg


default_cfgs = {
    "vig_224_gelu": _cfg(mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)),
    "vig_b_224_gelu": _cfg(crop_pct=0.95, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)),
}


def _pvig(blocks, channels, cfg, **kwargs):
    model = DeepGCN(_PvigOpt(blocks, channels, **kwargs))
    model.default_cfg = default_cfgs[cfg]
    return model


def pvig_ti_224_gelu(pretrained=False, **kwargs):
    return _pvig([2, 2, 6, 2], [48, 96, 240, 384], "vig_224_gelu", **kwargs)


def pvig_s_224_gelu(pretrained=False, **kwargs):
    return _pvig([2, 2, 6, 2], [80, 160, 400, 640], "vig_224_gelu", **kwargs)


def pvig_m_224_gelu(pretrained=False, **kwargs):
    return _pvig([2, 2, 16, 2], [96, 192, 384, 768], "vig_224_gelu", **kwargs)


def pvig_b_224_gelu(pretrained=False, **kwargs):
    return _pvig([2, 2, 18, 2], [128, 256, 512, 1024], "vig_b_224_gelu", **kwargs)
