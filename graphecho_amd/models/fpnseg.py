"""FPN segmentation network + adversarial Discriminator on the gfx950 kernels.

Drop-in for the reference's ``models/fpnseg.py`` (FPN :309-444, ResNet/Bottleneck :177-298, VGG16 :18-166,
Discriminator :447-511): same constructor signatures, same ``forward`` contract
``FPN(x) -> (logits, [p2, p3, p4, p5])``, same module tree and therefore the same ``state_dict`` keys and the
same RNG consumption at construction (identical seeds give identical initial weights).

What differs is *how* it runs: convolutions are fp32-MFMA implicit GEMMs, BatchNorm(+residual)(+ReLU) and
GroupNorm(+ReLU) are fused normalise kernels, the top-down path uses the fused bilinear-upsample+add kernel.
Reference quirks that are kept on purpose: ``ResNet50`` has blocks [3, 4, 5, 3]; ``num_blocks`` is ignored;
``gn1``/``gn2`` have one channel per group and, like ``conv2``/``semantic_branch``, are shared across levels;
the returned pyramid is the *un-smoothed* p2..p5.
"""
import math

import torch
import torch.nn as tnn

from .. import functional as GF
from .. import half as GH
from .. import nn as gnn
from .gradient_reversal import GradientReversal

__all__ = ["FPN", "Discriminator", "ResNet", "Bottleneck", "VGG16", "ResNet50", "ResNet101"]


_H_STEM = __import__("os").environ.get("GE_H_STEM", "1") != "0"
_H_FUSE_POOL = __import__("os").environ.get("GE_H_FUSE_POOL", "1") != "0"
_H_TOWERS = __import__("os").environ.get("GE_H_TOWERS", "1") != "0"


class _ConvBNStack(tnn.Sequential):
    """[Conv, BN, ReLU] * n + MaxPool as a Sequential (reference key layout), run with BN+ReLU fused."""

    def forward(self, x):
        """x: fp32 NCHW, or a blocked fp16 tensor handed on by the previous stack (functional.ACT_STORAGE = "f16": the
        activations between the layers of a stack -- and between stacks -- stay fp16 in HBM, graphecho_amd/half.py);
        returns the same kind (VGG16.forward converts what leaves the backbone)."""
        mods = list(self)
        half_ok = GF.ACT_STORAGE == "f16" and x.is_cuda
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, gnn.Conv2d) and i + 2 < len(mods) and isinstance(mods[i + 1], gnn.BatchNorm2d) \
                    and isinstance(mods[i + 2], gnn.ReLU):
                blocked = GH.is_blocked(x)
                # a 2x2 max-pool right behind the triple rides in the BatchNorm pass (fp16 path only)
                nxt = mods[i + 3] if i + 3 < len(mods) else None
                fuse = _H_FUSE_POOL and isinstance(nxt, gnn.MaxPool2d) and nxt.kernel_size == (2, 2) and nxt.stride == (2, 2) \
                    and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                if half_ok and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.groups == 1 \
                        and GH.supported(x.shape[0], m.in_channels, m.out_channels,
                                         *(x.shape[2:4] if blocked else x.shape[2:])):
                    x = GH.conv_bn(m, mods[i + 1], x if blocked else GH.to_blocked(x), relu=True, pool=fuse)
                    i += 1 if fuse else 0
                elif half_ok and _H_STEM and GH.stem_supported(x, m):      # the 1- / 3-channel first layer: fp32 image in, fp16 out
                    x = GH.conv_bn(m, mods[i + 1], x, relu=True)
                else:
                    x = gnn.conv_bn(m, mods[i + 1], GH.from_blocked(x) if blocked else x, relu=True)
                i += 3
            elif isinstance(m, gnn.MaxPool2d) and GH.is_blocked(x) and m.kernel_size == (2, 2) and m.stride == (2, 2) \
                    and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
                x = GH.max_pool2(x)
                i += 1
            else:
                x = m(GH.from_blocked(x) if GH.is_blocked(x) else x)
                i += 1
        return x


class VGG16(tnn.Module):
    """VGG16-BN trunk; returns the five post-pool feature maps (fpnseg.py:154-166)."""

    _PLAN = ((64, 2), (128, 2), (256, 3), (512, 3), (512, 3))

    def __init__(self, in_channels):
        super().__init__()
        cin = in_channels
        for b, (width, reps) in enumerate(self._PLAN, start=1):
            layers = []
            for _ in range(reps):
                layers += [gnn.Conv2d(cin, width, kernel_size=(3, 3), stride=(1, 1), padding=1),
                           gnn.BatchNorm2d(width), gnn.ReLU()]
                cin = width
            layers.append(gnn.MaxPool2d(kernel_size=(2, 2), stride=(2, 2)))
            setattr(self, f"block_{b}", _ConvBNStack(*layers))
        for m in self.modules():
            if isinstance(m, (tnn.Conv2d, tnn.Linear)):
                tnn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="leaky_relu")
                if m.bias is not None:
                    m.bias.detach().zero_()

    def forward(self, x):
        feats = []
        for b in range(1, 6):
            x = getattr(self, f"block_{b}")(x)
            if GH.is_blocked(x):      # the fp32 copy leaves the fp16 domain, x goes on inside it: their gradients meet in fp32
                f, x = GH.fork(x) if b < 5 else (GH.from_blocked(x), x)
                feats.append(f)
            else:
                feats.append(x)
        return feats


def conv3x3(in_planes, out_planes, stride=1):
    return gnn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return gnn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class Bottleneck(tnn.Module):
    expansion = 4

    def __init__(self, in_planes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv1x1(in_planes, planes)
        self.bn1 = gnn.BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn2 = gnn.BatchNorm2d(planes)
        self.conv3 = conv1x1(planes, planes * self.expansion)
        self.bn3 = gnn.BatchNorm2d(planes * self.expansion)
        self.relu = gnn.ReLU(inplace=False)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # x feeds conv1 AND the shortcut: the shortcut's gradient is merged inside conv1's dgrad epilogue
        # every conv produces the batch statistics of its output in its own epilogue (gnn.conv_bn)
        out, skip = gnn.conv_bn(self.conv1, self.bn1, x, relu=True, with_skip=True)
        out = gnn.conv_bn(self.conv2, self.bn2, out, relu=True)
        identity = skip if self.downsample is None else gnn.conv_bn(self.downsample[0], self.downsample[1], skip)
        # bn3 + residual add + ReLU in one pass (reference: out += identity; relu, fpnseg.py:203-210)
        return gnn.conv_bn(self.conv3, self.bn3, out, relu=True, residual=identity)


class ResNet(tnn.Module):
    def __init__(self, block, layers, in_channel, pretrained=False):
        super().__init__()
        self.inplanes = 64
        self.conv1 = gnn.Conv2d(in_channel, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = gnn.BatchNorm2d(64)
        self.relu = gnn.ReLU(inplace=False)
        self.maxpool = gnn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self._init_weights()
        if pretrained:
            raise RuntimeError("pretrained ImageNet weights are not available offline; load a state_dict instead")

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = tnn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                        gnn.BatchNorm2d(planes * block.expansion))
        stages = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        stages += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return tnn.Sequential(*stages)

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, tnn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, tnn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x):
        c1 = self.maxpool(gnn.conv_bn(self.conv1, self.bn1, x, relu=True))
        c2 = self.layer1(c1)
        c3 = self.layer2(c2)
        c4 = self.layer3(c3)
        c5 = self.layer4(c4)
        return [c1, c2, c3, c4, c5]


def ResNet50(in_channel=3, pretrained=True):
    """The reference's "ResNet-50": block counts [3, 4, 5, 3] (fpnseg.py:295)."""
    if pretrained:
        raise RuntimeError("pretrained ImageNet weights are not available offline; load a state_dict instead")
    return ResNet(Bottleneck, [3, 4, 5, 3], in_channel=in_channel)


def ResNet101(in_channel=3, pretrained=True):
    return ResNet(Bottleneck, [3, 4, 23, 3], in_channel=in_channel, pretrained=pretrained)


import os as _os

_FUSE_FANOUT = _os.environ.get("GE_FUSE_FANOUT", "1") != "0"    # tuning switch: plain autograd fan-out when 0


class FPN(tnn.Module):
    def __init__(self, num_blocks, num_classes, in_channel, back_bone="resnet", pretrained=False):
        super().__init__()
        self.in_planes = 64
        self.num_classes = num_classes
        if back_bone == "resnet":
            self.back_bone = ResNet50(in_channel=in_channel, pretrained=pretrained)
            widths = (2048, 1024, 512, 256)
        elif back_bone == "VGG16":
            self.back_bone = VGG16(in_channels=in_channel)
            widths = (512, 512, 256, 128)
        else:
            raise ValueError(f"unknown back_bone {back_bone!r}")
        self.toplayer = gnn.Conv2d(widths[0], 256, kernel_size=1, stride=1, padding=0)
        self.latlayer1 = gnn.Conv2d(widths[1], 256, kernel_size=1, stride=1, padding=0)
        self.latlayer2 = gnn.Conv2d(widths[2], 256, kernel_size=1, stride=1, padding=0)
        self.latlayer3 = gnn.Conv2d(widths[3], 256, kernel_size=1, stride=1, padding=0)
        self.smooth1 = gnn.Conv2d(256, 256, kernel_size=3, stride=1, padding=1)
        self.smooth2 = gnn.Conv2d(256, 256, kernel_size=3, stride=1, padding=1)
        self.smooth3 = gnn.Conv2d(256, 256, kernel_size=3, stride=1, padding=1)
        self.semantic_branch = gnn.Conv2d(256, 128, kernel_size=3, stride=1, padding=1)
        self.conv2 = gnn.Conv2d(256, 256, kernel_size=3, stride=1, padding=1)
        self.conv3 = gnn.Conv2d(128, self.num_classes, kernel_size=1, stride=1, padding=0)
        self.gn1 = gnn.GroupNorm(128, 128)
        self.gn2 = gnn.GroupNorm(256, 256)

    def _upsample(self, x, h, w):
        return GF.upsample_bilinear(x, (h, w))

    def _upsample_add(self, x, y):
        return GF.upsample_bilinear(x, y.shape[2:], add=y)

    def forward(self, x):
        features_map, smoothed = self.forward_pyramid(x, smooth=True)
        return self.forward_head(features_map, smoothed), features_map

    def forward_pyramid(self, x, smooth=False):
        """Backbone + top-down pathway: the un-smoothed [p2, p3, p4, p5] the reference returns (fpnseg.py:405-418).

        Every map with two consumers (c2..c4: next ResNet stage + lateral conv; p2..p4: smoothing conv + next top-down
        step + the pyramid handed to GModule / discriminators / Graphers) goes through `forward_with_skip` of ONE of
        its consumer convs: the other consumers read the alias that conv returns, and the gradient arriving through the
        alias is added inside that conv's data-gradient epilogue instead of by a separate full-size tensor add.
        smooth=True also returns (smooth3(p2), smooth2(p3), smooth1(p4)) computed that way."""
        bb = self.back_bone
        if isinstance(bb, ResNet) and _FUSE_FANOUT:
            c1 = bb.maxpool(gnn.conv_bn(bb.conv1, bb.bn1, x, relu=True))
            c2 = bb.layer1(c1)
            l3, c2 = self.latlayer3.forward_with_skip(c2)
            c3 = bb.layer2(c2)
            l2, c3 = self.latlayer2.forward_with_skip(c3)
            c4 = bb.layer3(c3)
            l1, c4 = self.latlayer1.forward_with_skip(c4)
            c5 = bb.layer4(c4)
        else:
            c1, c2, c3, c4, c5 = bb(x)
            l1, l2, l3 = self.latlayer1(c4), self.latlayer2(c3), self.latlayer3(c2)
        # top-down pathway with fused upsample+lateral add
        p5 = self.toplayer(c5)
        p4 = self._upsample_add(p5, l1)
        if not smooth or not _FUSE_FANOUT:
            p3 = self._upsample_add(p4, l2)
            p2 = self._upsample_add(p3, l3)
            return ([p2, p3, p4, p5], None) if smooth else [p2, p3, p4, p5]
        s4, p4 = self.smooth1.forward_with_skip(p4)
        p3 = self._upsample_add(p4, l2)
        s3, p3 = self.smooth2.forward_with_skip(p3)
        p2 = self._upsample_add(p3, l3)
        s2, p2 = self.smooth3.forward_with_skip(p2)
        return [p2, p3, p4, p5], (s2, s3, s4)

    def forward_head(self, features_map, smoothed=None):
        """Smoothing convs + semantic branch + x4 upsample -> logits (fpnseg.py:420-444)."""
        p2, p3, p4, p5 = features_map
        if smoothed is None:
            p4 = self.smooth1(p4)
            p3 = self.smooth2(p3)
            p2 = self.smooth3(p2)
        else:
            p2, p3, p4 = smoothed

        h, w = p2.shape[2], p2.shape[3]
        up = self._upsample
        s5 = up(self.gn2(self.conv2(p5), relu=True), h, w)
        s5 = up(self.gn2(self.conv2(s5), relu=True), h, w)
        s5 = up(self.gn1(self.semantic_branch(s5), relu=True), h, w)
        s4 = up(self.gn2(self.conv2(p4), relu=True), h, w)
        s4 = up(self.gn1(self.semantic_branch(s4), relu=True), h, w)
        s3 = up(self.gn1(self.semantic_branch(p3), relu=True), h, w)
        s2 = self.gn1(self.semantic_branch(p2), relu=True)
        return up(self.conv3(s2 + s3 + s4 + s5), 4 * h, 4 * w)


class Discriminator(tnn.Module):
    def __init__(self, num_convs=4, in_channels=256, grad_reverse_lambda=-1.0, grl_applied_domain="both",
                 patch_stride=None):
        super().__init__()
        tower = []
        for _ in range(num_convs):
            tower += [gnn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1),
                      gnn.GroupNorm(32, in_channels), gnn.ReLU()]
        self.add_module("dis_tower", tnn.Sequential(*tower))
        self.cls_logits = gnn.Conv2d(in_channels, 1, kernel_size=3, stride=1, padding=1)
        self.patch_stride = patch_stride
        assert patch_stride is None or type(patch_stride) == int, "wrong format of patch stride"
        if self.patch_stride:
            self.pool = tnn.AvgPool2d(kernel_size=3, stride=patch_stride, padding=1)
        for part in (self.dis_tower, self.cls_logits):
            for layer in part.modules():
                if isinstance(layer, tnn.Conv2d):
                    tnn.init.normal_(layer.weight, std=0.01)
                    tnn.init.constant_(layer.bias, 0)
        self.grad_reverse = GradientReversal(grad_reverse_lambda)
        assert grl_applied_domain in ("both", "target")
        self.grl_applied_domain = grl_applied_domain
        self.source_label = 1.0
        self.target_label = 0.0

    def _tower(self, x):
        mods = list(self.dis_tower)
        convs, gns = mods[0::3], mods[1::3]
        # functional.ACT_STORAGE = "f16": the conv -> GroupNorm(8 channels per group) -> ReLU tower stays in the blocked fp16 domain
        # (graphecho_amd/half.py) -- one cast in, one out, instead of an fp32 round trip around every conv
        if GF.ACT_STORAGE == "f16" and _H_TOWERS and x.is_cuda and x.dim() == 4 and all(
                g.num_channels == 8 * g.num_groups and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1)
                and GH.supported(x.shape[0], c.in_channels, c.out_channels, x.shape[2], x.shape[3]) for c, g in zip(convs, gns)):
            h = GH.to_blocked(x)
            for c, g in zip(convs, gns):
                h = GH.conv_gn8(c, g, h, relu=True)
            return self.cls_logits(GH.from_blocked(h))
        for i in range(0, len(mods), 3):
            x = mods[i + 1](mods[i](x), relu=True)
        return self.cls_logits(x)

    def forward(self, feature, domain="source"):
        features_s, features_t = feature
        features_s = self.grad_reverse(features_s)
        features_t = self.grad_reverse(features_t)
        if features_s.shape[1:] == features_t.shape[1:]:
            # GroupNorm is per-sample, so one batched pass over [source; target] is exact
            ns = features_s.shape[0]
            x = self._tower(torch.cat([features_s, features_t], dim=0))
            x_s, x_t = x[:ns], x[ns:]
        else:
            x_s, x_t = self._tower(features_s), self._tower(features_t)
        return GF.bce_with_logits(x_s, self.source_label) + GF.bce_with_logits(x_t, self.target_label)
