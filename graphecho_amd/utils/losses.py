"""Segmentation losses with the reference's contract (utils/losses.py:7-95): ``make_one_hot``, ``BinaryDiceLoss``
(per-sample ``1 - (sum p t + s) / (sum p^q + t^q + s)``) and ``DiceLoss`` (channel softmax, mean over channels of the
binary Dice terms).  The configuration the trainers use -- DiceLoss() with defaults -- is one fused HIP reduction
(`ge_dice_fwd/bwd`: softmax + the three per-(sample, channel) sums in one pass); every other configuration is
evaluated for all channels at once from the same softmax kernel.
"""
import torch
import torch.nn as nn

from .. import functional as GF

_REDUCERS = {"mean": lambda v: v.mean(), "sum": lambda v: v.sum(), "none": lambda v: v}


def _reduce(values, how):
    try:
        return _REDUCERS[how](values)
    except KeyError:
        raise Exception(f"Unexpected reduction {how}") from None


def _dice_per_sample(prob, target, smooth, power):
    """prob / target: (N, K) flattened per sample -> (N,) Dice losses."""
    overlap = (prob * target).sum(1)
    mass = (prob ** power + target ** power).sum(1)
    return 1 - (overlap + smooth) / (mass + smooth)


def make_one_hot(input, num_classes):
    """(N, 1, *) integer class map -> (N, num_classes, *) one-hot float tensor, on the input's device."""
    size = list(input.shape)
    size[1] = num_classes
    return torch.zeros(size, device=input.device).scatter_(1, input, 1)


class BinaryDiceLoss(nn.Module):
    def __init__(self, smooth=1, p=2, reduction="mean"):
        super().__init__()
        self.smooth, self.p, self.reduction = smooth, p, reduction

    def forward(self, predict, target):
        n = predict.shape[0]
        assert n == target.shape[0], "predict & target batch size don't match"
        per_sample = _dice_per_sample(predict.reshape(n, -1), target.reshape(n, -1), self.smooth, self.p)
        return _reduce(per_sample, self.reduction)


class DiceLoss(nn.Module):
    def __init__(self, weight=None, ignore_index=None, **kwargs):
        super().__init__()
        self.kwargs, self.weight, self.ignore_index = kwargs, weight, ignore_index

    def _is_default(self):
        kw = self.kwargs
        return self.weight is None and self.ignore_index is None and kw.get("p", 2) == 2 and \
            kw.get("reduction", "mean") == "mean"

    def forward(self, predict, target):
        assert predict.shape == target.shape, "predict & target shape do not match"
        if predict.dim() > 3 and self._is_default():
            return GF.dice_loss(predict, target.to(predict.dtype), float(self.kwargs.get("smooth", 1)))
        return self._all_channels(predict, target)

    def _all_channels(self, predict, target):
        """Generic configuration (and the un-batched (C, H, W) call of the temporal branch, whose "channel" axis is
        dim 1 exactly as in the reference): every channel's binary Dice from one softmax, weights / ignore_index as
        column masks."""
        kw = self.kwargs
        channels = target.shape[1]
        prob = GF.softmax_lastdim(predict.movedim(1, -1)).movedim(-1, 1)
        n = predict.shape[0]
        p2 = prob.reshape(n, channels, -1).transpose(0, 1)          # (C, N, K)
        t2 = target.to(prob.dtype).reshape(n, channels, -1).transpose(0, 1)
        per_channel = []
        for c in range(channels):
            if c == self.ignore_index:
                continue
            term = _reduce(_dice_per_sample(p2[c], t2[c], kw.get("smooth", 1), kw.get("p", 2)),
                           kw.get("reduction", "mean"))
            if self.weight is not None:
                assert self.weight.shape[0] == channels, f"Expect weight shape [{channels}], get[{self.weight.shape[0]}]"
                term = term * self.weight[c]
            per_channel.append(term)
        return sum(per_channel) / channels
