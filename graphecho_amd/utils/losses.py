"""Segmentation losses (reference utils/losses.py:24-95) on the fused softmax+Dice reduction kernel."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as GF


def make_one_hot(input, num_classes):
    """Class-index tensor [N, 1, *] -> one-hot [N, num_classes, *] (losses.py:7-21)."""
    shape = list(input.shape)
    shape[1] = num_classes
    result = torch.zeros(tuple(shape), device=input.device)
    return result.scatter_(1, input, 1)


class BinaryDiceLoss(nn.Module):
    """1 - (sum(p*t) + smooth) / (sum(p^p + t^p) + smooth) per sample, then `reduction` over the batch."""

    def __init__(self, smooth=1, p=2, reduction="mean"):
        super().__init__()
        self.smooth = smooth
        self.p = p
        self.reduction = reduction

    def forward(self, predict, target):
        assert predict.shape[0] == target.shape[0], "predict & target batch size don't match"
        predict = predict.contiguous().view(predict.shape[0], -1)
        target = target.contiguous().view(target.shape[0], -1)
        num = torch.sum(predict * target, dim=1) + self.smooth
        den = torch.sum(predict.pow(self.p) + target.pow(self.p), dim=1) + self.smooth
        loss = 1 - num / den
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        if self.reduction == "none":
            return loss
        raise Exception("Unexpected reduction {}".format(self.reduction))


class DiceLoss(nn.Module):
    """Softmax over channels, mean of the per-channel binary Dice losses (losses.py:81-95).

    The default configuration (no weight, no ignore_index, smooth=1, p=2, mean) runs as one fused HIP
    reduction; other configurations use the generic per-channel path on the same softmax kernel output.
    """

    def __init__(self, weight=None, ignore_index=None, **kwargs):
        super().__init__()
        self.kwargs = kwargs
        self.weight = weight
        self.ignore_index = ignore_index

    def forward(self, predict, target):
        assert predict.shape == target.shape, "predict & target shape do not match"
        kw = self.kwargs
        if predict.dim() == 3:  # un-batched (C, H, W) call made by the temporal branch: channels are dim 1 of (C,H,W)
            return self._generic(predict, target)
        fused = self.weight is None and self.ignore_index is None and kw.get("p", 2) == 2 and \
            kw.get("reduction", "mean") == "mean"
        if fused:
            return GF.dice_loss(predict, target.to(predict.dtype), float(kw.get("smooth", 1)))
        return self._generic(predict, target)

    def _generic(self, predict, target):
        dice = BinaryDiceLoss(**self.kwargs)
        total = 0
        prob = GF.softmax_lastdim(predict.movedim(1, -1)).movedim(-1, 1)
        for i in range(target.shape[1]):
            if i != self.ignore_index:
                d = dice(prob[:, i], target[:, i])
                if self.weight is not None:
                    assert self.weight.shape[0] == target.shape[1]
                    d = d * self.weight[i]
                total = total + d
        return total / target.shape[1]
