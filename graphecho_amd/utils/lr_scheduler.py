"""WarmupMultiStepLR (reference utils/lr_scheduler.py:9-50): linear/constant warm-up times a step decay."""
from bisect import bisect_right

import torch


class WarmupMultiStepLR(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500,
                 warmup_method="linear", last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        if warmup_method not in ("constant", "linear"):
            raise ValueError("Only 'constant' or 'linear' warmup_method accepted, got {}".format(warmup_method))
        self.milestones = milestones
        self.gamma = gamma
        self.warmup_factor = warmup_factor
        self.warmup_iters = warmup_iters
        self.warmup_method = warmup_method
        super().__init__(optimizer, last_epoch)

    def lr_factor(self, epoch):
        warm = 1.0
        if epoch < self.warmup_iters:
            if self.warmup_method == "constant":
                warm = self.warmup_factor
            else:
                alpha = float(epoch) / self.warmup_iters
                warm = self.warmup_factor * (1 - alpha) + alpha
        return warm * self.gamma ** bisect_right(self.milestones, epoch)

    def get_lr(self):
        f = self.lr_factor(self.last_epoch)
        return [base_lr * f for base_lr in self.base_lrs]
