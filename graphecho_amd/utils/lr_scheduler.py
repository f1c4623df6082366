"""Learning-rate schedule of the reference trainers (utils/lr_scheduler.py:9-50; train_camus_echo.py:437-445):
a step decay at given milestones multiplied by a warm-up ramp.  Same constructor contract, so the reference's
``WarmupMultiStepLR(optimizer, STEPS, GAMMA, warmup_factor=..., warmup_iters=..., warmup_method=...)`` call works.

The schedule is expressed as a pure function ``multiplier(t)`` of the scheduler's step counter, which the host tests
compare with the reference's formula over a sweep of t:

    decay(t)  = gamma ** (number of milestones <= t)
    warmup(t) = 1                                   for t >= warmup_iters
              = warmup_factor                       ("constant")
              = warmup_factor + (1 - warmup_factor) * t / warmup_iters      ("linear")
"""
import bisect

from torch.optim.lr_scheduler import _LRScheduler

_WARMUP_KINDS = ("constant", "linear")


def _decay(t, milestones, gamma):
    return gamma ** bisect.bisect_right(milestones, t)


def _warmup(t, kind, factor, iters):
    if t >= iters:
        return 1.0
    if kind == "constant":
        return factor
    frac = float(t) / iters
    return factor * (1 - frac) + frac


class WarmupMultiStepLR(_LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500,
                 warmup_method="linear", last_epoch=-1):
        steps = list(milestones)
        if any(b < a for a, b in zip(steps, steps[1:])):
            raise ValueError(f"milestones must be non-decreasing, got {milestones}")
        if warmup_method not in _WARMUP_KINDS:
            raise ValueError(f"warmup_method must be one of {_WARMUP_KINDS}, got {warmup_method!r}")
        self.milestones, self.gamma = milestones, gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def multiplier(self, t):
        return _warmup(t, self.warmup_method, self.warmup_factor, self.warmup_iters) * \
            _decay(t, self.milestones, self.gamma)

    lr_factor = multiplier   # earlier name, kept for callers

    def get_lr(self):
        m = self.multiplier(self.last_epoch)
        return [lr * m for lr in self.base_lrs]
