"""nn.Module layers backed by the HIP kernels.

Each layer subclasses the torch module it replaces so that constructor arguments, default initialisation,
parameter/buffer names and therefore ``state_dict`` keys are exactly those of the reference's layers; only
``forward`` is re-routed to graphecho_amd.functional (no ATen compute on the hot path).
"""
import torch
import torch.nn as tnn

from . import functional as GF


class Conv2d(tnn.Conv2d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.kernel_size[0] != self.kernel_size[1] or self.stride[0] != self.stride[1] or \
                self.padding[0] != self.padding[1] or self.dilation != (1, 1) or self.padding_mode != "zeros":
            raise NotImplementedError("graphecho_amd.nn.Conv2d: square kernels / symmetric stride+padding only")
        self._pack = GF.PackCache()

    def forward(self, x, bn_stats=False):
        """bn_stats=True -> (y, stats): BatchNorm moments of y fused into the conv epilogue."""
        return GF.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.groups, self._pack, bn_stats)

    def forward_with_skip(self, x, bn_stats=False):
        """(conv(x), skip[, stats]): use `skip` for every other consumer of x (see functional.conv2d_with_skip)."""
        if not (x.requires_grad and torch.is_grad_enabled()):
            out = self.forward(x, bn_stats)
            return (out[0], x, out[1]) if bn_stats else (out, x)
        return GF.conv2d_with_skip(x, self.weight, self.bias, self.stride[0], self.padding[0], self.groups, self._pack,
                                   bn_stats)


class Linear(tnn.Linear):
    def forward(self, x):
        return GF.linear(x, self.weight, self.bias)


class BatchNorm2d(tnn.BatchNorm2d):
    """Train-mode batch statistics (SyncBN when ``process_group`` is set and torch.distributed is initialised)."""

    process_group = None
    sync = False
    force_sync = False   # take the SyncBN code path even at world size 1 (single-GPU self-test)
    _pending_batches = 0  # num_batches_tracked increments not yet written to the device buffer

    def _flush_batches(self):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(self._pending_batches)
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_batches()   # the counter is kept on the host between checkpoints (one launch per save, not per step)
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, residual=None, relu=False, partial=None):
        training = self.training or not self.track_running_stats
        group = None
        if training and self.sync and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and (torch.distributed.get_world_size() > 1 or self.force_sync):
            group = self.process_group if self.process_group is not None else torch.distributed.group.WORLD
        segments = GF.BN_SEGMENTS if training else None
        if training and self.track_running_stats and self.num_batches_tracked is not None:
            self._pending_batches += len(segments) if segments else 1
        mom = 0.1 if self.momentum is None else self.momentum
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        return GF.batch_norm(x, self.weight, self.bias, rm, rv, training, mom, self.eps, residual, relu, group,
                             partial if training else None, segments)


def conv_bn(conv, bn, x, relu=False, residual=None, with_skip=False):
    """bn(conv(x)) (+residual)(+ReLU; relu="gelu": + erf-GELU) with the batch statistics computed in the conv epilogue
    (train mode), so the activation is written once and read once.  with_skip=True additionally returns the skip alias
    of x."""
    want = bn.training or not bn.track_running_stats
    if with_skip:
        out = conv.forward_with_skip(x, bn_stats=want)
        y, skip, part = (out[0], out[1], out[2]) if want else (out[0], out[1], None)
        return bn(y, residual=residual, relu=relu, partial=part), skip
    if want:
        y, part = conv(x, bn_stats=True)
        return bn(y, residual=residual, relu=relu, partial=part)
    return bn(conv(x), residual=residual, relu=relu)


def convert_sync_batchnorm(module, process_group=None):
    """Counterpart of torch.nn.SyncBatchNorm.convert_sync_batchnorm (train_camus_echo.py:130)."""
    for m in module.modules():
        if isinstance(m, BatchNorm2d):
            m.sync = True
            m.process_group = process_group
    return module


class GroupNorm(tnn.GroupNorm):
    def forward(self, x, relu=False):
        return GF.group_norm(x, self.num_groups, self.weight, self.bias, self.eps, relu)


class LayerNorm(tnn.LayerNorm):
    def forward(self, x):
        if len(self.normalized_shape) != 1:
            raise NotImplementedError("graphecho_amd.nn.LayerNorm: last-dim normalisation only")
        return GF.layer_norm(x, self.weight, self.bias, self.eps)


class ReLU(tnn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return GF.relu(x)


class GELU(tnn.Module):
    def forward(self, x):
        return GF.gelu(x)


class LeakyReLU(tnn.Module):
    def __init__(self, negative_slope=0.01, inplace=False):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, x):
        return GF.leaky_relu(x, self.negative_slope)


class PReLU(tnn.PReLU):
    def forward(self, x):
        return GF.prelu(x, self.weight)


class Hardswish(tnn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return GF.hardswish(x)


class MaxPool2d(tnn.MaxPool2d):
    def forward(self, x):
        return GF.max_pool2d(x, self.kernel_size, self.stride, self.padding)


class AdaptiveAvgPool2d1(tnn.Module):
    """nn.AdaptiveAvgPool2d(1)."""

    def forward(self, x):
        return GF.adaptive_avg_pool2d_1(x)


class InstanceNormMatrix(tnn.Module):
    """nn.InstanceNorm2d(1) applied to a (1, 1, N1, N2) matrix: whole-matrix standardisation, eps 1e-5."""

    def __init__(self, eps=1e-5):
        super().__init__()
        self.eps = eps

    def forward(self, m):
        flat = m.reshape(1, -1)
        return GF.layer_norm(flat, None, None, self.eps).reshape(m.shape)
