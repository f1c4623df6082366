"""GPU-side input formatting in front of ``FPN.forward``.

The reference formats every sample on DataLoader workers with MONAI (``AddChanneld`` -> ``Resized(mode='nearest')`` ->
``RandSpatialCropd`` / ``CenterSpatialCropd``), divides by 255, builds the one-hot masks with ``np.where`` + ``np.stack``
(datasets/cardiac_uda.py:128-155,248-286; datasets/camus.py:98-105,121-159) and folds clips ``(b,c,h,w,t)`` into the
batch on the device (train_camus_echo.py:247-251).  Here the raw uint8 frames / label maps go to HBM once and one
kernel per tensor does resize + crop + scale + fold (`ge_frames_prepare`, `ge_labels_onehot`).
"""
import torch

from ._lib import check, lib

_f32 = torch.float32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream as an int (the raw getter is ~10x cheaper than building a Stream object;
    a training step asks for it ~2000 times)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _geometry(src_hw, spatial_size, crop_size, clip_in, clip_length):
    S = int(spatial_size)
    crop = int(crop_size)
    if crop > S:
        raise ValueError(f"crop_size {crop} exceeds spatial_size {S}")
    To = int(clip_length) if clip_length is not None else int(clip_in)
    return S, crop, To


def _origins(N, S, crop, offsets, center, device):
    """(device int32 [N,2] or None, oy, ox): explicit per-sample origins, the centre crop, or the origin (0, 0)."""
    if offsets is not None:
        off = torch.as_tensor(offsets, dtype=torch.int32, device=device).reshape(N, 2).contiguous()
        if int(off.min()) < 0 or int(off.max()) + crop > S:
            raise ValueError("crop origin outside the resized frame")
        return off, 0, 0
    if center:   # MONAI CenterSpatialCrop: start = S // 2 - crop // 2
        o = S // 2 - crop // 2
        return None, o, o
    return None, 0, 0


def random_crop_origins(N, spatial_size, crop_size, generator=None):
    """Per-sample (y, x) origins like RandSpatialCropd(random_size=False): uniform in [0, S - crop]."""
    hi = int(spatial_size) - int(crop_size) + 1
    return torch.randint(0, hi, (N, 2), generator=generator, dtype=torch.int32)


def prepare_frames(src, spatial_size, crop_size, offsets=None, center=False, clip_length=None, divisor=255.0):
    """src: (N, C, H, W) or clips (N, C, H, W, T), uint8 or float32, on the HIP device.
    -> float32 (N*T', C, crop, crop): nearest-resized to spatial_size (and T' = clip_length frames), cropped, divided by `divisor`;
    clips are folded time-major per sample exactly like ``permute(0,4,1,2,3).reshape(-1,c,h,w)``."""
    if src.dtype not in (torch.uint8, torch.float32):
        raise TypeError("prepare_frames: uint8 or float32 frames expected")
    if not src.is_cuda:
        raise RuntimeError("prepare_frames: frames must be on the HIP device (no CPU path)")
    src = src.contiguous()
    clips = src.dim() == 5
    N, C, H, W = src.shape[:4]
    T = src.shape[4] if clips else 1
    S, crop, To = _geometry((H, W), spatial_size, crop_size, T, clip_length if clips else None)
    off, oy, ox = _origins(N, S, crop, offsets, center, src.device)
    dst = torch.empty((N * To, C, crop, crop), device=src.device, dtype=_f32)
    check(lib.ge_frames_prepare(_p(src), int(src.dtype == torch.float32), _p(dst), _p(off), N, C, H, W, T, S, To, crop,
                                oy, ox, float(divisor), _stream()), "frames_prepare")
    return dst


def onehot_labels(labels, class_values, spatial_size, crop_size, offsets=None, center=False, clip_length=None):
    """labels: (N, H, W) or (N, H, W, T) uint8 class ids on the HIP device; class_values e.g. (0, 1, 2) for
    background + LV + RV (cardiac_uda.py:130-133) or (1, 3) for CAMUS LV/LA (camus.py:99-101).
    -> float32 (N*T', len(class_values), crop, crop) one-hot planes with the frames' geometry."""
    if labels.dtype != torch.uint8:
        raise TypeError("onehot_labels: uint8 label maps expected")
    if not labels.is_cuda:
        raise RuntimeError("onehot_labels: label maps must be on the HIP device (no CPU path)")
    labels = labels.contiguous()
    clips = labels.dim() == 4
    N, H, W = labels.shape[:3]
    T = labels.shape[3] if clips else 1
    S, crop, To = _geometry((H, W), spatial_size, crop_size, T, clip_length if clips else None)
    off, oy, ox = _origins(N, S, crop, offsets, center, labels.device)
    vals = torch.as_tensor(list(class_values), dtype=torch.int32, device=labels.device)
    dst = torch.empty((N * To, vals.numel(), crop, crop), device=labels.device, dtype=_f32)
    check(lib.ge_labels_onehot(_p(labels), _p(dst), _p(off), _p(vals), N, vals.numel(), H, W, T, S, To, crop, oy, ox,
                               _stream()), "labels_onehot")
    return dst


class OverlapMeter:
    """Running TP/FP/FN/TN per class of (sigmoid(logit) > 0.5) against binary masks, kept on the device
    (the reference thresholds, concatenates and reduces whole validation sets on the host:
    train_camus_echo.py:350-417).  ``metrics()`` applies the reference's formulas with eps = 1e-5."""

    def __init__(self, num_classes, device):
        self.counts = torch.zeros((num_classes, 4), dtype=torch.int64, device=device)

    def update(self, logits, masks):
        logits, masks = logits.contiguous(), masks.to(_f32).contiguous()
        B, C, H, W = logits.shape
        if masks.shape != logits.shape or C != self.counts.shape[0]:
            raise ValueError("OverlapMeter.update: logits / masks shape mismatch")
        check(lib.ge_overlap_counts(_p(logits), _p(masks), _p(self.counts), B, C, H * W, _stream()), "overlap_counts")

    def metrics(self, eps=1e-5):
        """dict of per-class tensors: pixel_acc, dice, precision, specificity, recall (one host read)."""
        c = self.counts.to(torch.float64).cpu()
        tp, fp, fn, tn = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
        return {"pixel_acc": (tp + tn + eps) / (tp + tn + fp + fn + eps), "dice": (2 * tp + eps) / (2 * tp + fp + fn + eps),
                "precision": (tp + eps) / (tp + fp + eps), "specificity": (tn + eps) / (tn + fp + eps),
                "recall": (tp + eps) / (tp + fn + eps)}
