"""CPU restatement of the input-formatting step and the overlap metric (test infrastructure only).

Follows the reference: MONAI ``Resized(mode='nearest')`` = ``torch.nn.functional.interpolate(mode='nearest')``
(src = min(floor(dst * in/out), in - 1), scale in float32), ``Rand/CenterSpatialCropd``, ``/ 255.0``, ``np.where`` one-hot
(datasets/cardiac_uda.py:128-155,248-286; datasets/camus.py:98-105,121-159), clip fold train_camus_echo.py:247-251,
and ``_calculate_overlap_metrics`` train_camus_echo.py:402-417.  Pinned in tests/test_oracle_golden.py against
``torch.nn.functional.interpolate`` itself (the arithmetic MONAI calls; MONAI is not installed here).
"""
import numpy as np


def nearest_index(out_size, in_size):
    scale = np.float32(in_size) / np.float32(out_size)
    idx = np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(idx, in_size - 1)


def prepare_frames(src, spatial_size, crop_size, offsets=None, center=False, clip_length=None, divisor=255.0):
    src = np.asarray(src)
    clips = src.ndim == 5
    if not clips:
        src = src[..., None]
    N, C, H, W, T = src.shape
    S, crop = int(spatial_size), int(crop_size)
    To = int(clip_length) if (clips and clip_length is not None) else T
    iy, ix, it = nearest_index(S, H), nearest_index(S, W), nearest_index(To, T)
    out = np.empty((N, To, C, crop, crop), dtype=np.float32)
    for n in range(N):
        if offsets is not None:
            oy, ox = int(offsets[n][0]), int(offsets[n][1])
        elif center:
            oy = ox = S // 2 - crop // 2
        else:
            oy = ox = 0
        ys, xs = iy[oy:oy + crop], ix[ox:ox + crop]
        block = src[n][:, ys][:, :, xs][:, :, :, it]            # C, crop, crop, To
        out[n] = np.transpose(block, (3, 0, 1, 2)).astype(np.float32) / np.float32(divisor)
    return out.reshape(N * To, C, crop, crop)


def onehot_labels(labels, class_values, spatial_size, crop_size, offsets=None, center=False, clip_length=None):
    labels = np.asarray(labels)
    clips = labels.ndim == 4
    if not clips:
        labels = labels[..., None]
    stacked = np.stack([np.where(labels == v, 1, 0) for v in class_values], axis=1).astype(np.float32)   # N, NC, H, W, T
    out = prepare_frames(stacked, spatial_size, crop_size, offsets, center, clip_length if clips else None, 1.0)
    return out


def overlap_metrics(logits, masks, eps=1e-5):
    """Per class: (pixel_acc, dice, precision, specificity, recall) of (sigmoid(logit) > 0.5) vs mask."""
    out = []
    for c in range(logits.shape[1]):
        o = (1.0 / (1.0 + np.exp(-logits[:, c].astype(np.float64))) > 0.5).astype(np.float64).reshape(-1)
        t = masks[:, c].astype(np.float64).reshape(-1)
        tp, fp = np.sum(o * t), np.sum(o * (1 - t))
        fn, tn = np.sum((1 - o) * t), np.sum((1 - o) * (1 - t))
        out.append(((tp + tn + eps) / (tp + tn + fp + fn + eps), (2 * tp + eps) / (2 * tp + fp + fn + eps),
                    (tp + eps) / (tp + fp + eps), (tn + eps) / (tn + fp + eps), (tp + eps) / (tp + fn + eps)))
    return np.array(out)
