"""Deterministic, torch-version-independent weight fill (numpy PCG64 keyed by the tensor's name).

TEST INFRASTRUCTURE ONLY.  The reference's own initialisation (N(0, 0.01) linears) makes most losses nearly
input-independent at step 0, so fixtures use O(1/sqrt(fan_in)) weights to keep parity checks
sensitive.  The same function fills the reference modules (tools/gen_golden.py), the oracle and the HIP modules.
"""
import math
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) + 7919 * seed) & 0xFFFFFFFF))


def fill_state_dict(sd, seed=0):
    out = {}
    for name, t in sd.items():
        if not torch.is_floating_point(t):
            out[name] = t.clone()
            continue
        r = _rng(name, seed)
        shape = tuple(t.shape)
        if name.endswith("running_var"):
            v = r.uniform(0.5, 1.5, shape)
        elif name.endswith("running_mean"):
            v = r.normal(0, 0.1, shape)
        elif t.dim() >= 2 and "seed" not in name and "pos_embed" not in name and "queue" not in name:
            fan_in = int(np.prod(shape[1:]))
            v = r.normal(0, math.sqrt(2.0 / fan_in), shape)
        elif name.endswith(".weight"):      # norm scales
            v = 1.0 + 0.1 * r.normal(0, 1, shape)
        elif name.endswith(".bias"):
            v = 0.1 * r.normal(0, 1, shape)
        elif "pos_embed" in name:
            v = 0.1 * r.normal(0, 1, shape)
        else:                                # seed banks, queues, misc buffers
            v = r.normal(0, 1, shape)
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape).clone()
    return out


def det_tensor(name, shape, kind="normal", seed=0):
    """Deterministic input tensor keyed by name: 'normal' N(0,1), 'uniform' U[0,1)."""
    r = _rng("input:" + name, seed)
    v = r.normal(0, 1, shape) if kind == "normal" else r.uniform(0, 1, shape)
    return torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape).clone()


def rect_masks(B, nc, H, W, seed=0, jitter=16):
    """One-hot-ish (B, nc, H, W) masks: channel c is a jittered axis-aligned rectangle (every class present)."""
    base = [(40, 120, 50, 140), (100, 200, 100, 220), (150, 250, 30, 90), (20, 90, 160, 240), (60, 160, 10, 60)]
    r = _rng("masks", seed)
    m = torch.zeros(B, nc, H, W)
    sy, sx = H / 256.0, W / 256.0
    for b in range(B):
        for c in range(nc):
            y0, y1, x0, x1 = base[c % len(base)]
            dy, dx = r.integers(-jitter, jitter + 1, 2)
            ya, yb = int(max(0, (y0 + dy) * sy)), int(min(H, (y1 + dy) * sy))
            xa, xb = int(max(0, (x0 + dx) * sx)), int(min(W, (x1 + dx) * sx))
            m[b, c, ya:yb, xa:xb] = 1.0
    return m
