"""Range check of the fp16-stored gradients (config 5, conv_precision="f16s"): at every fp32 -> fp16 gradient cast of a step
(the step's loss scale: functional.h_scale_value) the largest scaled magnitude against fp16's maximum 65504 and the share of non-zero
elements that land below fp16's smallest NORMAL number 2^-14 (they keep fewer than 11 significant bits; below 2^-24 they vanish)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd._lib import lib
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

dev = torch.device("cuda:0")
tr = GraphEchoTrainer(dev, workload="temporal", back_bone="VGG16", in_channel=1, num_classes=4, image_size=256, seed=0,
                      conv_precision="f16s", clip_len=16, transport_method="sinkhorn_distance", seg_loss="cardiac")
xs, ms = synthetic_batch(8, 1, 4, 256, dev, 1234)
xt, _ = synthetic_batch(8, 1, 4, 256, dev, 4321)
def clip(seed, t=16):
    f, mk = synthetic_batch(t, 1, 4, 256, dev, seed)
    return (f.reshape(1, t, 1, 256, 256).permute(0, 2, 3, 4, 1).contiguous(), mk.reshape(1, t, 4, 256, 256).permute(0, 2, 3, 4, 1).contiguous())
cs, cm = clip(77); ct, _ = clip(78)
clips = {"source": cs, "target": ct, "masks": cm}
for _ in range(3):
    tr.step(xs, ms, xt, clips)
rows = []
# the two Python entry points that cast an fp32 gradient to fp16 with the loss scale
import graphecho_amd.half as GH
orig_fb = GH._FromBlockedFn.backward
def fb(ctx, dx):
    a = (dx.detach().abs() * SC[0])
    nz = a[a > 0]
    rows.append(("stack exit", tuple(dx.shape), a.max().item(), (nz < 2.0 ** -14).float().mean().item() if nz.numel() else 0.0,
                 (nz < 2.0 ** -24).float().mean().item() if nz.numel() else 0.0))
    return orig_fb(ctx, dx)
GH._FromBlockedFn.backward = staticmethod(fb)
orig_cb = GF._Conv2dFn._backward_h
def cb(ctx, xh, weight, dy, dskip):
    a = (dy.detach().abs() * SC[0])
    nz = a[a > 0]
    rows.append(("3x3 conv", tuple(dy.shape), a.max().item(), (nz < 2.0 ** -14).float().mean().item() if nz.numel() else 0.0,
                 (nz < 2.0 ** -24).float().mean().item() if nz.numel() else 0.0))
    return orig_cb(ctx, xh, weight, dy, dskip)
GF._Conv2dFn._backward_h = staticmethod(cb)
SC = [1.0]
for rep in range(2):      # the scale of a step comes from the step before it
    rows.clear()
    GF.h_scale_update()
    SC[0] = GF.h_scale_value(dev)
    tr.step(xs, ms, xt, clips)
torch.cuda.synchronize()
print(f"loss scale {SC[0]:g} ({'dynamic' if GF.H_DYNAMIC_SCALE else 'fixed'}); {len(rows)} gradient casts in the step")
print(f"largest scaled |g| over the step: {max(r[2] for r in rows):.3g} (fp16 max 65504)")
worst = sorted(rows, key=lambda r: -r[3])[:6]
for r in worst:
    print(f"  {r[0]:10s} {str(r[1]):24s} max {r[2]:.3g}  below 2^-14: {r[3]:.2%}  below 2^-24 (lost): {r[4]:.3%}")
tot = sum(1 for r in rows if r[3] > 0.01)
print(f"casts with more than 1 % of their non-zero elements below 2^-14: {tot} of {len(rows)}")
