"""Stage-by-stage comparison of the HIP pyramid-ViG against the CPU oracle (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.models.vig import pvig_ti_224_gelu, Grapher, FFN, Downsample  # noqa: E402
from oracle.vig import deepgcn_forward  # noqa: E402
from oracle.weights import det_tensor, fill_state_dict  # noqa: E402

dev = torch.device("cuda:0")
mod = pvig_ti_224_gelu(num_classes=10)
sd = mod.state_dict()
filled = fill_state_dict(sd, seed=5)
for k in sd:
    if "relative_pos" in k:
        filled[k] = sd[k].clone()
mod.load_state_dict(filled)
x = det_tensor("pvig.x", (2, 3, 224, 224), "uniform")
taps = []
with torch.no_grad():
    y_ref = deepgcn_forward(filled, x, [2, 2, 6, 2], taps=taps)
mod = mod.to(dev).train()
got = []
mod.stem.register_forward_hook(lambda m, i, o: got.append(("stem", o)))
for n, blk in enumerate(mod.backbone):
    if isinstance(blk, Downsample):
        blk.register_forward_hook(lambda m, i, o, n=n: got.append((f"backbone.{n} downsample", o)))
    else:
        blk[0].register_forward_hook(lambda m, i, o, n=n: got.append((f"backbone.{n}.0 grapher", o)))
        blk[1].register_forward_hook(lambda m, i, o, n=n: got.append((f"backbone.{n}.1 ffn", o)))
with torch.no_grad():
    y = mod(x.to(dev))
ref = dict(taps)
for name, o in got:
    if name == "stem":
        o = o + mod.pos_embed
        name = "stem+pos"
    r = ref[name]
    err = (o.cpu() - r).abs().max().item() / r.abs().max().item()
    print(f"{name:28s} shape {tuple(o.shape)}  rel err {err:.3e}")
print("logits rel err", ((y.cpu() - y_ref).abs().max() / y_ref.abs().max()).item())

# Teacher-forced: feed every block the ORACLE's input and count the nodes whose output differs.
print("\nteacher-forced per block (oracle input -> HIP block vs oracle block):")
names = [n for n, _ in taps]
with torch.no_grad():
    for n, blk in enumerate(mod.backbone):
        if isinstance(blk, Downsample):
            continue
        for sub, tag in ((blk[0], f"backbone.{n}.0 grapher"), (blk[1], f"backbone.{n}.1 ffn")):
            i = names.index(tag)
            xin, want = taps[i - 1][1], taps[i][1]
            out = sub(xin.to(dev)).cpu()
            scale = want.abs().max()
            node_err = ((out - want).abs() / scale).amax(dim=1)      # (B, H, W)
            nbad = int((node_err > 1e-4).sum())
            print(f"{tag:28s} nodes {node_err.numel():5d}  differing {nbad:3d}  max {node_err.max():.2e}  "
                  f"median {node_err.median():.2e}")
