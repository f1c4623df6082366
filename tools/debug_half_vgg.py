"""Per-layer comparison of the VGG16 backbone's backward under ACT_STORAGE = "f16" vs "f32": gradient arriving at every
conv -> BN -> ReLU output (relative L2), and every weight gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphecho_amd import functional as GF, half as GH, nn as gnn
from graphecho_amd.models import fpnseg
from graphecho_amd.models.fpnseg import VGG16

dev = torch.device("cuda:0")
torch.manual_seed(3)
net = VGG16(1).to(dev).train()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(2, 1, size, size, device=dev)
gen = torch.Generator(device="cpu").manual_seed(5)
proj = [None] * 5
rec = {}

def unblk(h):
    B, CB, H, W, _ = h.shape
    return h.float().permute(0, 1, 4, 2, 3).reshape(B, CB * 32, H, W)

orig_h, orig_f = GH.conv_bn, gnn.conv_bn
def wrap(fn, tag):
    def inner(conv, bn, x, relu=False, **kw):
        out = fn(conv, bn, x, relu=relu, **kw)
        key = id(conv)
        def hook(g):
            rec.setdefault(cur[0], {})[key] = (unblk(g) / GH.GRAD_SCALE if GH.is_blocked(g) else g).detach().clone()
        out.register_hook(hook)
        acts.setdefault(cur[0], {})[key] = (unblk(out) if GH.is_blocked(out) else out).detach().clone()
        return out
    return inner
cur = ["f32"]
acts = {}
GH.conv_bn = wrap(orig_h, "h")
gnn.conv_bn = wrap(orig_f, "f")
fpnseg.GH.conv_bn = GH.conv_bn

grads = {}
for mode in ("f32", "f16"):
    cur[0] = mode
    GF.ACT_STORAGE = mode
    for p in net.parameters():
        p.grad = None
    feats = net(x)
    for i, f in enumerate(feats):
        if proj[i] is None:
            proj[i] = (torch.randn(f.shape, generator=gen) / f.numel() ** 0.5).to(dev)
    loss = sum((f * r).sum() for f, r in zip(feats, proj))
    loss.backward()
    grads[mode] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
GF.ACT_STORAGE = "f32"
rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
names = {id(m): n for n, m in net.named_modules()}
for key in rec["f32"]:
    n = names[key]
    a, b = rec["f16"][key], rec["f32"][key]
    m = acts["f32"][key] > 0
    am, bm = a * m, b * m
    print(f"   masked-by-act>0: rel {rel(am, bm):.2e}  norm ratio {a.norm().item() / b.norm().item():.4f}  frac of |g|^2 outside mask {((b * ~m).norm() / b.norm()).item() ** 2:.3f} / {((a * ~m).norm() / a.norm()).item() ** 2:.3f}")
    print(f"{n:12s} act {rel(acts['f16'][key], acts['f32'][key]):.2e}  grad-at-output {rel(a, b):.2e}  |g| {b.norm().item():.3e}  max|g| {b.abs().max().item():.3e}"
          f"  dW {rel(grads['f16'][n + '.weight'], grads['f32'][n + '.weight']):.2e}")

# max-pool windows whose ARGMAX differs between the two runs (the fp16 run's activations differ from the fp32 run's by the
# accumulated rounding, 1e-3 .. 1e-2 relative: enough to swap the two largest entries of a window when they are that close),
# and the share of the pooled gradient's energy that sits in those windows: a swapped window moves its gradient to another
# pixel = 2 g^2 of squared error
import torch.nn.functional as F
for b in range(1, 6):
    last = [m for m in getattr(net, f"block_{b}") if isinstance(m, torch.nn.Conv2d)][-1]
    a16, a32, g = acts["f16"][id(last)], acts["f32"][id(last)], rec["f32"][id(last)]
    unf = lambda a: F.unfold(a.reshape(-1, 1, *a.shape[2:]), 2, stride=2)          # (B*C, 4, windows)
    u16, u32 = unf(a16), unf(a32)
    flip = (u16.argmax(dim=1) != u32.argmax(dim=1)) & (u32.max(dim=1).values > 0)
    tie = ((u16 == u16.max(dim=1, keepdim=True).values).sum(dim=1) > 1) & (u16.max(dim=1).values > 0)
    ge = unf(g).pow(2).sum(dim=1)
    share = ((ge * flip).sum() / ge.sum()).item()
    print(f"block_{b}: activation error {rel(a16, a32):.2e}; windows with another argmax {flip.float().mean().item():.3%} (exact fp16 ties "
          f"{tie.float().mean().item():.3%}); gradient energy in them {share:.3%} -> relative L2 error of this pool's backward alone "
          f"{(2 * share) ** 0.5:.3f}")
