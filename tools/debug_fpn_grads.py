import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.models.fpnseg import FPN
from graphecho_amd import functional as GF
from oracle.fpn import fpn_forward
from oracle.misc import seg_loss_cardiac
from oracle.weights import fill_state_dict

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()

bb, cin, nc, hw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(0)
net = FPN([2, 4, 23, 3], nc, cin, back_bone=bb)
sd = fill_state_dict(net.state_dict(), seed=1)
net.load_state_dict(sd)
gen = torch.Generator().manual_seed(3)
x = torch.rand(2, cin, hw, hw, generator=gen)
t = (torch.rand(2, nc, hw, hw, generator=gen) > 0.6).float()
params = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
ref_logits, ref_pyr = fpn_forward(params, x, True)
seg_loss_cardiac(ref_logits, t).backward()
p64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
p64 = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in p64.items()}
l64, _ = fpn_forward(p64, x.double(), True)
seg_loss_cardiac(l64, t.double()).backward()
print("cpu32 vs 64 logits", rel(ref_logits, l64))
dev = torch.device("cuda:0")
net = net.to(dev).train()
logits, pyr = net(x.to(dev))
(GF.dice_loss(logits, t.to(dev)) + GF.bce_with_logits(logits, t.to(dev))).backward()
print("logits", rel(logits, ref_logits))
print("hip vs 64 logits", rel(logits, l64))
errs = sorted(((rel(p.grad, p64[n].grad), rel(params[n].grad, p64[n].grad), n, p64[n].grad.abs().max().item()) for n, p in net.named_parameters()), reverse=True)
for e in errs[:14]:
    print("hip %.3e cpu32 %.3e %-45s refmax %.3e" % e)
