"""Discriminator forward + backward (four pyramid levels of config 3 / 5: 64^2, 32^2, 16^2, 8^2; source + target frames) under
ACT_STORAGE = f16: towers in the blocked fp16 domain (GE_H_TOWERS=1) vs the fp32 round trip around every conv (=0), and fp32."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd.models.fpnseg import Discriminator
dev = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
dis = [Discriminator().to(dev).train() for _ in range(4)]
feats = [(torch.randn(frames, 256, s, s, device=dev), torch.randn(frames, 256, s, s, device=dev)) for s in (64, 32, 16, 8)]
def step():
    loss = sum(d((a.requires_grad_(True), b.requires_grad_(True))) for d, (a, b) in zip(dis, feats))
    loss.backward()
for mode, prec in (("f32", "f32"), ("f16", "f16")):
    GF.ACT_STORAGE = mode
    GF.CONV_PRECISION = prec
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"ACT_STORAGE={mode} CONV_PRECISION={prec} GE_H_TOWERS={os.environ.get('GE_H_TOWERS', '1')}: {e0.elapsed_time(e1) / 20:.3f} ms per forward + backward of the four discriminators ({frames} + {frames} frames)")
GF.ACT_STORAGE, GF.CONV_PRECISION = "f32", "f32"
