import os, sys, torch
sys.path.insert(0, os.getcwd())
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
m = (torch.rand(32, 256, 256, device=dev) > 0.7).float()
for _ in range(5): GF.mask_boxes(m)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ball = torch.randn(4096, 4096, device=dev); ball @ ball
a.record()
for _ in range(50): GF.mask_boxes(m)
b.record(); torch.cuda.synchronize()
print(f"mask_boxes 32 x 256x256: {a.elapsed_time(b) / 50 * 1e3:.1f} us")
