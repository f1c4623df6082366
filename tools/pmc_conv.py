"""Runs one conv shape (fwd, dgrad, wgrad) a few times -- target for `rocprofv3 --pmc ...` counter collection."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF

B, Cin, H, W, Cout, k, s, p = [int(v) for v in (sys.argv[1:9] if len(sys.argv) > 8 else "32 256 64 64 256 3 1 1".split())]
dev = torch.device("cuda:0")
x = torch.randn(B, Cin, H, W, device=dev, requires_grad=True)
w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).requires_grad_(True)
cache = GF.PackCache()
for _ in range(4):
    y = GF.conv2d(x, w, None, s, p, 1, cache)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
