"""Target process for the rocprofv3 --pmc passes (tools/collect_profile.sh).

Launches, a few times each: two calibration kernels with KNOWN HBM byte counts -- act_fwd (4 B/lane loads and stores,
the access width of the conv loaders / epilogue) and bn_apply (16 B/lane) on a 512 MiB tensor (beyond the 256 MiB
Infinity Cache) -- and the dominant conv shape of the headline workload (3x3, 256->256 @ 64x64, batch 32: forward,
data-gradient, weight-gradient).  MI355X_MICROARCH.md says FETCH_SIZE/WRITE_SIZE are uncalibrated on gfx950 except
for 16 B/lane streaming reads (reported at 1/2): the calibration kernels give the correction factor per access width.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF

dev = torch.device("cuda:0")
cal = torch.randn(32, 256, 128, 128, device=dev)          # 512 MiB
ones, zeros = torch.ones(256, device=dev), torch.zeros(256, device=dev)
for _ in range(3):
    GF.relu(cal)                                          # act_fwd_kernel: reads 512 MiB, writes 512 MiB, 4 B/lane
    GF.batch_norm(cal, ones, zeros, zeros.clone(), ones.clone(), False, 0.1, 1e-5)   # bn_apply_kernel: 16 B/lane
del cal
B, Cin, H, W, Cout, k, s, p = 32, 256, 64, 64, 256, 3, 1, 1
x = torch.randn(B, Cin, H, W, device=dev, requires_grad=True)
w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).requires_grad_(True)
cache = GF.PackCache()
for _ in range(4):
    y = GF.conv2d(x, w, None, s, p, 1, cache)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
