"""Target process for the rocprofv3 --pmc passes over the blocked-fp16 conv kernels (tools/collect_half.sh): a calibration
kernel with a KNOWN byte count in the same access width (bnh_apply: 16 B/lane reads and writes over a 512 MiB fp16 tensor,
beyond the 256 MiB Infinity Cache) and config 5's dominant layer shape (256 -> 256 @ 64 x 64, 48 frames): forward, data
gradient, weight gradient, a few times each.  Algorithmic bytes per launch: 2 * (in + w + out) for the first two,
2 * (x + dz) + 4 * slabs written for the third."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd._lib import lib, check

dev = torch.device("cuda:0")
p = lambda t: t.data_ptr()
cal = torch.randn(64, 4, 128, 128, 32, device=dev).half()          # 128 Mi elements = 256 MiB read + 256 MiB written
out = torch.empty_like(cal)
C = 128
mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
for _ in range(3):
    check(lib.ge_h_bn_apply(p(cal), p(mean), p(invstd), None, None, p(out), 64, C, 128 * 128, 1, None), "cal")
del cal, out
B, Cin, Cout, H, W = 48, 256, 256, 64, 64
h = (torch.randn(B, Cin // 32, H, W, 32, device=dev)).half()
dz = (torch.randn(B, Cout // 32, H, W, 32, device=dev) * 0.1).half()
z, dh = torch.empty_like(dz), torch.empty_like(h)
w = torch.randn(Cout, Cin, 3, 3, device=dev) / 48.0
wp, wpt = GF._pack_weight_lp(w, 1, False, "f16"), GF._pack_weight_lp(w, 1, True, "f16")
stats = torch.empty(Cout, lib.ge_h_conv3x3_stat_parts(B, H, W), 3, device=dev)
ws = torch.empty(lib.ge_h_conv3x3_wgrad_workspace(B, Cin, Cout, H, W), device=dev)
dw = torch.empty_like(w)
for _ in range(4):
    check(lib.ge_h_conv3x3_fwd(p(h), p(wp), None, p(z), p(stats), B, Cin, Cout, H, W, None), "f")
    check(lib.ge_h_conv3x3_dgrad(p(dz), p(wpt), p(dh), B, Cin, Cout, H, W, None), "d")
    check(lib.ge_h_conv3x3_wgrad(p(h), p(dz), p(dw), p(ws), B, Cin, Cout, H, W, 1.0, None, 0, None), "w")
torch.cuda.synchronize()
n_in, n_w = 2 * h.numel(), 2 * w.numel()
print({"cal_bytes_read": 2 * 64 * C * 128 * 128, "cal_bytes_written": 2 * 64 * C * 128 * 128,
       "fwd_algorithmic_read": n_in + n_w, "fwd_algorithmic_written": 2 * z.numel() + 4 * stats.numel(),
       "wgrad_algorithmic_read": 2 * (h.numel() + dz.numel()), "wgrad_slab_bytes": 4 * ws.numel(),
       "gflop": 2.0 * B * H * W * Cout * Cin * 9 / 1e9})
