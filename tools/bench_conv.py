"""Micro-benchmark of the implicit-GEMM conv kernels on the dominant FPN shapes (TFLOP/s vs the 157.3 fp32-MFMA peak)."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF

SHAPES = [  # B, Cin, H, W, Cout, k, s, p, groups
    (32, 256, 64, 64, 256, 3, 1, 1, 1),
    (32, 256, 64, 64, 128, 3, 1, 1, 1),
    (32, 256, 32, 32, 256, 3, 1, 1, 1),
    (32, 256, 16, 16, 256, 3, 1, 1, 1),
    (32, 64, 64, 64, 64, 3, 1, 1, 1),
    (32, 128, 32, 32, 128, 3, 1, 1, 1),
    (32, 512, 8, 8, 512, 3, 1, 1, 1),
    (32, 64, 64, 64, 256, 1, 1, 0, 1),
    (32, 256, 64, 64, 64, 1, 1, 0, 1),
    (32, 1024, 16, 16, 256, 1, 1, 0, 1),
    (32, 256, 16, 16, 1024, 1, 1, 0, 1),
    (32, 2048, 8, 8, 512, 1, 1, 0, 1),
    (32, 512, 8, 8, 2048, 1, 1, 0, 1),
    (32, 128, 32, 32, 512, 1, 1, 0, 1),
    (32, 512, 32, 32, 128, 1, 1, 0, 1),
    (32, 3, 256, 256, 64, 7, 2, 3, 1),
    (32, 128, 64, 64, 128, 3, 2, 1, 1),
    (32, 512, 64, 64, 512, 1, 1, 0, 4),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = torch.device("cuda:0")
    print(f"{'shape':44s} {'fwd ms':>8s} {'TF':>6s} {'dgrad':>8s} {'TF':>6s} {'wgrad':>8s} {'TF':>6s}")
    for (B, Cin, H, W, Cout, k, s, p, g) in SHAPES:
        x = torch.randn(B, Cin, H, W, device=dev)
        w = torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05
        x.requires_grad_(True)
        w.requires_grad_(True)
        y = GF.conv2d(x, w, None, s, p, g)
        gy = torch.randn_like(y)
        flops = 2.0 * B * y.shape[2] * y.shape[3] * Cout * (Cin // g) * k * k
        cache = GF.PackCache()
        t_f = timeit(lambda: GF.conv2d(x.detach(), w.detach(), None, s, p, g, cache))
        xd, wd = x.detach(), w.detach()
        xg = xd.clone().requires_grad_(True)
        t_d = timeit(lambda: torch.autograd.grad(GF.conv2d(xg, wd, None, s, p, g, cache), xg, gy)) - t_f
        wg = wd.clone().requires_grad_(True)
        t_w = timeit(lambda: torch.autograd.grad(GF.conv2d(xd, wg, None, s, p, g, cache), wg, gy)) - t_f
        tf = lambda t: flops / (t * 1e-3) / 1e12
        print(f"{str((B, Cin, H, W, Cout, k, s, p, g)):44s} {t_f:8.3f} {tf(t_f):6.1f} {t_d:8.3f} {tf(t_d):6.1f} "
              f"{t_w:8.3f} {tf(t_w):6.1f}", flush=True)


if __name__ == "__main__":
    main()
