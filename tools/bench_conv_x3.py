"""bf16x3 (fp32-accurate, six bf16 MFMA products per fp32 product) vs exact-fp32 MFMA conv kernels: time of forward /
data gradient / weight gradient per shape, and the error of both against an fp64 reference on small cases.
usage: bench_conv_x3.py [accuracy|speed|all]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
GF.BX3_HYBRID = False      # every layer and pass on the bf16x3 kernels
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def run(prec, x, w, g, st, k):
    GF.CONV_PRECISION = prec
    xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = GF.conv2d(xx, ww, None, st, k // 2, 1, GF.PackCache())
    dx, dw = torch.autograd.grad(y, (xx, ww), g)
    GF.CONV_PRECISION = "f32"
    return y.detach(), dx, dw


if what in ("accuracy", "all"):
    print("== error against an fp64 reference, relative to sum |a||b| (max over the tensor)")
    torch.manual_seed(0)
    for (B, Cin, H, Cout, k, st) in [(2, 64, 32, 64, 3, 1), (2, 256, 16, 128, 3, 1), (4, 128, 16, 256, 1, 1), (2, 64, 32, 64, 3, 2),
                                     (2, 512, 8, 512, 3, 1), (1, 2048, 8, 256, 1, 1)]:
        x = torch.randn(B, Cin, H, H, device=dev) * torch.exp(torch.randn(B, Cin, 1, 1, device=dev))
        w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
        xd, wd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
        yd = torch.nn.functional.conv2d(xd, wd, None, st, k // 2)
        g = torch.randn(yd.shape, device=dev) * 1e-3
        dxd, dwd = torch.autograd.grad(yd, (xd, wd), g.double().cpu())
        # scale = sum |a||b| of the same contraction
        ya = torch.nn.functional.conv2d(xd.detach().abs(), wd.detach().abs(), None, st, k // 2)
        xa, wa = xd.detach().abs().requires_grad_(True), wd.detach().abs().requires_grad_(True)
        dxa, dwa = torch.autograd.grad(torch.nn.functional.conv2d(xa, wa, None, st, k // 2), (xa, wa), g.double().cpu().abs())
        line = f"B{B} Cin{Cin} {H}x{H} Cout{Cout} k{k}s{st}:"
        for prec in ("f32", "bf16x3", "f16"):
            y, dx, dw = run(prec, x, w, g, st, k)
            e = [((a.double().cpu() - r).abs() / s.clamp_min(1e-300)).max().item()
                 for a, r, s in ((y, yd.detach(), ya), (dx, dxd, dxa), (dw, dwd, dwa))]
            line += f"  {prec}: fwd {e[0]:.1e} dgrad {e[1]:.1e} wgrad {e[2]:.1e} |"
        print(line)

if what in ("speed", "all"):
    print("== time per launch")
    tot = {"f32": 0.0, "bf16x3": 0.0}
    for (B, Cin, H, Cout, k, st) in [(32, 256, 64, 256, 3, 1), (32, 256, 64, 128, 3, 1), (32, 256, 32, 256, 3, 1), (32, 256, 16, 256, 3, 1),
                                     (32, 64, 64, 64, 3, 1), (32, 128, 32, 128, 3, 1), (32, 512, 8, 512, 3, 1), (32, 64, 64, 256, 1, 1),
                                     (32, 256, 64, 64, 1, 1), (32, 256, 64, 256, 1, 1), (32, 1024, 16, 256, 1, 1), (32, 256, 16, 1024, 1, 1),
                                     (32, 512, 32, 128, 1, 1), (32, 128, 64, 128, 3, 2), (8, 256, 64, 256, 3, 1), (8, 256, 16, 256, 3, 1)]:
        x = torch.randn(B, Cin, H, H, device=dev, requires_grad=True)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).requires_grad_(True)
        res = {}
        for prec in ("f32", "bf16x3"):
            GF.CONV_PRECISION = prec
            cache = GF.PackCache()
            t_f = timeit(lambda: GF.conv2d(x.detach(), w.detach(), None, st, k // 2, 1, cache))
            y = GF.conv2d(x, w.detach(), None, st, k // 2, 1, cache)
            g = torch.randn_like(y)
            t_d = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
            y2 = GF.conv2d(x.detach(), w, None, st, k // 2, 1, cache)
            t_w = timeit(lambda: torch.autograd.grad(y2, w, g, retain_graph=True))
            res[prec] = (t_f, t_d, t_w)
            tot[prec] += t_f + t_d + t_w
        GF.CONV_PRECISION = "f32"
        fl = 2.0 * B * (H // st) ** 2 * Cout * Cin * k * k
        a, b = res["f32"], res["bf16x3"]
        print(f"B{B} Cin{Cin} {H}x{H} Cout{Cout} k{k}s{st}: fwd {a[0]*1e3:7.1f} -> {b[0]*1e3:7.1f} us ({fl/b[0]/1e9:6.1f} TF, x{a[0]/b[0]:.2f})"
              f" | dgrad {a[1]*1e3:7.1f} -> {b[1]*1e3:7.1f} us (x{a[1]/b[1]:.2f}) | wgrad {a[2]*1e3:7.1f} -> {b[2]*1e3:7.1f} us ({fl/b[2]/1e9:6.1f} TF, x{a[2]/b[2]:.2f})")
    print(f"sum: f32 {tot['f32']:.2f} ms, bf16x3 {tot['bf16x3']:.2f} ms (x{tot['f32']/tot['bf16x3']:.2f})")
