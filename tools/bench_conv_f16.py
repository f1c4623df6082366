"""fp16-input MFMA conv kernels vs the fp32 MFMA kernels (forward; same shapes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (B, Cin, H, Cout, k, st) in [(32, 256, 64, 256, 3, 1), (32, 256, 64, 128, 3, 1), (32, 256, 32, 256, 3, 1), (32, 256, 16, 256, 3, 1),
                                 (32, 64, 64, 64, 3, 1), (32, 512, 8, 512, 3, 1), (32, 64, 64, 256, 1, 1), (32, 256, 64, 256, 1, 1),
                                 (32, 1024, 16, 256, 1, 1), (32, 256, 16, 1024, 1, 1), (32, 128, 64, 128, 3, 2)]:
    x = torch.randn(B, Cin, H, H, device=dev, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).requires_grad_(True)
    res = {}
    for prec in ("f32", "f16"):
        GF.CONV_PRECISION = prec
        cache = GF.PackCache()
        t_f = timeit(lambda: GF.conv2d(x.detach(), w.detach(), None, st, k // 2, 1, cache))
        y = GF.conv2d(x, w.detach(), None, st, k // 2, 1, cache)
        g = torch.randn_like(y)
        t_d = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
        res[prec] = (t_f, t_d)
    GF.CONV_PRECISION = "f32"
    fl = 2.0 * B * (H // st) ** 2 * Cout * Cin * k * k
    print(f"B{B} Cin{Cin} {H}x{H} Cout{Cout} k{k}s{st}: fwd f32 {res['f32'][0]*1e3:7.1f} us ({fl/res['f32'][0]/1e9:6.1f} TF) | f16 {res['f16'][0]*1e3:7.1f} us ({fl/res['f16'][0]/1e9:6.1f} TF)"
          f" || dgrad f32 {res['f32'][1]*1e3:7.1f} us | f16 {res['f16'][1]*1e3:7.1f} us ({fl/res['f16'][1]/1e9:6.1f} TF)")
