"""Forward conv with / without the fused BatchNorm-statistics epilogue (cost of the epilogue per shape)."""
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
for (B, Cin, H, Cout, k) in [(32, 64, 64, 256, 1), (32, 256, 64, 64, 1), (32, 64, 64, 64, 3), (32, 256, 64, 256, 1), (32, 512, 32, 128, 1),
                             (32, 128, 32, 512, 1), (32, 256, 64, 256, 3), (32, 1024, 16, 256, 1), (32, 256, 16, 256, 3)]:
    x = torch.randn(B, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    cache = GF.PackCache()
    t0 = timeit(lambda: GF.conv2d(x, w, None, 1, k // 2, 1, cache))
    t1 = timeit(lambda: GF.conv2d(x, w, None, 1, k // 2, 1, cache, True))
    fl = 2.0 * B * H * H * Cout * Cin * k * k
    print(f"B{B} Cin{Cin} {H}x{H} Cout{Cout} k{k}: plain {t0:.4f} ms ({fl/t0/1e9:.1f} TF)  +stats {t1:.4f} ms ({fl/t1/1e9:.1f} TF)  epilogue cost {100*(t1-t0)/t0:.1f}%")
