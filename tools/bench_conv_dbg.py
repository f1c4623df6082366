import os, sys, torch
sys.path.insert(0, "/root/repo")
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (B, Cin, H, Cout, k) in [(32, 64, 64, 256, 1), (32, 256, 64, 64, 1), (32, 256, 64, 256, 1), (32, 512, 32, 128, 1), (32, 128, 32, 512, 1), (32, 1024, 16, 256, 1), (32, 256, 16, 1024, 1), (32, 256, 64, 256, 3)]:
    x = torch.randn(B, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    cache = GF.PackCache()
    t0 = timeit(lambda: GF.conv2d(x, w, None, 1, k // 2, 1, cache))
    fl = 2.0 * B * H * H * Cout * Cin * k * k
    print(f"dbg={os.environ.get('GE_CONV_DEBUG','0')} B{B} Cin{Cin} {H}x{H} Cout{Cout} k{k}: {t0*1e3:8.1f} us ({fl/t0/1e9:6.1f} TF)  mfma-floor {fl/157.3e12*1e6:6.1f} us")
