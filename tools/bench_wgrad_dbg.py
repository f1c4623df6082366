import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd._lib import lib, check
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
SHAPES = [(32, 256, 64, 256, 3), (32, 256, 64, 128, 3), (32, 256, 64, 256, 1), (32, 64, 64, 256, 1), (32, 1024, 16, 256, 1), (32, 256, 16, 256, 3)]
if len(sys.argv) > 1 and sys.argv[1] == "1x1":
    SHAPES = [(32, 256, 64, 256, 1), (32, 512, 32, 256, 1), (32, 512, 64, 256, 1), (32, 256, 16, 1024, 1), (32, 1024, 16, 256, 1), (32, 128, 32, 512, 1), (32, 64, 64, 256, 1), (32, 256, 64, 64, 1), (32, 512, 32, 128, 1), (32, 512, 8, 2048, 1)]
for (B, Cin, H, Cout, k) in SHAPES:
    x = torch.randn(B, Cin, H, H, device=dev); dy = torch.randn(B, Cout, H, H, device=dev)
    dw = torch.empty(Cout, Cin, k, k, device=dev)
    ws = torch.empty(lib.ge_conv2d_wgrad_workspace(B, Cin, Cout, H, H, k, k, 1), device=dev)
    t = timeit(lambda: check(lib.ge_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, Cin, H, H, Cout, H, H, k, k, 1, k // 2, 1, 0, None)))
    fl = 2.0 * B * H * H * Cout * Cin * k * k
    print(f"dbg={os.environ.get('GE_CONV_DEBUG','0')} wgrad B{B} Cin{Cin} {H}x{H} Cout{Cout} k{k}: {t*1e3:8.1f} us ({fl/t/1e9:6.1f} TF)")
