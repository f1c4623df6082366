"""Race hunt for the direct-to-LDS 1x1 loader (the kernel counts its own vmcnt): every shape is launched N times, forward
(+ fused moments) and data gradient (+ addend), and each result must equal the first one bit for bit; the first one is
checked against torch on the CPU.  usage: stress_conv1x1.py [repeats]"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd._lib import lib, check
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for (B, Cin, H, Cout) in [(32, 64, 64, 256), (32, 256, 64, 64), (32, 256, 64, 256), (32, 512, 32, 128), (32, 1024, 16, 256), (8, 256, 64, 256),
                          (4, 128, 32, 512), (3, 32, 20, 4), (32, 2048, 8, 512)]:
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn(B, Cin, H, H, generator=g); w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    gy = torch.randn(B, Cout, H, H, generator=g); add = torch.randn(B, Cin, H, H, generator=g)
    xd, wd, gd, ad = x.to(dev), w.to(dev), gy.to(dev), add.to(dev)
    cache = GF.PackCache()
    wp = GF._pack_weight(wd, 1, True)
    y0, s0 = GF.conv2d(xd, wd, None, 1, 0, 1, cache, True)      # (s0 is None on split-K layers: no moments epilogue there)
    k_f = lib.ge_last_conv_kernel().decode()
    dx0 = torch.empty_like(xd)
    args = (gd.data_ptr(), wp.data_ptr(), ad.data_ptr(), dx0.data_ptr(), B, Cin, H, H, Cout, H, H, 1, 1, 1, 0, 1, None)
    check(lib.ge_conv2d_dgrad(*args))
    k_d = lib.ge_last_conv_kernel().decode()
    e_f = (y0.cpu() - F.conv2d(x, w)).abs().max().item()
    e_d = (dx0.cpu() - (torch.nn.grad.conv2d_input(x.shape, w, gy) + add)).abs().max().item()
    diff = 0
    dx = torch.empty_like(xd)
    for i in range(N):
        y, s = GF.conv2d(xd, wd, None, 1, 0, 1, cache, True)
        check(lib.ge_conv2d_dgrad(gd.data_ptr(), wp.data_ptr(), ad.data_ptr(), dx.data_ptr(), B, Cin, H, H, Cout, H, H, 1, 1, 1, 0, 1, None))
        same_s = (s is None and s0 is None) or (s is not None and s0 is not None and torch.equal(s, s0))
        if not (torch.equal(y, y0) and same_s and torch.equal(dx, dx0)):
            diff += 1
    bad += diff + (e_f > 1e-3) + (e_d > 1e-3)
    print(f"B{B} {Cin}->{Cout} @{H}x{H}: fwd err {e_f:.1e} dgrad err {e_d:.1e}; {diff} of {N} repeats differ  [{k_f[-28:]} | {k_d[-28:]}]")
print("FAILED" if bad else "all repeats bit-identical")
sys.exit(1 if bad else 0)
