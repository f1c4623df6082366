"""Race hunt for the Winograd kernels of ge_wino.hip (LDS-DMA with hand-counted vmcnt, three rotating filter buffers): every
shape N times, forward (+ bias) and data gradient (+ addend), beside unrelated traffic on a second stream; each result must equal
the first bit for bit, the first is checked against an fp64 convolution.  usage: stress_wino.py [repeats]"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd._lib import lib, check

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = lambda t: None if t is None else t.data_ptr()
side = torch.cuda.Stream()
noise_a = torch.randn(64 << 20, device=dev)
noise_b = torch.empty_like(noise_a)
bad = 0
for (B, Cin, Cout, H, W) in [(32, 256, 256, 64, 64), (32, 256, 128, 64, 64), (32, 64, 64, 64, 64), (32, 128, 128, 32, 32), (16, 256, 256, 32, 32),
                             (8, 64, 128, 128, 128), (64, 256, 256, 16, 16), (8, 256, 256, 16, 16), (8, 256, 256, 32, 32), (8, 512, 512, 8, 16), (2, 8, 64, 256, 256), (5, 72, 192, 36, 96), (64, 128, 64, 8, 16)]:
    if not (lib.ge_wino3x3_covered(B, Cin, Cout, H, W) and lib.ge_wino3x3_covered(B, Cout, Cin, H, W)):
        print(f"B{B} {Cin}->{Cout} @{H}x{W}: not covered, skipped")
        continue
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    dy = torch.randn(B, Cout, H, W, generator=g).to(dev)
    add = torch.randn(B, Cin, H, W, generator=g).to(dev)
    u = torch.empty(lib.ge_wino3x3_weight_floats(Cin, Cout), device=dev)
    ut = torch.empty_like(u)
    wsf = torch.empty(max(1, lib.ge_wino3x3_workspace(B, Cin, Cout, H, W)), device=dev)
    wsd = torch.empty(max(1, lib.ge_wino3x3_workspace(B, Cout, Cin, H, W)), device=dev)
    check(lib.ge_wino3x3_pack_weight(p(w), p(u), Cout, Cin, 0, None), "pack")
    check(lib.ge_wino3x3_pack_weight(p(w), p(ut), Cin, Cout, 1, None), "pack_t")

    wg = bool(lib.ge_wino3x3_wgrad_covered(B, Cin, Cout, H, W))      # the weight gradient too (ge_wino_wgrad.hip; GE_WNW_WS picks the kernel)
    wsw = torch.empty(max(1, lib.ge_wino3x3_wgrad_workspace(B, Cin, Cout, H, W)) if wg else 1, device=dev)

    def run():
        y = torch.empty(B, Cout, H, W, device=dev)
        dx = torch.empty(B, Cin, H, W, device=dev)
        dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
        check(lib.ge_wino3x3_fwd(p(x), p(u), p(bias), None, p(y), None, p(wsf), B, Cin, Cout, H, W, None), "fwd")
        check(lib.ge_wino3x3_fwd(p(dy), p(ut), None, p(add), p(dx), None, p(wsd), B, Cout, Cin, H, W, None), "dgrad")
        if wg:
            check(lib.ge_wino3x3_wgrad(p(x), p(dy), p(dw), p(wsw), B, Cin, Cout, H, W, 0, None), "wgrad")
        return y, dx, dw

    first = run()
    nb = min(B, 2)
    ref = F.conv2d(x[:nb].double(), w.double(), bias.double(), padding=1)
    refd = torch.nn.grad.conv2d_input(x[:nb].shape, w.double(), dy[:nb].double(), padding=1) + add[:nb].double()
    e_f = ((first[0][:nb].double() - ref).abs().max() / ref.abs().max()).item()
    e_d = ((first[1][:nb].double() - refd).abs().max() / refd.abs().max()).item()
    diff = 0
    for i in range(N):
        with torch.cuda.stream(side):
            noise_b.copy_(noise_a)
        out = run()
        if not (torch.equal(out[0], first[0]) and torch.equal(out[1], first[1]) and torch.equal(out[2], first[2])):
            diff += 1
    torch.cuda.synchronize()
    bad += diff + (e_f > 5e-6) + (e_d > 5e-6)
    print(f"B{B} {Cin}->{Cout} @{H}x{W}: fwd err {e_f:.1e} dgrad err {e_d:.1e} (vs fp64){' + wgrad' if wg else ''}; {diff} of {N} repeats differ", flush=True)
print("FAILED" if bad else "all repeats bit-identical")
sys.exit(1 if bad else 0)
