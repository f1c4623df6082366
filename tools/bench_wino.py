"""Winograd F(2x2, 3x3) fp32 kernels (csrc/ge_wino.hip) against the direct implicit-GEMM kernels (ge_mfma.hip) on the large 3x3 /
stride 1 layers of the config-2 step: error of both against an fp64 convolution, time and effective TFLOP/s (direct-conv FLOPs over
time) of forward and data gradient.  python tools/bench_wino.py [frames]"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd._lib import lib, check

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
p = lambda t: None if t is None else t.data_ptr()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for (Cin, Cout, S) in [(256, 256, 64), (256, 128, 64), (128, 128, 64), (64, 64, 64), (128, 128, 32), (256, 256, 32), (256, 256, 16), (512, 512, 8)]:
    H = W = S
    ok_f, ok_d = lib.ge_wino3x3_covered(B, Cin, Cout, H, W), lib.ge_wino3x3_covered(B, Cout, Cin, H, W)
    if not (ok_f and ok_d):
        print(f"{Cin}->{Cout} @{S}x{S}x{B}: not covered (fwd {ok_f}, dgrad {ok_d})")
        continue
    sp_f, sp_d = lib.ge_wino3x3_splits(B, Cin, Cout, H, W), lib.ge_wino3x3_splits(B, Cout, Cin, H, W)
    wsf = torch.empty(max(1, lib.ge_wino3x3_workspace(B, Cin, Cout, H, W)), device=dev)
    wsd = torch.empty(max(1, lib.ge_wino3x3_workspace(B, Cout, Cin, H, W)), device=dev)
    torch.manual_seed(Cin + S)
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5)
    bias = torch.randn(Cout, device=dev)
    dy = torch.randn(B, Cout, H, W, device=dev)
    add = torch.randn(B, Cin, H, W, device=dev)
    flops = 2.0 * B * H * W * Cout * Cin * 9
    u = torch.empty(lib.ge_wino3x3_weight_floats(Cin, Cout), device=dev)
    ut = torch.empty_like(u)
    check(lib.ge_wino3x3_pack_weight(p(w), p(u), Cout, Cin, 0, None), "pack")
    check(lib.ge_wino3x3_pack_weight(p(w), p(ut), Cin, Cout, 1, None), "pack_t")
    y = torch.empty(B, Cout, H, W, device=dev)
    dx = torch.empty_like(x)
    fw = lambda: check(lib.ge_wino3x3_fwd(p(x), p(u), p(bias), None, p(y), None, p(wsf), B, Cin, Cout, H, W, None), "wino fwd")
    dg = lambda: check(lib.ge_wino3x3_fwd(p(dy), p(ut), None, p(add), p(dx), None, p(wsd), B, Cout, Cin, H, W, None), "wino dgrad")
    fw(); dg()
    nb = min(B, 4)
    ref = F.conv2d(x[:nb].double(), w.double(), bias.double(), padding=1)
    refd = torch.nn.grad.conv2d_input(x[:nb].shape, w.double(), dy[:nb].double(), padding=1) + add[:nb].double()
    e_w = ((y[:nb].double() - ref).abs().max() / ref.abs().max()).item()
    e_wd = ((dx[:nb].double() - refd).abs().max() / refd.abs().max()).item()
    # the direct kernels
    wp, wpt = GF._pack_weight(w, 1, False), GF._pack_weight(w, 1, True)
    y2 = torch.empty_like(y)
    dx2 = torch.empty_like(x)
    # the direct route as functional.conv2d takes it: split over K where the tile grid cannot fill the chip
    nf = lib.ge_conv2d_fwd_workspace(B, Cin, Cout, H, W, 3, 3, 1)
    nd = lib.ge_conv2d_dgrad_workspace(B, Cin, H, W, Cout, 3, 3, 1, 1)
    wf2, wd2 = torch.empty(max(1, nf), device=dev), torch.empty(max(1, nd), device=dev)
    if nf:
        fd = lambda: check(lib.ge_conv2d_fwd_splitk(p(x), p(wp), p(bias), p(y2), B, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, p(wf2), None), "fwd")
    else:
        fd = lambda: check(lib.ge_conv2d_fwd(p(x), p(wp), p(bias), p(y2), None, B, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, 0, None), "fwd")
    if nd:
        dd = lambda: check(lib.ge_conv2d_dgrad_splitk(p(dy), p(wpt), p(add), p(dx2), B, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, p(wd2), None), "dgrad")
    else:
        dd = lambda: check(lib.ge_conv2d_dgrad(p(dy), p(wpt), p(add), p(dx2), B, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, None), "dgrad")
    fd(); dd()
    e_d = ((y2[:nb].double() - ref).abs().max() / ref.abs().max()).item()
    e_dd = ((dx2[:nb].double() - refd).abs().max() / refd.abs().max()).item()
    tw, td, twd, tdd = timeit(fw), timeit(fd), timeit(dg), timeit(dd)
    print(f"{Cin}->{Cout} @{S}x{S}x{B} [splits {sp_f}/{sp_d}]: fwd wino {tw * 1e3:.3f} ms ({flops / tw / 1e12:.0f} TF eff, err {e_w:.1e}) direct {td * 1e3:.3f} ms "
          f"({flops / td / 1e12:.0f} TF, err {e_d:.1e}) x{td / tw:.2f} | dgrad wino {twd * 1e3:.3f} ms (err {e_wd:.1e}) direct {tdd * 1e3:.3f} ms "
          f"(err {e_dd:.1e}) x{tdd / twd:.2f}", flush=True)
