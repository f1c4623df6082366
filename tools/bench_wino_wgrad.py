"""Winograd F(3x3, 2x2) weight gradient (csrc/ge_wino_wgrad.hip) against the direct kernel (ge_mfma.hip) on the 3x3 / stride 1 layers
of the training step: time of both (kernel + slab reduce), effective TFLOP/s (direct-conv FLOPs over time), error of both against
an fp64 correlation on a slice of the batch.  python tools/bench_wino_wgrad.py [frames]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


for (Cin, Cout, S) in [(256, 256, 64), (256, 128, 64), (128, 128, 64), (64, 64, 64), (128, 128, 32), (256, 256, 32), (256, 256, 16), (512, 512, 16)]:
    H = W = S
    if not lib.ge_wino3x3_wgrad_covered(B, Cin, Cout, H, W):
        print(f"{Cin}->{Cout} @{S}x{S}x{B}: not covered")
        continue
    routed = bool(lib.ge_wino3x3_wgrad_supported(B, Cin, Cout, H, W))
    torch.manual_seed(Cin + S)
    x = torch.randn(B, Cin, H, W, device=dev)
    dy = torch.randn(B, Cout, H, W, device=dev)
    flops = 2.0 * B * H * W * Cout * Cin * 9
    ws = torch.empty(lib.ge_wino3x3_wgrad_workspace(B, Cin, Cout, H, W), device=dev)
    wsd = torch.empty(lib.ge_conv2d_wgrad_workspace(B, Cin, Cout, H, W, 3, 3, 1), device=dev)
    dw, dwd = torch.empty(Cout, Cin, 3, 3, device=dev), torch.empty(Cout, Cin, 3, 3, device=dev)
    fw = lambda: check(lib.ge_wino3x3_wgrad(p(x), p(dy), p(dw), p(ws), B, Cin, Cout, H, W, 0, None), "wino wgrad")
    fd = lambda: check(lib.ge_conv2d_wgrad(p(x), p(dy), p(dwd), p(wsd), B, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, 0, None), "direct wgrad")
    fw(); fd()
    nb = min(B, 2)      # error on a two-frame problem of the same layer (the fp64 reference of the full batch would take minutes)
    dw2, dwd2 = torch.empty_like(dw), torch.empty_like(dw)
    e_w = e_d = float("nan")
    if lib.ge_wino3x3_wgrad_covered(nb, Cin, Cout, H, W):
        ws2 = torch.empty(lib.ge_wino3x3_wgrad_workspace(nb, Cin, Cout, H, W), device=dev)
        wsd2 = torch.empty(lib.ge_conv2d_wgrad_workspace(nb, Cin, Cout, H, W, 3, 3, 1), device=dev)
        check(lib.ge_wino3x3_wgrad(p(x), p(dy), p(dw2), p(ws2), nb, Cin, Cout, H, W, 0, None), "w2")
        check(lib.ge_conv2d_wgrad(p(x), p(dy), p(dwd2), p(wsd2), nb, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, 0, None), "d2")
        ref = torch.nn.grad.conv2d_weight(x[:nb].double(), (Cout, Cin, 3, 3), dy[:nb].double(), padding=1)
        e_w = ((dw2.double() - ref).abs().max() / ref.abs().max()).item()
        e_d = ((dwd2.double() - ref).abs().max() / ref.abs().max()).item()
    agree = ((dw.double() - dwd.double()).abs().max() / dwd.double().abs().max()).item()
    tw, td = timeit(fw), timeit(fd)
    print(f"{Cin}->{Cout} @{S}x{S}x{B} [{lib.ge_wino3x3_wgrad_splits(B, Cin, Cout, H, W)} splits{'' if routed else ', NOT routed'}]: wino {tw * 1e3:.3f} ms ({flops / tw / 1e12:.0f} TF eff, "
          f"{flops * 16 / 36 / tw / 157.3e12:.3f} of the MFMA peak executed, err {e_w:.1e}) direct {td * 1e3:.3f} ms ({flops / td / 1e12:.0f} TF, err {e_d:.1e}) "
          f"x{td / tw:.2f}; full batch wino vs direct {agree:.1e}", flush=True)
