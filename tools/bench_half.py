"""Microbench of the fp16-ACTIVATION-STORAGE kernels (csrc/ge_half.hip) on config 5's VGG16 layers (48 frames of 256x256):
forward / data gradient / weight gradient TFLOP/s of every distinct 3x3 layer next to the fp32-storage f16 kernels of
ge_mfma_f16.hip, and the BatchNorm / pooling passes in GB/s.  python tools/bench_half.py [frames] [--json out]"""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from graphecho_amd import functional as GF  # noqa: E402
from graphecho_amd import half as GH  # noqa: E402
from graphecho_amd._lib import lib, check  # noqa: E402

LAYERS = [(64, 64, 256), (64, 128, 128), (128, 128, 128), (128, 256, 64), (256, 256, 64), (256, 512, 32), (512, 512, 32),
          (512, 512, 16)]


def timeit(fn, iters=10, warm=3):
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


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 48
    out = []
    dev = torch.device("cuda:0")
    st = None
    for Cin, Cout, S in LAYERS:
        B, H, W = frames, S, S
        flops = 2.0 * B * H * W * Cout * Cin * 9
        x = torch.randn(B, Cin, H, W, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5)
        h = GH.to_blocked(x)
        z = torch.empty(B, Cout // 32, H, W, 32, device=dev, dtype=torch.float16)
        dz = (torch.randn(B, Cout // 32, H, W, 32, device=dev) * 0.1).half()
        dh = torch.empty_like(h)
        wp = GF._pack_weight_lp(w, 1, False, "f16")
        wpt = GF._pack_weight_lp(w, 1, True, "f16")
        stats = torch.empty(Cout, lib.ge_h_conv3x3_stat_parts(B, H, W), 3, device=dev)
        ws = torch.empty(lib.ge_h_conv3x3_wgrad_workspace(B, Cin, Cout, H, W), device=dev)
        dw = torch.empty_like(w)
        p = lambda t: t.data_ptr()
        rec = {"layer": f"{Cin}->{Cout}@{S}x{S}x{B}", "gflop": round(flops / 1e9, 1)}
        t = timeit(lambda: check(lib.ge_h_conv3x3_fwd(p(h), p(wp), None, p(z), p(stats), B, Cin, Cout, H, W, st), "f"))
        rec["h_fwd_tflops"] = round(flops / t / 1e12, 1)
        t = timeit(lambda: check(lib.ge_h_conv3x3_dgrad(p(dz), p(wpt), p(dh), B, Cin, Cout, H, W, st), "d"))
        rec["h_dgrad_tflops"] = round(flops / t / 1e12, 1)
        t = timeit(lambda: check(lib.ge_h_conv3x3_wgrad(p(h), p(dz), p(dw), p(ws), B, Cin, Cout, H, W, 1.0, None, 0, st), "w"))
        rec["h_wgrad_tflops"] = round(flops / t / 1e12, 1)
        # the weight-gradient kernel and its slab reduce apart (the library records an event between the two launches)
        e0, mid, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        mid.record()
        lib.ge_set_wgrad_split_event(mid.cuda_event)
        e0.record()
        check(lib.ge_h_conv3x3_wgrad(p(h), p(dz), p(dw), p(ws), B, Cin, Cout, H, W, 1.0, None, 0, st), "w")
        e1.record()
        lib.ge_set_wgrad_split_event(None)
        torch.cuda.synchronize()
        rec["h_wgrad_kernel_us"], rec["h_wgrad_reduce_us"] = round(1e3 * e0.elapsed_time(mid), 1), round(1e3 * mid.elapsed_time(e1), 1)
        rec["h_wgrad_kernel_tflops"] = round(flops / (e0.elapsed_time(mid) * 1e-3) / 1e12, 1)
        rec["h_wgrad_slab_mb"] = round(4 * ws.numel() / 1e6, 1)
        # the fp32-storage f16 kernels on the same layer
        y = torch.empty(B, Cout, H, W, device=dev)
        dy = torch.randn(B, Cout, H, W, device=dev)
        dx = torch.empty_like(x)
        lp = GF.lp_fns("f16")
        parts = lp["fwd_stat_parts"](B, Cout, H, W, 1)
        stats2 = torch.empty(Cout, parts, 3, device=dev)
        t = timeit(lambda: check(lp["fwd"](p(x), p(wp), None, p(y), p(stats2), B, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, 0, st), "f"))
        rec["lp_fwd_tflops"] = round(flops / t / 1e12, 1)
        t = timeit(lambda: check(lp["dgrad"](p(dy), p(wpt), None, p(dx), B, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, st), "d"))
        rec["lp_dgrad_tflops"] = round(flops / t / 1e12, 1)
        ws2 = torch.empty(lp["wgrad_workspace"](B, Cin, Cout, H, W, 3, 3, 1), device=dev)
        t = timeit(lambda: check(lp["wgrad"](p(x), p(dy), p(dw), p(ws2), B, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, 0, st), "w"))
        rec["lp_wgrad_tflops"] = round(flops / t / 1e12, 1)
        # BatchNorm passes on z (Cout channels): fp16 blocked vs fp32
        C, HW = Cout, H * W
        mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        a = torch.empty_like(z)
        n = z.numel()
        t = timeit(lambda: check(lib.ge_h_bn_apply(p(z), p(mean), p(invstd), p(gamma), p(beta), p(a), B, C, HW, 1, st), "a"))
        rec["h_bn_apply_gbs"] = round(4 * n / t / 1e9)
        part = torch.empty(C * B * lib.ge_h_bn_slices(HW) * 2, device=dev)
        sums = torch.empty(C, 2, device=dev)
        t = timeit(lambda: check(lib.ge_h_bn_bwd_reduce(p(dz), p(z), p(mean), p(invstd), p(gamma), p(beta), 1, p(part), p(sums),
                                                        None, None, 0, 1.0, None, B, C, HW, st), "r"))
        rec["h_bn_bwd_reduce_gbs"] = round(4 * n / t / 1e9)
        t = timeit(lambda: check(lib.ge_h_bn_bwd_apply(p(dz), p(z), p(mean), p(invstd), p(gamma), p(beta), 1, p(sums), 1.0 / (B * HW), 1.0, None,
                                                       p(a), B, C, HW, st), "b"))
        rec["h_bn_bwd_apply_gbs"] = round(6 * n / t / 1e9)
        yh = torch.empty(B, Cout // 32, H // 2, W // 2, 32, device=dev, dtype=torch.float16)
        t = timeit(lambda: check(lib.ge_h_maxpool2_fwd(p(z), p(yh), B, C, H, W, st), "p"))
        rec["h_pool_fwd_gbs"] = round(2.5 * n / t / 1e9)
        t = timeit(lambda: check(lib.ge_h_from_f32(p(y), p(z), B, C, HW, 1.0, None, st), "c"))
        rec["h_from_f32_gbs"] = round(6 * n / t / 1e9)
        t = timeit(lambda: check(lib.ge_h_to_f32(p(z), p(y), B, C, HW, 1.0, None, st), "c"))
        rec["h_to_f32_gbs"] = round(6 * n / t / 1e9)
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del x, h, z, dz, dh, y, dy, dx, ws, ws2, a
        torch.cuda.empty_cache()
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
