"""Race hunt for the blocked-fp16 kernels of ge_half.hip (LDS-DMA operands with hand-counted vmcnt, three rotating weight
stages, transpose reads in the weight gradient): every shape is launched N times -- forward (+ epilogue moments), data
gradient, weight gradient, fp32-epilogue forms -- while a second stream keeps the chip busy with unrelated traffic, and each
result must equal the first one bit for bit; the first one is checked against torch on the CPU (fp16-rounded operands).
usage: stress_half.py [repeats]"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd import half as GH
from graphecho_amd._lib import lib, check

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = lambda t: None if t is None else t.data_ptr()


def blk(x):
    B, C, H, W = x.shape
    return x.view(B, C // 32, 32, H, W).permute(0, 1, 3, 4, 2).contiguous().half()


def unblk(h):
    B, CB, H, W, _ = h.shape
    return h.float().permute(0, 1, 4, 2, 3).reshape(B, CB * 32, H, W)


side = torch.cuda.Stream()
noise_a = torch.randn(64 << 20, device=dev)
noise_b = torch.empty_like(noise_a)
bad = 0
# B, Cin, Cout, H, W: every tile variant the host picks (256 x 128 wide tiles, 128 x 128, 64-channel tiles, the 16- / 32-column
# single-buffer weight gradient, W = 128 halos)
for (B, Cin, Cout, H, W) in [(48, 64, 64, 256, 256), (16, 64, 128, 128, 128), (16, 128, 256, 64, 64), (16, 256, 512, 32, 32),
                             (48, 512, 512, 32, 32), (48, 512, 512, 16, 16), (3, 64, 192, 32, 64), (2, 128, 128, 16, 128),
                             (6, 256, 256, 32, 32)]:
    if not GH.supported(B, Cin, Cout, H, W):
        print(f"B{B} {Cin}->{Cout} @{H}x{W}: not covered, skipped")
        continue
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    gy = torch.randn(B, Cout, H, W, generator=g) * 0.1
    h, dz = blk(x).to(dev), blk(gy).to(dev)
    wd = w.to(dev)
    wp, wpt = GF._pack_weight_lp(wd, 1, False, "f16"), GF._pack_weight_lp(wd, 1, True, "f16")
    nparts = lib.ge_h_conv3x3_stat_parts(B, H, W)
    ws = torch.empty(lib.ge_h_conv3x3_wgrad_workspace(B, Cin, Cout, H, W), device=dev)

    def run():
        z = torch.empty(B, Cout // 32, H, W, 32, device=dev, dtype=torch.float16)
        st = torch.empty(Cout, nparts, 3, device=dev)
        dh = torch.empty(B, Cin // 32, H, W, 32, device=dev, dtype=torch.float16)
        dw = torch.empty_like(wd)
        y32 = torch.empty(B, Cout, H, W, device=dev)
        check(lib.ge_h_conv3x3_fwd(p(h), p(wp), None, p(z), p(st), B, Cin, Cout, H, W, None), "fwd")
        check(lib.ge_h_conv3x3_dgrad(p(dz), p(wpt), p(dh), B, Cin, Cout, H, W, None), "dgrad")
        check(lib.ge_h_conv3x3_wgrad(p(h), p(dz), p(dw), p(ws), B, Cin, Cout, H, W, 1.0, None, 0, None), "wgrad")
        check(lib.ge_h_conv3x3_fwd_f32(p(h), p(wp), None, p(y32), None, B, Cin, Cout, H, W, None), "fwd_f32")
        return z, st, dh, dw, y32

    first = run()
    torch.cuda.synchronize()
    # reference on the rounded operands (CPU, fp32): forward and weight gradient of a slice (the CPU conv is slow)
    nb = min(B, 2)
    xr, wr, gr = unblk(h[:nb].cpu()), w.half().float(), unblk(dz[:nb].cpu())
    e_f = (unblk(first[0][:nb].cpu()) - F.conv2d(xr, wr, padding=1)).abs().max().item()
    e_d = (unblk(first[2][:nb].cpu()) - torch.nn.grad.conv2d_input(xr.shape, wr, gr, padding=1)).abs().max().item()
    diff = 0
    for i in range(N):
        with torch.cuda.stream(side):      # unrelated HBM traffic beside the kernels under test
            noise_b.copy_(noise_a)
        out = run()
        if not all(torch.equal(a, b) for a, b in zip(out, first)):
            diff += 1
    torch.cuda.synchronize()
    tol_f = 4e-3 * max(1.0, F.conv2d(xr, wr, padding=1).abs().max().item())
    bad += diff + (e_f > tol_f) + (e_d > 4e-3)
    print(f"B{B} {Cin}->{Cout} @{H}x{W}: fwd err {e_f:.1e} dgrad err {e_d:.1e}; {diff} of {N} repeats differ", flush=True)
print("FAILED" if bad else "all repeats bit-identical")
sys.exit(1 if bad else 0)
