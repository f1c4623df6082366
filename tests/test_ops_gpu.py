"""Parity of every HIP kernel (through the C ABI / autograd bindings) against plain PyTorch-CPU fp32 ops.

Tolerances: fp32 everywhere; contractions are compared at rtol 2e-4 of the output scale (accumulation order
differs from MKL), element-wise / normalisation kernels at 1e-5..1e-4, indices bit-exact.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(a, b, rtol=2e-4, atol=None, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= (atol or 0.0) + rtol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.3e}, rtol {rtol:g})"


def grads(fn, inputs, gout):
    ins = [t.clone().requires_grad_(True) if t is not None and t.is_floating_point() else t for t in inputs]
    out = fn(*ins)
    out.backward(gout.to(out.device))
    return out, [None if (t is None or not t.is_floating_point()) else t.grad for t in ins]


CONV_CASES = [
    # B, Cin, H, W, Cout, k, s, p, groups, bias
    (2, 3, 32, 32, 64, 7, 2, 3, 1, False),
    (2, 64, 16, 16, 64, 3, 1, 1, 1, False),
    (2, 256, 16, 16, 256, 3, 1, 1, 1, True),
    (2, 256, 16, 16, 128, 3, 1, 1, 1, True),
    (2, 128, 16, 16, 128, 3, 2, 1, 1, False),
    (3, 64, 8, 8, 256, 1, 1, 0, 1, False),
    (2, 256, 16, 16, 512, 1, 2, 0, 1, False),
    (2, 128, 8, 8, 4, 1, 1, 0, 1, True),
    (2, 512, 9, 7, 256, 1, 1, 0, 4, True),   # grouped 1x1 (BasicConv), odd spatial size
    (2, 20, 11, 13, 24, 3, 1, 1, 1, True),   # ragged everything
    (2, 22, 12, 12, 40, 3, 1, 1, 1, False),  # K = 198: multiple of 18 but not of 36 (exact-K loader only on some tiles)
    (2, 26, 10, 10, 48, 1, 1, 0, 1, True),   # 1x1, K = 26: not a multiple of the chunk (general loader)
    (2, 48, 10, 10, 26, 1, 1, 0, 1, True),   # 1x1, K = 48 forward (exact) / 26 data-gradient (general), M ragged
    (3, 36, 9, 9, 36, 3, 2, 1, 2, True),     # grouped + strided: K = 162 per group, parity-decomposed data gradient
    (1, 256, 8, 8, 256, 3, 2, 0, 1, True),   # TGCN.prediction: 3x3 s2 p0
    (2, 8, 12, 12, 8, 5, 1, 2, 1, True),     # generic kernel size path
    (2, 96, 15, 13, 160, 3, 2, 1, 1, False),  # stride-2 3x3 on odd maps: parity-decomposed data gradient
    (3, 40, 9, 7, 72, 1, 2, 0, 1, False),     # stride-2 1x1 on odd maps: dense GEMM scattered to even positions
    (2, 64, 14, 14, 64, 3, 2, 1, 2, True),    # stride-2 3x3, grouped
    (2, 256, 64, 64, 1, 3, 1, 1, 1, True),    # Discriminator.cls_logits at p2: the one-output-channel kernels (ge_conv_c1.hip)
    (3, 17, 13, 12, 1, 3, 1, 1, 1, False),    # ... odd H (the last row pair is half empty), ragged channel groups
    (3, 17, 13, 9, 1, 3, 1, 1, 1, False),     # ... W not a multiple of 4: GEMM path
    (5, 256, 8, 8, 1, 3, 1, 1, 1, True),      # ... p5: small map, stays on the GEMM path (M = 1)
    (2, 24, 40, 36, 1, 3, 1, 1, 1, True),     # ... 1440 positions, 24 channels = 3 per wave
    (2, 256, 16, 16, 256, 3, 1, 1, 1, False),  # 16x16 stage at 2 frames: split-K forward and data gradient
    (4, 512, 8, 8, 512, 3, 1, 1, 1, True),    # 8x8 stage: split-K with bias (added by split 0 only)
    (4, 2048, 8, 8, 512, 1, 1, 0, 1, False),  # 1x1 with long K on the 8x8 stage: split-K on the exact loader
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(dev, case):
    from graphecho_amd import functional as GF

    B, Cin, H, W, Cout, k, s, p, g, has_bias = case
    gen = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin // g, k, k, generator=gen) / math.sqrt(Cin // g * k * k)
    b = torch.randn(Cout, generator=gen) if has_bias else None
    ref, (rdx, rdw, rdb) = grads(lambda x, w, b: F.conv2d(x, w, b, s, p, 1, g), [x, w, b],
                                 gout := torch.randn(*F.conv2d(x, w, b, s, p, 1, g).shape, generator=gen))
    out, (dx, dw, db) = grads(lambda x, w, b: GF.conv2d(x, w, b, s, p, g),
                              [x.to(dev), w.to(dev), None if b is None else b.to(dev)], gout)
    close(out, ref, what="conv fwd")
    close(dx, rdx, what="conv dgrad")
    close(dw, rdw, what="conv wgrad")
    if has_bias:
        close(db, rdb, what="conv bias grad")


WGRAD3X3_CASES = [   # B, Cin, H, W, Cout: 3x3 / stride 1 / pad 1 shapes of the patch-staged weight-gradient kernel
    (2, 256, 64, 64, 256),    # 128 x 128 tiles, WC = 32, two chunks per row (the FPN head's shape)
    (3, 40, 32, 32, 72),      # 64 x 64 tiles, WC = 32, one chunk per row, ragged M / J (40 * 9 = 360 = 5.6 tiles)
    (2, 256, 16, 16, 200),    # 128 x 128 tiles, WC = 16 (two rows per chunk), M ragged
    (4, 24, 16, 16, 64),      # 64 x 64 tiles, WC = 16
    (4, 512, 8, 8, 512),      # 128 x 128 tiles, WC = 8 (four rows per chunk): backbone layer4
    (8, 20, 8, 8, 36),        # 64 x 64 tiles, WC = 8
    (1, 130, 24, 64, 130),    # non-square map, channel count that is no multiple of anything
    (2, 64, 12, 8, 64),       # H * W = 96 = 3 chunks per image, WC = 8
    (1, 32, 5, 8, 32),        # H * W = 40 is not a multiple of 32: falls back to the per-tap loader
    (2, 16, 20, 20, 16),      # W = 20: per-tap loader
]


@pytest.mark.parametrize("B,Cin,H,W,Cout", WGRAD3X3_CASES)
def test_conv2d_wgrad_3x3_patch_kernel(dev, B, Cin, H, W, Cout):
    """conv_wgrad3x3_kernel (X operand staged as a halo'd patch) against torch: every (tile, WC) instantiation, ragged
    tiles, image borders on every side; and bit-identical to the per-tap kernel's K-order? no -- same split-K plan,
    same chunk order, so the two kernels agree to the last bit (GE_WGRAD_PATCH=0 selects the per-tap one; checked by
    tools/bench_wgrad3x3.py on the GPU box, here the reference is torch)."""
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(B * 1000 + Cin)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / math.sqrt(Cin * 9)
    gout = torch.randn(B, Cout, H, W, generator=gen)
    _, (_, rdw) = grads(lambda x, w: F.conv2d(x, w, None, 1, 1), [x, w], gout)
    _, (_, dw) = grads(lambda x, w: GF.conv2d(x, w, None, 1, 1), [x.to(dev), w.to(dev)], gout)
    close(dw, rdw, what="3x3 wgrad")


@pytest.mark.parametrize("shape", [(2, 16, 9, 7, 40, 3, 1, 1), (3, 64, 16, 16, 200, 1, 1, 0), (2, 32, 4, 4, 96, 3, 2, 1)])
def test_conv2d_fused_bn_statistics(dev, shape):
    """conv2d(bn_stats=True): the per-tile moments written by the conv epilogue merge to the batch statistics of y
    (incl. ragged tiles where a wave owns no valid column)."""
    from graphecho_amd import functional as GF

    B, Cin, H, W, Cout, k, s_, p = shape
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=gen) + 0.5
    w = torch.randn(Cout, Cin, k, k, generator=gen) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=gen)
    y_ref = F.conv2d(x, w, b, s_, p)
    y, part = GF.conv2d(x.to(dev), w.to(dev), b.to(dev), s_, p, 1, None, True)
    close(y, y_ref, what="conv fwd with stats")
    rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    out = GF.batch_norm(y, None, None, rm, rv, True, 0.1, 1e-5, None, False, None, part)
    ref = F.batch_norm(y_ref, torch.zeros(Cout), torch.ones(Cout), None, None, True, 0.1, 1e-5)
    close(out, ref, 2e-4, what="bn from fused stats")
    close(rm, 0.1 * y_ref.mean((0, 2, 3)), 2e-4, what="running mean from fused stats")


def test_conv2d_dgrad_with_skip_addend(dev):
    """ge_conv2d_dgrad(addend=...) returns dgrad + addend (used to merge skip-connection gradients in the epilogue)."""
    from graphecho_amd._lib import lib, check
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(21)
    for (B, Cin, H, W, Cout, k, s_, p) in [(2, 64, 16, 16, 128, 1, 1, 0), (2, 64, 16, 16, 128, 1, 2, 0),
                                           (2, 48, 12, 12, 48, 3, 1, 1), (2, 48, 13, 11, 48, 3, 2, 1)]:
        x = torch.randn(B, Cin, H, W, generator=gen, requires_grad=True)
        w = torch.randn(Cout, Cin, k, k, generator=gen) / math.sqrt(Cin * k * k)
        y = F.conv2d(x, w, None, s_, p)
        gy = torch.randn(*y.shape, generator=gen)
        add = torch.randn(B, Cin, H, W, generator=gen)
        (ref,) = torch.autograd.grad(y, x, gy)
        wd, gyd, addd = w.to(dev), gy.to(dev), add.to(dev)
        wp = GF._pack_weight(wd, 1, True)
        dx = torch.empty(B, Cin, H, W, device=dev)
        check(lib.ge_conv2d_dgrad(gyd.data_ptr(), wp.data_ptr(), addd.data_ptr(), dx.data_ptr(), B, Cin, H, W, Cout,
                                  y.shape[2], y.shape[3], k, k, s_, p, 1, None))
        close(dx, ref + add, what=f"dgrad+addend k{k}s{s_}")


@pytest.mark.parametrize("B,Cin,H,W,Cout,bias", [(8, 256, 64, 64, 256, True),     # 128x128 tiles (2048 of them, K = 256)
                                                  (32, 64, 64, 64, 256, False),    # short K: 64x128 tiles
                                                  (3, 32, 20, 20, 4, True),        # M = 4, N = 1200: partial tiles on both operands
                                                  (2, 48, 12, 28, 72, False),      # M, N, plane no multiples of the tile
                                                  (16, 128, 32, 32, 512, True)])
def test_conv2d_1x1_direct_to_lds_loader(dev, B, Cin, H, W, Cout, bias):
    """1x1 / stride-1 layers on the direct-to-LDS loader (buffer_load_dwordx4 ... lds, three stages, the kernel's own vmcnt
    accounting): forward with bias + fused BatchNorm moments, data gradient with a skip addend, against fp64 on the CPU at the
    fp32 contraction tolerance -- including partial tiles, where out-of-range lanes must land ZEROS in LDS -- and the launch
    really is the LDSD instantiation (last template argument true)."""
    from graphecho_amd import functional as GF
    from graphecho_amd._lib import lib, check

    gen = torch.Generator().manual_seed(B + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 1, 1, generator=gen) / Cin ** 0.5
    b = torch.randn(Cout, generator=gen) if bias else None
    g = torch.randn(B, Cout, H, W, generator=gen)
    add = torch.randn(B, Cin, H, W, generator=gen)
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double())
    ref_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), g.double()) + add.double()
    y, stats = GF.conv2d(x.to(dev), w.to(dev), None if b is None else b.to(dev), 1, 0, 1, GF.PackCache(), True)
    name = lib.ge_last_conv_kernel().decode()
    assert name.startswith("conv_gemm_kernel<") and name.endswith("true, true, true>"), name
    close(y, ref, what="1x1 forward (direct-to-LDS)")
    rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    out = GF.batch_norm(y, None, None, rm, rv, True, 0.1, 1e-5, None, False, None, stats)
    close(out, F.batch_norm(ref, None, None, None, None, True, 0.1, 1e-5), what="BatchNorm from the fused moments")
    close(rm, 0.1 * ref.mean((0, 2, 3)), what="running mean from the fused moments")
    wp = GF._pack_weight(w.to(dev), 1, True)
    dx = torch.empty(B, Cin, H, W, device=dev)
    gd, addd = g.to(dev), add.to(dev)
    check(lib.ge_conv2d_dgrad(gd.data_ptr(), wp.data_ptr(), addd.data_ptr(), dx.data_ptr(), B, Cin, H, W, Cout, H, W, 1, 1, 1, 0,
                              1, None))
    if Cout % 16 == 0:      # the data gradient contracts over Cout: whole 16-deep chunks needed
        assert lib.ge_last_conv_kernel().decode().endswith("true, true, true>")
    close(dx, ref_dx, what="1x1 data gradient + skip addend (direct-to-LDS)")
    check(lib.ge_conv2d_dgrad(gd.data_ptr(), wp.data_ptr(), None, dx.data_ptr(), B, Cin, H, W, Cout, H, W, 1, 1, 1, 0, 1, None))
    close(dx, ref_dx - add.double(), what="1x1 data gradient without addend (accumulator registers stored as they are)")


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s_,p,G", [(4, 64, 16, 16, 256, 1, 1, 0, 1), (3, 48, 12, 10, 40, 1, 1, 0, 4), (2, 32, 13, 11, 24, 3, 2, 1, 1),
                                                       (2, 3, 32, 32, 16, 7, 2, 3, 1), (8, 256, 32, 32, 1024, 1, 1, 0, 1)])
def test_conv2d_wgrad_with_bias_gradient(dev, B, Cin, H, W, Cout, k, s_, p, G):
    """ge_conv2d_wgrad_bias: the bias gradient as row sums of the dY tile inside the weight-gradient pass (general MFMA kernel:
    1x1, grouped, strided 3x3, 7x7), overwrite and accumulate, against torch; the 3x3 / s1 / p1 patch kernel declines."""
    from graphecho_amd._lib import lib, check

    gen = torch.Generator().manual_seed(B * 7 + Cout)
    x = torch.randn(B, Cin, H, W, generator=gen, requires_grad=True)
    w = (torch.randn(Cout, Cin // G, k, k, generator=gen) / math.sqrt(Cin // G * k * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=gen, requires_grad=True)
    y = F.conv2d(x, w, b, s_, p, 1, G)
    gy = torch.randn(*y.shape, generator=gen)
    y.backward(gy)
    Ho, Wo = y.shape[2:]
    key = (B, Cin, Cout, H, W, Ho, Wo, k, k, s_, p, G)
    assert lib.ge_conv2d_wgrad_fuses_bias(*key) == 1
    assert lib.ge_conv2d_wgrad_fuses_bias(8, 64, 64, 32, 32, 32, 32, 3, 3, 1, 1, 1) == 0
    xd, gyd = x.detach().to(dev), gy.to(dev)
    ws = torch.empty(lib.ge_conv2d_wgrad_workspace(B, Cin, Cout, Ho, Wo, k, k, G), device=dev)
    dw = torch.full((Cout, Cin // G, k, k), 7.0, device=dev)
    db = torch.full((Cout,), -3.0, device=dev)
    args = (xd.data_ptr(), gyd.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, Cin, H, W, Cout, Ho, Wo, k, k, s_, p, G)
    check(lib.ge_conv2d_wgrad_bias(*args, 0, None))
    close(dw, w.grad, what="dw (overwrite)")
    close(db, b.grad, what="db (overwrite)")
    check(lib.ge_conv2d_wgrad_bias(*args, 1, None))
    close(dw, 2 * w.grad, what="dw (accumulate)")
    close(db, 2 * b.grad, what="db (accumulate)")


def test_conv2d_fpn_shape_batch(dev):
    """The dominant FPN shape (256->256 3x3 @64x64) at batch 2, forward only (CPU reference stays cheap)."""
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(7)
    x = torch.randn(2, 256, 64, 64, generator=gen)
    w = torch.randn(256, 256, 3, 3, generator=gen) / 48.0
    b = torch.randn(256, generator=gen)
    close(GF.conv2d(x.to(dev), w.to(dev), b.to(dev), 1, 1, 1), F.conv2d(x, w, b, 1, 1), what="conv 256@64^2")


def test_syncbn_segment_kernels_against_float64(dev):
    """The SyncBN kernels that take all segments of a concatenated batch per launch (round 6), called through the C ABI with a
    hand-made two-"rank" exchange: ge_bn_stats_channel_segs (moments from x) and ge_bn_finalize_segs (moments from conv-epilogue
    triples) against float64 per segment; ge_bn_fwd_channel_segs_sync on a gathered buffer holding the segment's moments split
    over two ranks; ge_bn_bwd_reduce_channel_segs / ge_bn_bwd_apply_channel_segs against autograd through the float64 batch norm."""
    import ctypes

    from graphecho_amd._lib import lib, check

    gen = torch.Generator().manual_seed(5)
    B, C, H, W = 6, 48, 12, 8
    HW, bounds = H * W, [(0, 2), (2, 4)]
    S = len(bounds)
    x = (torch.randn(B, C, H, W, generator=gen) * 2 + 0.5).to(dev)
    seg = (ctypes.c_int * (4 * S))(*[v for b0, bs in bounds for v in (b0, bs, 0, 0)])
    stats = torch.empty(S, C, 3, device=dev)
    check(lib.ge_bn_stats_channel_segs(x.data_ptr(), seg, S, C, HW, stats.data_ptr(), None))
    xd = x.double().cpu()
    for s, (b0, bs) in enumerate(bounds):
        xs = xd[b0:b0 + bs]
        close(stats[s, :, 0], torch.full((C,), float(bs * HW)), 0, what="count")
        close(stats[s, :, 1], xs.mean((0, 2, 3)), 1e-5, what="mean from x")
        close(stats[s, :, 2], ((xs - xs.mean((0, 2, 3), keepdim=True)) ** 2).sum((0, 2, 3)), 1e-5, what="M2 from x")
    # conv-epilogue style triples: one per (channel, 16 positions of one frame); segment s owns triples [s0, s0 + n)
    width = 16
    tri = xd.reshape(B, C, HW // width, width)
    part = torch.stack([torch.full(tri.shape[:3], float(width), dtype=torch.float64), tri.mean(-1),
                        ((tri - tri.mean(-1, keepdim=True)) ** 2).sum(-1)], -1)          # [B][C][HW/16][3]
    part = part.permute(1, 0, 2, 3).reshape(C, -1, 3).float().contiguous().to(dev)       # [C][nb][3]
    nb = part.shape[1]
    seg2 = (ctypes.c_int * (4 * S))(*[v for b0, bs in bounds for v in (b0, bs, b0 * HW // width, bs * HW // width)])
    stats2 = torch.empty(S, C, 3, device=dev)
    check(lib.ge_bn_finalize_segs(part.data_ptr(), nb * 3, 3, seg2, S, C, HW, stats2.data_ptr(), None))
    close(stats2[..., 1], stats[..., 1], 1e-5, what="mean from triples")
    close(stats2[..., 2], stats[..., 2], 1e-4, what="M2 from triples")
    # "two ranks": this rank's segment moments + a second rank holding OTHER frames; the merged statistics must be those of both
    x_other = (torch.randn(B, C, H, W, generator=gen) - 0.3).to(dev)
    stats_o = torch.empty(S, C, 3, device=dev)
    check(lib.ge_bn_stats_channel_segs(x_other.data_ptr(), seg, S, C, HW, stats_o.data_ptr(), None))
    gathered = torch.stack([stats, stats_o]).contiguous()                                 # [world = 2][S][C][3]
    gamma, beta = torch.rand(C, generator=gen).to(dev) + 0.5, torch.randn(C, generator=gen).to(dev)
    y, mean, invstd = torch.empty_like(x), torch.empty(S, C, device=dev), torch.empty(S, C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    check(lib.ge_bn_fwd_channel_segs_sync(x.data_ptr(), gathered.data_ptr(), 2, seg, S, gamma.data_ptr(), beta.data_ptr(), None,
                                          y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), rm.data_ptr(), rv.data_ptr(), C, HW,
                                          1e-5, 0.1, 1, None))
    xo = x_other.double().cpu()
    gd, bd = gamma.double().cpu(), beta.double().cpu()
    for s, (b0, bs) in enumerate(bounds):
        both = torch.cat([xd[b0:b0 + bs], xo[b0:b0 + bs]])
        mu, var = both.mean((0, 2, 3)), both.var((0, 2, 3), unbiased=False)
        close(mean[s], mu, 1e-5, what="merged mean")
        close(invstd[s], (var + 1e-5).rsqrt(), 1e-5, what="merged invstd")
        ref = torch.relu((xd[b0:b0 + bs] - mu[None, :, None, None]) * (var + 1e-5).rsqrt()[None, :, None, None] *
                         gd[None, :, None, None] + bd[None, :, None, None])
        close(y[b0:b0 + bs], ref, 1e-5, what="y of segment")
    # backward of one rank alone (world 1): sums, then dx against autograd
    mean1, invstd1 = torch.empty(S, C, device=dev), torch.empty(S, C, device=dev)
    for s, (b0, bs) in enumerate(bounds):
        xs = xd[b0:b0 + bs]
        mean1[s] = xs.mean((0, 2, 3)).float().to(dev)
        invstd1[s] = (xs.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt().float().to(dev)
    dy = torch.randn(B, C, H, W, generator=gen).to(dev)
    sums, dgamma, dbeta = torch.empty(S, C, 2, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
    check(lib.ge_bn_bwd_reduce_channel_segs(dy.data_ptr(), x.data_ptr(), None, mean1.data_ptr(), invstd1.data_ptr(), gamma.data_ptr(),
                                            beta.data_ptr(), 1, sums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), 0, seg, S, C,
                                            HW, None))
    inv = (ctypes.c_float * S)(*[1.0 / (bs * HW) for _b0, bs in bounds])
    dx = torch.empty_like(x)
    check(lib.ge_bn_bwd_apply_channel_segs(dy.data_ptr(), x.data_ptr(), None, mean1.data_ptr(), invstd1.data_ptr(), gamma.data_ptr(),
                                           beta.data_ptr(), 1, sums.data_ptr(), inv, seg, S, dx.data_ptr(), None, C, HW, None))
    dg_ref, db_ref = torch.zeros(C, dtype=torch.float64), torch.zeros(C, dtype=torch.float64)
    for s, (b0, bs) in enumerate(bounds):
        xs = xd[b0:b0 + bs].clone().requires_grad_(True)
        g2, b2 = gd.clone().requires_grad_(True), bd.clone().requires_grad_(True)
        out = torch.relu(F.batch_norm(xs, None, None, g2, b2, True, 0.1, 1e-5))
        gx, gg, gb = torch.autograd.grad(out, [xs, g2, b2], dy[b0:b0 + bs].double().cpu())
        close(dx[b0:b0 + bs], gx, 2e-5, what="dx of segment")
        dg_ref += gg
        db_ref += gb
    close(dgamma, dg_ref, 2e-5, what="dgamma over the segments")
    close(dbeta, db_ref, 2e-5, what="dbeta over the segments")


@pytest.mark.parametrize("shape", [(4, 64, 16, 16), (3, 20, 7, 5), (2, 256, 8, 8)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
def test_batch_norm(dev, shape, relu, res):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(1)
    C = shape[1]
    x = torch.randn(*shape, generator=gen) * 2 + 0.5
    gam = torch.rand(C, generator=gen) + 0.5
    bet = torch.randn(C, generator=gen)
    r = torch.randn(*shape, generator=gen) if res else None
    gout = torch.randn(*shape, generator=gen)

    def ref_fn(x, gam, bet, r):
        y = F.batch_norm(x, rm_c, rv_c, gam, bet, True, 0.1, 1e-5)
        if r is not None:
            y = y + r
        return F.relu(y) if relu else y

    rm_c, rv_c = torch.zeros(C), torch.ones(C)
    ref, rg = grads(ref_fn, [x, gam, bet, r], gout)
    rm_g, rv_g = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    out, gg = grads(lambda x, gam, bet, r: GF.batch_norm(x, gam, bet, rm_g, rv_g, True, 0.1, 1e-5, r, relu),
                    [x.to(dev), gam.to(dev), bet.to(dev), None if r is None else r.to(dev)], gout)
    close(out, ref, 1e-4, what="bn fwd")
    close(rm_g, rm_c, 1e-4, what="running_mean")
    close(rv_g, rv_c, 1e-4, what="running_var")
    for a, b_, n in zip(gg, rg, ("dx", "dgamma", "dbeta", "dres")):
        if b_ is not None:
            close(a, b_, 2e-4, what="bn " + n)
    # eval mode
    ev = GF.batch_norm(x.to(dev), gam.to(dev), bet.to(dev), rm_g, rv_g, False, 0.1, 1e-5)
    close(ev, F.batch_norm(x, rm_c, rv_c, gam, bet, False, 0.1, 1e-5), 1e-4, what="bn eval")


@pytest.mark.parametrize("shape,G", [((2, 128, 16, 16), 128), ((3, 256, 8, 8), 32), ((2, 64, 5, 7), 8),
                                     ((2, 64, 32, 32), 8),      # register-resident, 256 threads (L = 8192)
                                     ((2, 48, 64, 64), 6),      # register-resident, 1024 threads (L = 32768)
                                     ((2, 24, 16, 16), 24),     # one channel per group (FPN head form), L = 256
                                     ((1, 16, 64, 96), 2)])     # L = 49152: streaming fallback
@pytest.mark.parametrize("relu", [False, True])
def test_group_norm(dev, shape, G, relu):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(2)
    C = shape[1]
    x = torch.randn(*shape, generator=gen) * 1.5 + 0.3
    gam, bet = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    gout = torch.randn(*shape, generator=gen)
    f = lambda x, g, b: (F.relu(F.group_norm(x, G, g, b, 1e-5)) if relu else F.group_norm(x, G, g, b, 1e-5))
    ref, rg = grads(f, [x, gam, bet], gout)
    out, gg = grads(lambda x, g, b: GF.group_norm(x, G, g, b, 1e-5, relu), [x.to(dev), gam.to(dev), bet.to(dev)], gout)
    close(out, ref, 1e-4, what="gn fwd")
    for a, b_, n in zip(gg, rg, ("dx", "dgamma", "dbeta")):
        close(a, b_, 2e-4, what="gn " + n)


@pytest.mark.parametrize("R,D,affine", [(37, 256, True), (300, 256, False), (1, 70 * 90, False)])
def test_layer_norm(dev, R, D, affine):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(3)
    x = torch.randn(R, D, generator=gen) * 2 + 1
    gam = torch.rand(D, generator=gen) + 0.5 if affine else None
    bet = torch.randn(D, generator=gen) if affine else None
    gout = torch.randn(R, D, generator=gen)
    ref, rg = grads(lambda x, g, b: F.layer_norm(x, (D,), g, b, 1e-5), [x, gam, bet], gout)
    out, gg = grads(lambda x, g, b: GF.layer_norm(x, g, b, 1e-5),
                    [x.to(dev), None if gam is None else gam.to(dev), None if bet is None else bet.to(dev)], gout)
    close(out, ref, 1e-4, what="ln fwd")
    for a, b_, n in zip(gg, rg, ("dx", "dgamma", "dbeta")):
        if b_ is not None:
            close(a, b_, 2e-4, what="ln " + n)


@pytest.mark.parametrize("hi,ho", [((8, 8), (16, 16)), ((16, 16), (64, 64)), ((7, 5), (13, 11)), ((8, 8), (8, 8)),
                                    ((1, 1), (4, 4)), ((8, 8), (64, 64)), ((32, 32), (64, 64)), ((5, 9), (1, 33)),
                                    ((3, 300), (9, 301)), ((16, 16), (160, 160))])
@pytest.mark.parametrize("with_add", [False, True])
def test_upsample_bilinear(dev, hi, ho, with_add):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(4)
    x = torch.randn(2, 6, *hi, generator=gen)
    add = torch.randn(2, 6, *ho, generator=gen) if with_add else None
    gout = torch.randn(2, 6, *ho, generator=gen)

    def ref_fn(x, add):
        y = F.interpolate(x, size=ho, mode="bilinear", align_corners=True)
        return y + add if add is not None else y

    ref, rg = grads(ref_fn, [x, add], gout)
    out, gg = grads(lambda x, add: GF.upsample_bilinear(x, ho, add), [x.to(dev), None if add is None else add.to(dev)],
                    gout)
    close(out, ref, 2e-5, what="upsample fwd")      # align_corners weights in fp32: 1.6e-5 of the tensor maximum at 8 -> 8 (identity grid)
    close(gg[0], rg[0], 2e-5, what="upsample dx")
    if with_add:
        close(gg[1], rg[1], 1e-6, what="upsample dadd")


@pytest.mark.parametrize("planes,hi,ho", [((2, 7), (8, 8), (64, 64)), ((1, 5), (32, 32), (64, 64)), ((3, 11), (4, 4), (8, 8)),
                                          ((1, 1), (16, 16), (64, 64)), ((2, 9), (16, 8), (32, 64)), ((1, 3), (64, 64), (256, 256)),
                                          ((2, 5), (9, 9), (9, 30)), ((1, 2), (2, 2), (200, 3))])
def test_upsample_bilinear_backward_group_tails(dev, planes, hi, ho):
    """The streaming backward packs several planes into a workgroup: plane counts that leave a partial last group,
    one-plane groups, non-square scales, and the sizes of the FPN head (8/16/32 -> 64, 64 -> 256); run twice: the
    accumulation order is fixed, so the gradient is bit-identical run to run."""
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(41)
    x = torch.randn(*planes, *hi, generator=gen)
    gout = torch.randn(*planes, *ho, generator=gen)
    ref, rg = grads(lambda x: F.interpolate(x, size=ho, mode="bilinear", align_corners=True), [x], gout)
    out, gg = grads(lambda x: GF.upsample_bilinear(x, ho, None), [x.to(dev)], gout)
    _, gg2 = grads(lambda x: GF.upsample_bilinear(x, ho, None), [x.to(dev)], gout)
    close(out, ref, 1e-5, what="upsample fwd")
    close(gg[0], rg[0], 1e-5, what="upsample dx")
    assert torch.equal(gg[0], gg2[0])


@pytest.mark.parametrize("k,s,p,hw", [(3, 2, 1, (16, 16)), (2, 2, 0, (12, 10)), (3, 2, 1, (9, 7)), (3, 2, 1, (1, 5)),
                                      (3, 2, 1, (64, 33)), (3, 1, 1, (8, 9)), (5, 3, 2, (17, 13))])
def test_max_pool(dev, k, s, p, hw):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 5, *hw, generator=gen)
    ref0 = F.max_pool2d(x, k, s, p)
    gout = torch.randn(*ref0.shape, generator=gen)
    ref, rg = grads(lambda x: F.max_pool2d(x, k, s, p), [x], gout)
    out, gg = grads(lambda x: GF.max_pool2d(x, k, s, p), [x.to(dev)], gout)
    close(out, ref, 0, 0, what="maxpool fwd")
    close(gg[0], rg[0], 1e-6, what="maxpool bwd")


@pytest.mark.parametrize("r,hw", [(8, (64, 64)), (4, (32, 32)), (2, (9, 7))])
def test_avg_pool_and_global(dev, r, hw):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(6)
    x = torch.randn(2, 5, *hw, generator=gen)
    gout = torch.randn(*F.avg_pool2d(x, r, r).shape, generator=gen)
    ref, rg = grads(lambda x: F.avg_pool2d(x, r, r), [x], gout)
    out, gg = grads(lambda x: GF.avg_pool2d(x, r), [x.to(dev)], gout)
    close(out, ref, 1e-5, what="avgpool fwd")
    close(gg[0], rg[0], 1e-6, what="avgpool bwd")
    g1 = torch.randn(2, 5, 1, 1, generator=gen)
    ref, rg = grads(lambda x: F.adaptive_avg_pool2d(x, 1), [x], g1)
    out, gg = grads(lambda x: GF.adaptive_avg_pool2d_1(x), [x.to(dev)], g1)
    close(out, ref, 1e-5, what="global avg fwd")
    close(gg[0], rg[0], 1e-6, what="global avg bwd")


def test_activations(dev):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(8)
    x = torch.randn(3, 7, 11, generator=gen) * 3
    gout = torch.randn(3, 7, 11, generator=gen)
    for mine, ref_f in ((GF.relu, F.relu), (GF.gelu, F.gelu)):
        ref, rg = grads(ref_f, [x], gout)
        out, gg = grads(mine, [x.to(dev)], gout)
        close(out, ref, 1e-5, what="act fwd")
        close(gg[0], rg[0], 1e-5, what="act bwd")


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_matmul(dev, ta, tb):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(9)
    M, K, N = 150, 83, 70
    a = torch.randn((K, M) if ta else (M, K), generator=gen)
    b = torch.randn((N, K) if tb else (K, N), generator=gen)
    gout = torch.randn(M, N, generator=gen)
    f = lambda a, b: 0.5 * ((a.t() if ta else a) @ (b.t() if tb else b))
    ref, rg = grads(f, [a, b], gout)
    out, gg = grads(lambda a, b: GF.matmul(a, b, ta, tb, 0.5), [a.to(dev), b.to(dev)], gout)
    close(out, ref, what="matmul fwd")
    close(gg[0], rg[0], what="matmul da")
    close(gg[1], rg[1], what="matmul db")


# (278, 256, 512) / (261, 256, 256) / (1100, 256, 300): weight AND bias gradient in one launch (ge_gemm_rowsum: 32 x 32-tile
# kernel, K = rows >= 64); (5, ...) and (40, ...): too few rows, GEMM + column sum
@pytest.mark.parametrize("rows,i,o,bias", [(278, 256, 512, True), (5, 256, 1, True), (600, 512, 256, False), (261, 256, 256, True),
                                           (40, 256, 256, True), (1100, 256, 300, True)])
def test_linear(dev, rows, i, o, bias):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(10)
    x = torch.randn(rows, i, generator=gen)
    w = torch.randn(o, i, generator=gen) / math.sqrt(i)
    b = torch.randn(o, generator=gen) if bias else None
    gout = torch.randn(rows, o, generator=gen)
    ref, rg = grads(F.linear, [x, w, b], gout)
    out, gg = grads(GF.linear, [x.to(dev), w.to(dev), None if b is None else b.to(dev)], gout)
    close(out, ref, what="linear fwd")
    for a, b_, n in zip(gg, rg, ("dx", "dw", "db")):
        if b_ is not None:
            close(a, b_, what="linear " + n)


def test_softmax(dev):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(11)
    x = torch.randn(77, 333, generator=gen) * 4
    gout = torch.randn(77, 333, generator=gen)
    ref, rg = grads(lambda x: torch.softmax(x * 0.0625, -1), [x], gout)
    out, gg = grads(lambda x: GF.softmax_lastdim(x, 0.0625), [x.to(dev)], gout)
    close(out, ref, 1e-5, what="softmax fwd")
    close(gg[0], rg[0], 1e-5, what="softmax bwd")


KNN_CASES = [(2, 256, 64, 64, 9, 1, True), (2, 64, 300, 70, 9, 2, True), (1, 48, 200, None, 9, 1, False),
             (2, 256, 64, 64, 9, 1, "zeros"),
             # round 6 (query-side normalisation inside the top-k kernel): channels that are no multiple of the prologue's 128-channel
             # slab or of the 16-channel chunk, ragged row blocks, K > 16 (one row per wave), relative positions with candidates
             (2, 200, 130, 33, 9, 1, True), (1, 300, 257, 64, 20, 1, True), (2, 72, 196, 49, 9, 2, False)]


@pytest.mark.parametrize("B,C,N,M,k,d,mode", KNN_CASES)
def test_knn_bit_exact_vs_c_oracle(dev, B, C, N, M, k, d, mode):
    """k-NN indices must equal the C oracle exactly (same pinned arithmetic order), incl. the all-ties case."""
    from graphecho_amd import functional as GF
    from oracle.knn import knn_graph as knn_ref

    gen = torch.Generator().manual_seed(12)
    x = torch.randn(B, C, N, 1, generator=gen)
    y = None if M is None else torch.randn(B, C, M, 1, generator=gen)
    if mode == "zeros":  # TGCN step 0: hidden state is all zeros -> every distance ties
        y = torch.zeros(B, C, M, 1)
    rp = torch.randn(1, N, M if M else N, generator=gen) * 0.1 if mode is False else None
    ref = knn_ref(x.numpy(), None if y is None else y.numpy(), k, d, None if rp is None else rp.numpy(), True)
    out = GF.knn_graph(x.to(dev), None if y is None else y.to(dev), k, d, None if rp is None else rp.to(dev), True)
    assert out.dtype == torch.int64 and tuple(out.shape) == ref.shape
    assert np.array_equal(out.cpu().numpy(), ref), f"mismatching entries: {(out.cpu().numpy() != ref).sum()}"


@pytest.mark.parametrize("B,C,N,M,k,normalize", [(2, 256, 4096, 256, 9, True), (3, 256, 1024, 256, 9, True), (2, 96, 520, 130, 18, True),
                                                 (2, 64, 300, 70, 9, False)])
def test_knn_fused_normalisation_equals_two_pass_form(dev, B, C, N, M, k, normalize):
    """ge_knn_topk_fused (the kernel normalises its own query rows: fmaf chains ascending in c, IEEE division per element) against
    ge_knn_prepare(x) + ge_knn_topk at the step's sizes: every index identical, and both equal to the C oracle on a row sample."""
    from graphecho_amd import functional as GF
    from oracle.knn import knn_graph as knn_ref

    gen = torch.Generator().manual_seed(N + M)
    x = torch.randn(B, C, N, 1, generator=gen) * (1 + 3 * torch.rand(B, 1, N, 1, generator=gen))      # rows of very different norms
    y = torch.randn(B, C, M, 1, generator=gen)
    out = {}
    for fused in (True, False):
        prev, GF.KNN_FUSED = GF.KNN_FUSED, fused
        try:
            out[fused] = GF.knn_graph(x.to(dev), y.to(dev), k, 1, None, normalize).cpu()
        finally:
            GF.KNN_FUSED = prev
    assert torch.equal(out[True], out[False]), f"{(out[True] != out[False]).sum().item()} entries differ"
    rows = slice(0, min(N, 192))
    ref = knn_ref(x[:1, :, rows].numpy(), y[:1].numpy(), k, 1, None, normalize)
    assert np.array_equal(out[True][:, :1, rows].numpy(), ref)


def test_mr_aggregate(dev):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(13)
    B, C, N, M, K = 2, 24, 100, 30, 9
    x = torch.randn(B, C, N, 1, generator=gen)
    y = torch.randn(B, C, M, 1, generator=gen)
    idx = torch.randint(0, M, (B, N, K), generator=gen)
    ctr = torch.arange(N).view(1, N, 1).expand(B, N, K)
    edge = torch.stack([idx, ctr]).contiguous()
    gout = torch.randn(B, 2 * C, N, 1, generator=gen)

    def ref_fn(x, y):
        bi = torch.arange(B).view(B, 1, 1, 1)
        xj = y[:, :, :, 0][bi, torch.arange(C).view(1, C, 1, 1), idx.unsqueeze(1)]      # B,C,N,K
        xi = x[:, :, :, 0][bi, torch.arange(C).view(1, C, 1, 1), ctr.unsqueeze(1)]
        m = (xj - xi).max(-1, keepdim=True)[0]
        return torch.cat([x.unsqueeze(2), m.unsqueeze(2)], dim=2).reshape(B, 2 * C, N, 1)

    ref, rg = grads(ref_fn, [x, y], gout)
    out, gg = grads(lambda x, y: GF.mr_aggregate(x, edge.to(dev), y), [x.to(dev), y.to(dev)], gout)
    close(out, ref, 1e-6, what="mr fwd")
    close(gg[0], rg[0], 1e-5, what="mr dx")
    close(gg[1], rg[1], 1e-5, what="mr dy")
    # self-graph (y is None)
    idx2 = torch.randint(0, N, (B, N, K), generator=gen)
    edge2 = torch.stack([idx2, ctr]).contiguous()

    def ref_self(x):
        bi = torch.arange(B).view(B, 1, 1, 1)
        ci = torch.arange(C).view(1, C, 1, 1)
        xs = x[:, :, :, 0]
        m = (xs[bi, ci, idx2.unsqueeze(1)] - xs[bi, ci, ctr.unsqueeze(1)]).max(-1, keepdim=True)[0]
        return torch.cat([x.unsqueeze(2), m.unsqueeze(2)], dim=2).reshape(B, 2 * C, N, 1)

    ref, rg = grads(ref_self, [x], gout)
    out, gg = grads(lambda x: GF.mr_aggregate(x, edge2.to(dev)), [x.to(dev)], gout)
    close(out, ref, 1e-6, what="mr self fwd")
    close(gg[0], rg[0], 1e-5, what="mr self dx")


@pytest.mark.parametrize("B,C,N,M,K", [(2, 24, 100, 30, 9), (3, 70, 300, 77, 9), (2, 64, 1100, 256, 9),
                                       (2, 40, 130, None, 9), (2, 40, 130, 50, 5), (1, 256, 64, None, 12),
                                       (2, 12, 700, None, 9), (1, 20, 1300, 300, 9), (2, 8, 513, 512, 27)])
def test_mr_aggregate_tiled(dev, B, C, N, M, K):
    """LDS-tiled kernels (graphs tagged centre-is-self, as knn_graph returns them): ragged channel / node tiles,
    several node splits in the backward, y given and the self graph.  Forward values and arg-max routing exact."""
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(131)
    Mm = N if M is None else M
    x = torch.randn(B, C, N, 1, generator=gen)
    y = None if M is None else torch.randn(B, C, M, 1, generator=gen)
    idx = torch.randint(0, Mm, (B, N, K), generator=gen)
    ctr = torch.arange(N).view(1, N, 1).expand(B, N, K)
    edge = torch.stack([idx, ctr]).contiguous()
    gout = torch.randn(B, 2 * C, N, 1, generator=gen)

    def ref_fn(x, y=None):
        bi = torch.arange(B).view(B, 1, 1, 1)
        ci = torch.arange(C).view(1, C, 1, 1)
        src = (x if y is None else y)[:, :, :, 0]
        m = (src[bi, ci, idx.unsqueeze(1)] - x[:, :, :, 0].unsqueeze(-1)).max(-1, keepdim=True)[0]
        return torch.cat([x.unsqueeze(2), m.unsqueeze(2)], dim=2).reshape(B, 2 * C, N, 1)

    ins = [x] if y is None else [x, y]
    ref, rg = grads(ref_fn, ins, gout)
    e_dev = edge.to(dev)
    e_dev._ge_centre_is_self = True
    out, gg = grads(lambda *a: GF.mr_aggregate(a[0], e_dev, a[1] if len(a) > 1 else None), [t.to(dev) for t in ins], gout)
    assert torch.equal(out.cpu(), ref), "tiled mr forward must be exact"
    close(gg[0], rg[0], 1e-5, what="tiled mr dx")
    if y is not None:
        close(gg[1], rg[1], 1e-5, what="tiled mr dy")
    # the general kernels (untagged edge_index) give the same forward bits
    out2 = GF.mr_aggregate(x.to(dev), edge.to(dev), None if y is None else y.to(dev))
    assert torch.equal(out2, out.detach())
    # the deterministic gather (default) and the LDS-atomic scatter of rounds 1-3 sum the same terms
    assert GF.MR_BWD_DETERMINISTIC
    GF.MR_BWD_DETERMINISTIC = False
    try:
        _, gs = grads(lambda *a: GF.mr_aggregate(a[0], e_dev, a[1] if len(a) > 1 else None), [t.to(dev) for t in ins], gout)
    finally:
        GF.MR_BWD_DETERMINISTIC = True
    for u, v in zip(gg, gs):
        close(u, v, 1e-5, what="gather vs scatter backward")


@pytest.mark.parametrize("B,C,N,M,K", [(2, 24, 64, None, 9), (3, 40, 100, 30, 9), (2, 16, 128, None, 12), (32, 256, 64, 64, 9)])
def test_mr_backward_one_launch_form_equals_build_plus_gather(dev, B, C, N, M, K):
    """ge_mrconv_gather_bwd_small (list inverted per workgroup in LDS, one launch: graphs of <= 128 nodes by default) and
    ge_mr_inv_build + ge_mrconv_gather_bwd_det walk the same list in the same order: identical bits, also for the sizes
    the dispatcher would not give to the one-launch form."""
    from graphecho_amd._lib import lib, check

    gen = torch.Generator().manual_seed(5)
    Mm = N if M is None else M
    idx = torch.randint(0, Mm, (B, N, K), generator=gen)
    edge = torch.stack([idx, torch.arange(N).view(1, N, 1).expand(B, N, K)]).contiguous().to(dev)
    dout = torch.randn(B, 2 * C, N, generator=gen).to(dev)
    argk = torch.randint(0, K, (B, C, N), generator=gen).to(torch.uint8).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    outs = []
    for form in ("small", "det"):
        dx = torch.full((B, C, N), float("nan"), device=dev)
        dy = dx if M is None else torch.full((B, C, Mm), float("nan"), device=dev)
        if form == "small":
            if not lib.ge_mrconv_gather_bwd_small_ok(N, Mm, K, 1):
                pytest.skip("one-launch form not offered for this size (GE_MR_SMALL, default 128 nodes)")
            check(lib.ge_mrconv_gather_bwd_small(p(dout), p(edge), p(argk), p(dx), p(dy), B, C, N, Mm, K, st), "small")
        else:
            J = lib.ge_mr_inv_chunk()
            inv = torch.empty((B, N * K), device=dev, dtype=torch.int32)
            off = torch.empty((B, -(-N // J), Mm + 1), device=dev, dtype=torch.int32)
            check(lib.ge_mr_inv_build(p(edge), p(inv), p(off), B, N, Mm, K, st), "build")
            ws_n = lib.ge_mrconv_gather_bwd_det_workspace(B, C, N, Mm, K, int(M is None))
            ws = torch.empty(max(ws_n, 1), device=dev)
            check(lib.ge_mrconv_gather_bwd_det(p(dout), p(inv), p(off), p(argk), p(dx), p(dy), p(ws), B, C, N, Mm, K, st), "det")
        outs.append((dx.clone(), dy.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()


@pytest.mark.parametrize("B,P1,P2,D", [(4, 64, 64, 256), (1, 64, 50, 32), (2, 20, 33, 16)])
def test_sinkhorn_distance(dev, B, P1, P2, D):
    """Transport plan / cost / gradients within 1e-3 rel of the oracle (BASELINE.json parity bar)."""
    from graphecho_amd import functional as GF
    from oracle.misc import sinkhorn_distance as ref_sd

    gen = torch.Generator().manual_seed(14)
    x = torch.rand(B, P1, D, generator=gen)
    y = torch.rand(B, P2, D, generator=gen)
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    cost, pi, C, nits = ref_sd(xr, yr, 0.1, 5, "none")
    gpi = torch.randn(*pi.shape, generator=gen) * 0.01
    (cost.sum() + (pi * gpi).sum()).backward()
    xg, yg = x.to(dev).requires_grad_(True), y.to(dev).requires_grad_(True)
    c2, p2, C2, n2 = GF.sinkhorn_distance(xg, yg, 0.1, 5, 0.1)
    (c2.sum() + (p2 * gpi.to(dev)).sum()).backward()
    assert int(n2.item()) == nits
    close(C2, C.reshape(C2.shape), 1e-5, what="cost matrix")
    close(p2, pi.reshape(p2.shape), 1e-3, what="plan")
    close(c2, cost.reshape(c2.shape), 1e-3, what="cost")
    close(xg.grad, xr.grad, 2e-3, what="dx")
    close(yg.grad, yr.grad, 2e-3, what="dy")


# (7, 5): the launch chain; the others: the co-operative one-launch kernels (16 workgroups, rows split), both register tilings
# (<= 20 / <= 40 rows per workgroup), every column-block count, the size limits
@pytest.mark.parametrize("N1,N2", [(7, 5), (130, 97), (278, 376), (16, 64), (333, 512), (640, 100), (401, 203), (290, 412)])
def test_sinkhorn_rpm(dev, N1, N2):
    from graphecho_amd import functional as GF
    from oracle.misc import sinkhorn_rpm as ref_rpm

    gen = torch.Generator().manual_seed(15)
    a = torch.randn(1, N1, N2, generator=gen)
    gout = torch.randn(1, N1, N2, generator=gen)
    ref, rg = grads(lambda a: ref_rpm(a, 20).exp(), [a], gout)
    out, gg = grads(lambda a: GF.sinkhorn_rpm(a, 20).exp(), [a.to(dev)], gout)
    close(out, ref, 1e-3, what="rpm plan")
    close(gg[0], rg[0], 2e-3, what="rpm grad")


@pytest.mark.parametrize("N1,N2", [(7, 5), (70, 93)])
def test_affinity_mlp(dev, N1, N2):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(16)
    H = 512
    P, Q = torch.randn(N1, H, generator=gen), torch.randn(N2, H, generator=gen)
    b1, w2, b2 = torch.randn(H, generator=gen), torch.randn(1, H, generator=gen) / 20, torch.randn(1, generator=gen)
    gout = torch.randn(N1, N2, generator=gen)
    f = lambda P, Q, b1, w2, b2: (F.relu(P[:, None, :] + Q[None, :, :] + b1) @ w2.t()).squeeze(-1) + b2
    ref, rg = grads(f, [P, Q, b1, w2, b2], gout)
    out, gg = grads(GF.affinity_mlp, [t.to(dev) for t in (P, Q, b1, w2, b2)], gout)
    close(out, ref, what="affinity fwd")
    for a, b_, n in zip(gg, rg, ("dP", "dQ", "db1", "dw2", "db2")):
        close(a, b_, what="affinity " + n)


@pytest.mark.parametrize("B,C,N,M,K", [(2, 5, 7, 4, 3), (3, 32, 300, 17, 9), (1, 64, 1024, 1024, 9), (2, 8, 33, 4000, 2)])
def test_batched_index_select_matches_torch_gather(dev, B, C, N, M, K):
    """models.vig.batched_index_select (vig.py:209-229): forward gather and backward scatter-add with repeated ids."""
    from graphecho_amd.models.vig import batched_index_select

    gen = torch.Generator().manual_seed(B * 100 + N)
    src = torch.randn(B, C, M, 1, generator=gen)
    idx = torch.randint(0, M, (B, N, K), generator=gen)
    gout = torch.randn(B, C, N, K, generator=gen)

    def ref_fn(x):
        flat = x.reshape(B, C, M)
        return torch.gather(flat, 2, idx.reshape(B, 1, N * K).expand(B, C, N * K)).reshape(B, C, N, K)

    ref, rg = grads(ref_fn, [src], gout)
    out, gg = grads(lambda x: batched_index_select(x, idx.to(dev)), [src.to(dev)], gout)
    assert torch.equal(out.cpu(), ref)
    close(gg[0], rg[0], 1e-5, what="batched_index_select d src")
    with pytest.raises(RuntimeError):
        batched_index_select(src.to(dev), idx.to(dev).int())


def test_extra_activations_and_neighbour_reductions(dev):
    """act_layer's leakyrelu / hswish (vig.py:433-450) and the max / sum over the neighbour dimension that EdgeConv2d,
    GraphSAGE and GINConv2d apply (vig.py:122,136,157), forward and backward against torch."""
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(23)
    x = torch.randn(3, 7, 50, 9, generator=gen) * 3
    x[0, 0, 0, :3] = torch.tensor([-3.0, 3.0, 0.0])            # hardswish's breakpoints
    g = torch.randn(3, 7, 50, 9, generator=gen)
    for mine, ref in ((lambda t: GF.leaky_relu(t, 0.2), lambda t: F.leaky_relu(t, 0.2)),
                      (GF.hardswish, F.hardswish), (GF.gelu, F.gelu), (GF.relu, F.relu)):
        want, rg = grads(ref, [x], g)
        got, gg = grads(mine, [x.to(dev)], g)
        close(got, want, 1e-6, what="activation")
        close(gg[0], rg[0], 1e-6, what="activation grad")
    a = torch.tensor([0.2])
    want, rg = grads(lambda t, w: F.prelu(t, w), [x, a], g)
    got, gg = grads(GF.prelu, [x.to(dev), a.to(dev)], g)
    close(got, want, 1e-6, what="prelu")
    close(gg[0], rg[0], 1e-6, what="prelu d x")
    close(gg[1], rg[1], 1e-4, what="prelu d slope")
    x[1, 2, 3, :] = 0.5                                           # a row of ties: the first neighbour wins
    g1 = torch.randn(3, 7, 50, 1, generator=gen)
    want, rg = grads(lambda t: t.max(-1, keepdim=True)[0], [x], g1)
    got, gg = grads(GF.neighbour_max, [x.to(dev)], g1)
    assert torch.equal(got.cpu(), want)
    tie = gg[0][1, 2, 3].cpu()
    assert tie[0] == g1[1, 2, 3, 0] and not tie[1:].any()
    mask = torch.ones_like(x, dtype=torch.bool)
    mask[1, 2, 3] = False
    assert torch.equal(gg[0].cpu()[mask], rg[0][mask])
    want, rg = grads(lambda t: t.sum(-1, keepdim=True), [x], g1)
    got, gg = grads(GF.neighbour_sum, [x.to(dev)], g1)
    close(got, want, 1e-6, what="neighbour sum")
    assert torch.equal(gg[0].cpu(), rg[0])


def test_batched_weight_packing_equals_per_layer_packing(dev):
    """optim.WeightPacker's one-launch packing (LDS-tiled transposes) == ge_conv2d_pack_weight layer by layer, both
    layouts: grouped, 1x1 / 3x3 / 7x7, channel counts that are not multiples of the 64-wide tile, and a layer whose
    per-channel block (Ci_g*kh*kw > 4160) takes the unstaged path."""
    from graphecho_amd import functional as GF, nn as gnn
    from graphecho_amd.optim import FlatAdam

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = gnn.Conv2d(3, 48, 7, padding=3)
            self.b = gnn.Conv2d(48, 72, 3, padding=1, groups=4)
            self.c = gnn.Conv2d(72, 130, 1)
            self.d = gnn.Conv2d(130, 64, 3, padding=1, bias=False)
            self.e = gnn.Conv2d(520, 8, 3, padding=1)          # 520*9 = 4680 floats per output channel
            self.f = gnn.Conv2d(8, 8, 1, groups=2)

    torch.manual_seed(3)
    net = Net().to(dev)
    opt = FlatAdam(net, lr=1e-3)
    seen = 0
    for m in net.modules():
        if not isinstance(m, gnn.Conv2d):
            continue
        for tr in (False, True):
            want = GF._pack_weight(m.weight, m.groups, tr)
            got = m._pack.get(m.weight, m.groups, tr)
            if tr and m.groups == 1 and m.weight.shape[2] == 1:
                assert got is m.weight
                continue
            assert m._pack.static_key is not None and got.data_ptr() != want.data_ptr()
            assert torch.equal(got.reshape(-1), want.reshape(-1)), (tuple(m.weight.shape), m.groups, tr)
            seen += 1
    assert seen == 11
    for p in net.parameters():
        p.grad.normal_()
    opt.step()                                   # repacks after the update
    for m in net.modules():
        if isinstance(m, gnn.Conv2d):
            assert torch.equal(m._pack.get(m.weight, m.groups, False).reshape(-1),
                               GF._pack_weight(m.weight, m.groups, False).reshape(-1))


def test_losses(dev):
    from graphecho_amd import functional as GF
    from oracle.misc import dice_loss as ref_dice

    gen = torch.Generator().manual_seed(17)
    x = torch.randn(3, 4, 32, 32, generator=gen) * 2
    t = (torch.rand(3, 4, 32, 32, generator=gen) > 0.7).float()
    g = torch.tensor(0.7)
    ref, rg = grads(lambda x: F.binary_cross_entropy_with_logits(x, t), [x], g)
    out, gg = grads(lambda x: GF.bce_with_logits(x, t.to(dev)), [x.to(dev)], g)
    close(out, ref, 1e-5, what="bce")
    close(gg[0], rg[0], 1e-5, what="bce grad")
    ref, rg = grads(lambda x: F.binary_cross_entropy_with_logits(x, torch.ones_like(x)), [x], g)
    out, gg = grads(lambda x: GF.bce_with_logits(x, 1.0), [x.to(dev)], g)
    close(out, ref, 1e-5, what="bce const")
    close(gg[0], rg[0], 1e-5, what="bce const grad")
    ref, rg = grads(lambda x: ref_dice(x, t), [x], g)
    out, gg = grads(lambda x: GF.dice_loss(x, t.to(dev)), [x.to(dev)], g)
    close(out, ref, 1e-5, what="dice")
    close(gg[0], rg[0], 1e-4, what="dice grad")
    # class counts on both sides of the register-resident kernel's limit (C <= 8), several partial blocks per sample
    for (B, C, hw) in [(2, 1, 16), (2, 8, 72), (2, 11, 40), (1, 3, 130)]:
        x = torch.randn(B, C, hw, hw, generator=gen) * 2
        t = (torch.rand(B, C, hw, hw, generator=gen) > 0.7).float()
        ref, rg = grads(lambda x: ref_dice(x, t), [x], g)
        out, gg = grads(lambda x: GF.dice_loss(x, t.to(dev)), [x.to(dev)], g)
        close(out, ref, 1e-5, what=f"dice C={C}")
        close(gg[0], rg[0], 1e-4, what=f"dice grad C={C}")


def test_optimizers_match_torch(dev):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(18)
    p0 = torch.randn(1000, generator=gen)
    gs = [torch.randn(1000, generator=gen) for _ in range(3)]
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=3e-4, weight_decay=1e-4)
    pg, m, v = p0.to(dev).clone(), torch.zeros(1000, device=dev), torch.zeros(1000, device=dev)
    for i, g in enumerate(gs):
        p.grad = g.clone()
        opt.step()
        GF.adam_step_(pg, g.to(dev), m, v, 3e-4, 0.9, 0.999, 1e-8, 1e-4, i + 1)
    close(pg, p.data, 1e-6, what="adam")
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pg, buf = p0.to(dev).clone(), torch.zeros(1000, device=dev)
    for i, g in enumerate(gs):
        p.grad = g.clone()
        opt.step()
        GF.sgd_step_(pg, g.to(dev), buf, 0.01, 0.9, 1e-4, i == 0)
    close(pg, p.data, 1e-6, what="sgd")


@pytest.mark.parametrize("H,W,S,crop,T,To", [(300, 420, 328, 256, 1, 1), (150, 131, 124, 112, 1, 1), (96, 80, 72, 64, 5, 8),
                                             (64, 64, 64, 64, 3, 3)])
def test_input_formatting_bit_exact(dev, H, W, S, crop, T, To):
    """ge_frames_prepare / ge_labels_onehot against the CPU oracle: byte/index work, must be bit-exact (uint8 and
    float32 sources, explicit / centre / origin crops, clip fold and time resize)."""
    from graphecho_amd import data as gd
    from oracle import data as od

    rng = np.random.default_rng(4)
    N, C = 3, 2
    img = rng.integers(0, 256, (N, C, H, W, T), dtype=np.uint8)
    lab = rng.integers(0, 5, (N, H, W, T), dtype=np.uint8)
    if T == 1:
        img, lab = img[..., 0], lab[..., 0]
    offs = [(0, 0), (S - crop, S - crop), ((S - crop) // 2, (S - crop) // 3)]
    clip = To if T > 1 else None
    for kw in (dict(offsets=offs), dict(center=True), dict()):
        ref = od.prepare_frames(img, S, crop, clip_length=clip, **kw)
        out = gd.prepare_frames(torch.from_numpy(img).to(dev), S, crop, clip_length=clip, **kw)
        assert np.array_equal(out.cpu().numpy(), ref)
        outf = gd.prepare_frames(torch.from_numpy(img.astype(np.float32)).to(dev), S, crop, clip_length=clip, **kw)
        assert np.array_equal(outf.cpu().numpy(), ref)
        refl = od.onehot_labels(lab, (0, 1, 2, 4), S, crop, clip_length=clip, **kw)
        outl = gd.onehot_labels(torch.from_numpy(lab).to(dev), (0, 1, 2, 4), S, crop, clip_length=clip, **kw)
        assert np.array_equal(outl.cpu().numpy(), refl)
    with pytest.raises(ValueError):
        gd.prepare_frames(torch.from_numpy(img).to(dev), S, S + 1)
    with pytest.raises(RuntimeError):
        gd.prepare_frames(torch.from_numpy(img), S, crop)


def test_overlap_meter_matches_reference_formulas(dev):
    from graphecho_amd.data import OverlapMeter
    from oracle.data import overlap_metrics

    gen = torch.Generator().manual_seed(21)
    logits = torch.randn(5, 3, 70, 50, generator=gen)
    logits[0, 0, :3] = 0.0    # sigmoid(0) = 0.5 is NOT > 0.5
    masks = (torch.rand(5, 3, 70, 50, generator=gen) > 0.6).float()
    meter = OverlapMeter(3, dev)
    meter.update(logits[:2].to(dev), masks[:2].to(dev))
    meter.update(logits[2:].to(dev), masks[2:].to(dev))
    ref = overlap_metrics(logits.numpy(), masks.numpy())
    got = meter.metrics()
    for i, key in enumerate(("pixel_acc", "dice", "precision", "specificity", "recall")):
        assert np.allclose(got[key].numpy(), ref[:, i], rtol=0, atol=1e-12), key
    assert int(meter.counts.sum()) == logits.numel()


@pytest.mark.parametrize("shape", [(3, 5, 7, 11), (2, 256, 64, 64), (1, 1, 1, 3)])
def test_mean_square(dev, shape):
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(31)
    x = torch.randn(*shape, generator=gen) * 1.7
    ref, rg = grads(lambda x: 0.01 * (x * x).mean(), [x.double()], torch.ones((), dtype=torch.float64))
    out, gg = grads(lambda x: 0.01 * GF.mean_square(x), [x.to(dev)], torch.ones(()))
    close(out, ref.float(), 1e-5, what="mean_square fwd")
    close(gg[0], rg[0].float(), 1e-5, what="mean_square bwd")


F16_CONV_CASES = [
    # B, Cin, H, W, Cout, k, s, p, groups, bias
    (2, 64, 16, 16, 64, 3, 1, 1, 1, False),
    (2, 256, 16, 16, 128, 3, 1, 1, 1, True),
    (3, 64, 9, 7, 256, 1, 1, 0, 1, False),      # ragged N tile
    (2, 128, 16, 16, 128, 3, 2, 1, 1, False),   # stride 2 (generic strided data-gradient)
    (2, 256, 16, 16, 512, 1, 2, 0, 1, True),
    (2, 512, 9, 7, 256, 1, 1, 0, 4, True),      # grouped 1x1: 128 -> 64 per group
    (4, 256, 32, 32, 256, 3, 1, 1, 1, True),    # 128 x 128 tiles
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,groups,bias", F16_CONV_CASES)
def test_conv2d_f16_mfma_path(dev, B, Cin, H, W, Cout, k, s, p, groups, bias):
    """fp16-input MFMA conv (config 5's conv path): forward and data gradient against the exact reference evaluated on
    fp16-rounded operands (what the kernel computes: rounding error of the inputs only, fp32 accumulation), and
    against the unrounded fp32 result within fp16 resolution; fused BN statistics and the skip addend included."""
    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(77)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin // groups, k, k, generator=gen) / (Cin // groups * k * k) ** 0.5
    b = torch.randn(Cout, generator=gen) if bias else None
    xh, wh = x.half().float(), w.half().float()
    ref = F.conv2d(xh.double(), wh.double(), None if b is None else b.double(), s, p, 1, groups)
    gout = torch.randn(ref.shape, generator=gen)
    ref_dx = torch.nn.grad.conv2d_input(x.shape, wh.double(), gout.half().float().double(), s, p, 1, groups)
    exact = F.conv2d(x.double(), w.double(), None if b is None else b.double(), s, p, 1, groups)
    assert GF.CONV_PRECISION == "f32"
    GF.CONV_PRECISION = "f16"
    try:
        xg = x.to(dev).requires_grad_(True)
        wg = w.to(dev).requires_grad_(True)
        bg = None if b is None else b.to(dev).requires_grad_(True)
        out = GF.conv2d(xg, wg, bg, s, p, groups, GF.PackCache(), True)
        y, stats = out
        y.backward(gout.to(dev))
    finally:
        GF.CONV_PRECISION = "f32"
    scale = ref.abs().max().item()
    assert (y.detach().cpu().double() - ref).abs().max().item() <= 2e-5 * scale, "fp16-operand forward is not exact"
    assert (y.detach().cpu().double() - exact).abs().max().item() <= 4e-3 * scale
    dscale = ref_dx.abs().max().item()
    assert (xg.grad.cpu().double() - ref_dx).abs().max().item() <= 2e-5 * dscale, "fp16-operand dgrad is not exact"
    ref_dw = torch.nn.grad.conv2d_weight(xh.double(), w.shape, gout.half().float().double(), s, p, 1, groups)
    assert (wg.grad.cpu().double() - ref_dw).abs().max().item() <= 2e-5 * ref_dw.abs().max().item(), \
        "fp16-operand wgrad is not exact"
    # fused BatchNorm moments of the fp16-path output
    n = stats[..., 0].sum(1)
    mean = (stats[..., 0] * stats[..., 1]).sum(1) / n
    yd = y.detach()
    assert torch.allclose(n.cpu(), torch.full((Cout,), float(yd.numel() // Cout)))
    assert torch.allclose(mean.cpu(), yd.mean((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,P1,P2,D", [(1, 64, 64, 256), (4, 64, 64, 256), (64, 64, 64, 256), (3, 64, 50, 96), (2, 80, 70, 33),
                                       (5, 7, 130, 64)])
def test_sinkhorn_distance_one_launch_form(dev, B, P1, P2, D):
    """sd_fused_kernel (cost tile, iterations, stopping rule across the batch, plan and cost in ONE launch; the
    workgroups meet at a device counter) against the cost / iterate / finalize launches and the oracle: same stopping
    iteration, cost matrix within 2e-5, cost / plan within 2e-4 of the three-launch form, gradients 1e-3; repeated calls (the
    meeting point must be left clean) and an early-stopping case (identical point sets: error 0 after one iteration)."""
    from graphecho_amd import functional as GF
    from oracle.misc import sinkhorn_distance as ref_sd

    gen = torch.Generator().manual_seed(B * 1000 + P1 + P2)
    x = torch.rand(B, P1, D, generator=gen)
    y = torch.rand(B, P2, D, generator=gen)
    assert GF.lib.ge_sinkhorn_distance_fused_ok(B, P1, P2) == 1
    res = {}
    for fused in (True, False):
        GF.SD_FUSED = fused
        try:
            for rep in range(3 if fused else 1):
                xg, yg = x.to(dev).requires_grad_(True), y.to(dev).requires_grad_(True)
                cost, pi, C, nits = GF.sinkhorn_distance(xg, yg, 0.1, 5)
                (cost.sum() + (pi * C).sum()).backward()
                res[(fused, rep)] = (cost.detach(), pi.detach(), C.detach(), int(nits.item()), xg.grad, yg.grad)
        finally:
            GF.SD_FUSED = True
    a, b = res[(True, 2)], res[(False, 0)]
    assert a[3] == b[3] == res[(True, 0)][3]
    for u, v, what in zip(a[:3] + a[4:], b[:3] + b[4:], ("cost", "plan", "cost matrix", "dx", "dy")):
        # the two forms add the cost's 256 squares in different orders: last-bit differences of C, which the plan
        # exp((-C + u + v) / 0.1) and the gradients amplify tenfold and more
        close(u, v, {"cost matrix": 2e-5, "cost": 2e-4, "plan": 2e-4}.get(what, 1e-3), what=what)
    for u, v in zip(a[:3], res[(True, 0)][:3]):          # repeated launches: the meeting point was left clean
        assert torch.equal(u, v)
    rc, rp, rC = ref_sd(x, y, 0.1, 5)[:3]
    close(a[0], rc, 1e-3, what="cost vs oracle")
    close(a[1], rp, 1e-3, what="plan vs oracle")
    # identical point sets with a uniform cost structure stop early: every rank must agree on the iteration
    if P1 == P2:
        xe = x[:, :1].expand(B, P1, D).contiguous().to(dev)
        c1, p1, _, n1 = GF.sinkhorn_distance(xe, xe.clone(), 0.1, 5)
        GF.SD_FUSED = False
        try:
            c2, p2, _, n2 = GF.sinkhorn_distance(xe, xe.clone(), 0.1, 5)
        finally:
            GF.SD_FUSED = True
        assert int(n1.item()) == int(n2.item()) < 5
        close(p1, p2, 1e-5, what="early-stop plan")


def test_run_to_run_reproducibility(dev):
    """What is bit-reproducible and what is not.  Conv (fwd / dgrad / split-K wgrad with its ordered
    slab reduce), BatchNorm, GroupNorm (ordered per-wave partials), bilinear resize, pooling: identical bits on every run.
    Round 4: the max-relative BACKWARD on k-NN graphs too (inverse neighbour lists + gather, no float atomics).  What
    is left outside: arbitrary user edge_index tensors (general mr_bwd_kernel / edge_gather_bwd_kernel: LDS float atomics,
    off the trainers' path)."""
    import torch.nn.functional as F

    from graphecho_amd import functional as GF

    gen = torch.Generator().manual_seed(91)
    x = torch.randn(4, 64, 32, 32, generator=gen).to(dev)
    w = torch.randn(64, 64, 3, 3, generator=gen).to(dev)
    gam, bet = torch.rand(64, generator=gen).to(dev), torch.randn(64, generator=gen).to(dev)
    gout = torch.randn(4, 64, 32, 32, generator=gen).to(dev)

    def conv_gn():
        xi, wi, g, b = (t.clone().requires_grad_(True) for t in (x, w, gam, bet))
        y = GF.group_norm(GF.conv2d(xi, wi, None, 1, 1), 32, g, b, 1e-5, True)
        y.backward(gout)
        return [y.detach(), xi.grad, wi.grad, g.grad, b.grad]

    a, b = conv_gn(), conv_gn()
    for u, v in zip(a, b):
        assert torch.equal(u, v), "conv / GroupNorm path must be bit-reproducible"
    # max-relative aggregation on a k-NN graph: forward exact, backward within the stated tolerance
    feat = torch.randn(2, 64, 1024, 1, generator=gen).to(dev)
    cand = F.avg_pool2d(feat.reshape(2, 64, 32, 32), 2, 2).reshape(2, 64, 256, 1).contiguous()
    edge = GF.knn_graph(feat, cand, 9, 1, None, normalize=True)

    def mr():
        f, c = feat.clone().requires_grad_(True), cand.clone().requires_grad_(True)
        o = GF.mr_aggregate(f, edge, c)
        o.backward(torch.ones_like(o) * 0.37)
        return o.detach(), f.grad, c.grad

    (o1, gf1, gc1), (o2, gf2, gc2) = mr(), mr()
    assert torch.equal(o1, o2) and torch.equal(gf1, gf2)
    # round 4: the backward on k-NN graphs is a gather over inverse neighbour lists in a fixed order -- exact
    assert GF.MR_BWD_DETERMINISTIC and torch.equal(gc1, gc2), "max-relative backward must be bit-reproducible"
    # ... for random (non-constant) gradients, the self graph (y is x) and a ragged node count too
    for (B, C, N, r) in ((3, 40, 1000, 0), (2, 64, 4096, 4), (2, 24, 300, 0)):
        side = int(N ** 0.5)
        f0 = torch.randn(B, C, N, 1, generator=gen).to(dev)
        c0 = F.avg_pool2d(f0.reshape(B, C, side, side), r, r).reshape(B, C, -1, 1).contiguous() if r else None
        e0 = GF.knn_graph(f0, c0, 9, 1, None, normalize=True)
        g0 = torch.randn(B, 2 * C, N, 1, generator=gen).to(dev)
        runs = []
        for _ in range(3):
            f = f0.clone().requires_grad_(True)
            c = c0.clone().requires_grad_(True) if c0 is not None else None
            GF.mr_aggregate(f, e0, c).backward(g0)
            runs.append((f.grad, None if c is None else c.grad))
        for fg, cg in runs[1:]:
            assert torch.equal(fg, runs[0][0]) and (cg is None or torch.equal(cg, runs[0][1]))


def test_gmodule_front_end_kernels_vs_torch_restatement(dev):
    """ge_mask_boxes / ge_fcos_labels / ge_gather_nodes_* against the torch restatement of the same steps
    (GModule.masks_to_boxes, PrototypeComputation.label_maps / sample), which the CPU suite pins to the oracle: boxes and
    byte labels exactly (integer results), gathered rows and their scattered gradients exactly (pure data movement)."""
    from graphecho_amd import functional as GF
    from graphecho_amd.models.graph_matching import GModule
    from oracle.weights import det_tensor, rect_masks

    gm = GModule(256, 4, dev).to(dev)
    gen = gm.graph_generator
    for seed, B in ((5, 3), (9, 8)):
        masks = rect_masks(B, 4, 256, 256, seed=seed)
        masks[1, 2] = 0                                  # empty class channel -> full-image box
        masks[0, 1, 17, 200] = float("nan")              # NaN counts as non-zero, as in torch
        md = masks.to(dev)
        boxes = GF.mask_boxes(md.reshape(-1, 256, 256)).view(B, 4, 4)
        want = gm.find_bbox(md)
        assert torch.equal(boxes, want)
        feats = [det_tensor(f"fe{seed}.f{l}", (B, 16, s, s)).to(dev) for l, s in enumerate((64, 32, 16, 8))]
        levels = [(f.shape[2], f.shape[3], gm.fpn_strides[l]) for l, f in enumerate(feats)]
        lab = GF.fcos_labels(boxes, levels, gen.SIZES_OF_INTEREST[:4])
        ref = torch.cat([l.reshape(B, -1) for l in gen.label_maps(gm.compute_locations(feats), want)], dim=1)
        assert lab.dtype == torch.uint8 and torch.equal(lab.long(), ref)
        # rows: the host plan against the cumsum / searchsorted sampler
        counts = [[int((l > 0).sum()), int((l == 0).sum())] for l in gen.label_maps(gm.compute_locations(feats), want)]
        fr = [f.clone().requires_grad_(True) for f in feats]
        ref_nodes, ref_lab = gen.sample(fr, gen.label_maps(gm.compute_locations(feats), want), gen.plan(counts))
        level, index, node_lab, unique, present = gen.plan_rows(lab.cpu().numpy(), [h * w for h, w, _ in levels])
        assert unique and np.array_equal(node_lab, ref_lab.cpu().numpy())
        fh = [f.clone().requires_grad_(True) for f in feats]
        nodes = GF.gather_nodes(fh, torch.as_tensor(level).to(dev), torch.as_tensor(index).to(dev), unique, present)
        assert torch.equal(nodes, ref_nodes)
        g = torch.randn(nodes.shape, generator=torch.Generator().manual_seed(seed)).to(dev)
        nodes.backward(g)
        ref_nodes.backward(g)
        for a, b, pr in zip(fh, fr, present):
            if pr:
                assert torch.equal(a.grad, b.grad)
            else:
                assert a.grad is None and (b.grad is None or not b.grad.any())
    # repeated locations: gradients add up (atomic path)
    f = [torch.randn(2, 8, 4, 4, device=dev, requires_grad=True)]
    level = torch.zeros(5, dtype=torch.int64, device=dev)
    index = torch.tensor([3, 3, 17, 3, 31], device=dev)
    rows = GF.gather_nodes(f, level, index, unique=False)
    rows.backward(torch.ones_like(rows))
    want = torch.zeros(2, 8, 16, device=dev)
    want[0, :, 3], want[1, :, 1], want[1, :, 15] = 3.0, 1.0, 1.0
    assert torch.equal(f[0].grad.reshape(2, 8, 16), want)


@pytest.mark.parametrize("shape", [(7, 5), (64, 96), (300, 517), (600, 600)])
def test_match_o2o_loss_vs_torch_expression(dev, shape):
    """ge_match_o2o_* (GModule._forward_aff's one-to-one loss on the log plan, graph_matching.py:577-590) against the reference's
    own chain of torch ops in fp64: loss, M, and the gradient wrt the log plan with a second consumer of M attached."""
    from graphecho_amd import functional as GF

    N1, N2 = shape
    g = torch.Generator().manual_seed(N1 * 1000 + N2)
    # log of a sub-stochastic plan: entries in (0, 1), every row with at least one same-class column
    X0 = (torch.rand(N1, N2, generator=g) * 0.9 + 0.02).log().to(dev)
    lab1 = torch.randint(0, 4, (N1,), generator=g).float().to(dev)
    lab2 = torch.randint(0, 4, (N2,), generator=g).float().to(dev)
    W = torch.randn(N1, N2, generator=g).to(dev)

    def ref(X):
        M = X.exp()
        target = (lab1.long()[:, None] == lab2.long()[None, :]).to(X.dtype)
        indx = (M * target).max(-1)[1]
        tp = M.gather(1, indx[:, None])
        tp_loss = (-0.25 * (1 - tp) ** 2 * torch.log(tp)).mean() / tp.shape[0]
        fp_mask = 1.0 - target
        fp_terms = -0.75 * M ** 2 * torch.log(1 - M) * fp_mask
        fp_loss = fp_terms.sum() / fp_mask.sum() / (M * fp_mask).sum().detach()
        return tp_loss + fp_loss, M

    Xr = X0.double().requires_grad_(True)
    lr, Mr = ref(Xr)
    (lr * 3.0 + (Mr * W.double()).sum() * 1e-3).backward()
    Xh = X0.clone().requires_grad_(True)
    lh, Mh = GF.match_o2o_loss(Xh, lab1, lab2)
    (lh * 3.0 + (Mh * W).sum() * 1e-3).backward()
    assert abs(lh.item() - lr.item()) <= 2e-5 * abs(lr.item()), (lh.item(), lr.item())
    assert (Mh.double() - Mr).abs().max().item() <= 1e-6
    scale = Xr.grad.abs().max().item()
    assert (Xh.grad.double() - Xr.grad).abs().max().item() <= 2e-5 * scale, ((Xh.grad.double() - Xr.grad).abs().max().item(), scale)
    # loss only (no gradient reaches M)
    Xh2 = X0.clone().requires_grad_(True)
    GF.match_o2o_loss(Xh2, lab1, lab2)[0].backward()
    Xr2 = X0.double().requires_grad_(True)
    ref(Xr2)[0].backward()
    assert (Xh2.grad.double() - Xr2.grad).abs().max().item() <= 2e-5 * Xr2.grad.abs().max().item()


def test_seed_bank_update_vs_torch_expression(dev):
    """ge_seed_bank_update (GModule.update_seed's momentum blend, graph_matching.py:532-567) against the reference's torch ops:
    kept-row class means, cosine similarity with the bank rows, blend for the classes present; absent classes untouched."""
    import numpy as np
    import torch.nn.functional as F
    from graphecho_amd import functional as GF

    g = torch.Generator().manual_seed(11)
    nc, N, D = 5, 333, 256
    nodes = torch.randn(N, D, generator=g).to(dev)
    bank0 = torch.randn(nc, D, generator=g).to(dev)
    cls = torch.randint(-1, nc - 1, (N,), generator=g).numpy().astype(np.int32)      # class nc - 1 absent, -1 = dropped rows
    has = np.array([1, 1, 1, 1, 0], dtype=np.int32)
    sel = np.zeros((nc, N), dtype=np.float32)
    for c in range(nc):
        sel[c, cls == c] = 1.0
    cnt = sel.sum(1, keepdim=False)[:, None] if hasattr(sel, "keepdim") else sel.sum(1)[:, None]
    means = (torch.from_numpy(sel).to(dev).double() @ nodes.double()) / torch.from_numpy(cnt).to(dev).double()
    mom = F.cosine_similarity(means, bank0.double(), dim=1).unsqueeze(1)
    new = bank0.double() * mom + means * (1.0 - mom)
    want = torch.where(torch.from_numpy(has.astype(bool))[:, None].to(dev), new, bank0.double())
    bank = bank0.clone()
    tab = torch.from_numpy(np.concatenate([cls, has])).to(dev)
    GF.seed_bank_update(bank, nodes, tab, nc)
    assert torch.equal(bank[4], bank0[4])
    assert (bank.double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()


@pytest.mark.parametrize("shape", [(37, 37, True), (260, 141, False), (600, 555, False)])
def test_mha1_block_vs_fp64_and_composed_form(dev, shape, monkeypatch):
    """ge_mha1_* (the single-head attention block of GModule / TGCN in one call per direction, transformer.py:28-78) against
    (a) the same formula in fp64 torch with the SAME dropout keep masks, every input and parameter gradient, a gradient reaching
    the returned attention included; (b) the composed module (GE_FUSED_MHA=0 path) with dropout off."""
    from graphecho_amd import functional as GF
    from graphecho_amd.models import transformer as T

    def rel(a, b):
        return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()

    Nk, Nq, same = shape
    D = 256
    g = torch.Generator().manual_seed(Nk * 7 + Nq)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    key = rnd(Nk, D)
    value = key if same else rnd(Nk, D)
    query = key if same else rnd(Nq, D)
    Nq = query.shape[0]
    Ws = [rnd(D, D) / 16 for _ in range(4)]
    bs = [rnd(D) * 0.1 for _ in range(4)]
    gamma, beta = 1 + 0.1 * rnd(D), 0.1 * rnd(D)
    m_att = (torch.rand(Nq, Nk, generator=g) < 0.9).float().to(dev)
    m_out = (torch.rand(Nq, D, generator=g) < 0.9).float().to(dev)
    g_out, g_att = rnd(Nq, D), rnd(Nq, Nk) * 0.1
    scale, ms = D ** -0.5, 1.0 / 0.9

    def ref(key, value, query, Ws, bs, gamma, beta):
        k, v, q = key @ Ws[0].T + bs[0], value @ Ws[1].T + bs[1], query @ Ws[2].T + bs[2]
        A = torch.softmax(q @ k.T * scale, -1) * (m_att.double() * ms)
        z = query + ((A @ v) @ Ws[3].T + bs[3]) * (m_out.double() * ms)
        return torch.nn.functional.layer_norm(z, (D,), gamma, beta, 1e-5), A

    leaves64 = [t.double().requires_grad_(True) for t in ([key] if same else [key, value, query]) + Ws + bs + [gamma, beta]]
    kvq = [leaves64[0]] * 3 if same else leaves64[:3]
    rest = leaves64[1:] if same else leaves64[3:]
    o64, a64 = ref(*kvq, rest[:4], rest[4:8], rest[8], rest[9])
    ((o64 * g_out.double()).sum() + (a64 * g_att.double()).sum()).backward()
    leaves = [t.clone().requires_grad_(True) for t in ([key] if same else [key, value, query]) + Ws + bs + [gamma, beta]]
    kvq32 = [leaves[0]] * 3 if same else leaves[:3]
    r32 = leaves[1:] if same else leaves[3:]
    out, att = GF.mha1(*kvq32, r32[0], r32[4], r32[1], r32[5], r32[2], r32[6], r32[3], r32[7], r32[8], r32[9], m_att, m_out,
                       scale, ms)
    ((out * g_out).sum() + (att * g_att).sum()).backward()
    assert rel(out, o64) < 2e-5 and rel(att, a64) < 2e-5, (rel(out, o64), rel(att, a64))
    for a, b in zip(leaves, leaves64):      # (the key bias has a zero gradient -- softmax is shift-invariant: absolute floor)
        err = (a.grad.double() - b.grad).abs().max().item()
        assert err <= 1e-4 * max(b.grad.abs().max().item(), 1e-1), (tuple(a.shape), err, b.grad.abs().max().item())

    # (b) module level, dropout off: fused call == composed ops
    mod = T.MultiHeadAttention(256, 1, dropout=0.0, version="v2").to(dev).train()
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(T, "FUSED_MHA", fused)
        mod.zero_grad()
        ins = [t.clone().requires_grad_(True) for t in (key, value, query)]
        o, a = mod(*ins)
        ((o * g_out).sum() + (a * g_att).sum()).backward()
        res[fused] = (o.detach(), a.detach(), [t.grad for t in ins], [p.grad.clone() for p in mod.parameters()])
    assert rel(res[True][0], res[False][0].double()) < 1e-6 and rel(res[True][1], res[False][1].double()) < 1e-6
    for x, y in zip(res[True][2] + res[True][3], res[False][2] + res[False][3]):
        err = (x.double() - y.double()).abs().max().item()
        assert err <= 2e-5 * max(y.abs().max().item(), 1e-1), (tuple(x.shape), err)


def test_sinkhorn_rpm_cooperative_kernels_repeat_bit_for_bit(dev):
    """The 16 co-operating workgroups of rpm_coop_fwd/bwd_kernel meet at a counter barrier every iteration and exchange partials with
    agent-scope stores: forward + backward 40 times beside copy traffic on another stream, every result equal to the first."""
    from graphecho_amd import functional as GF

    side = torch.cuda.Stream()
    na = torch.randn(16 << 20, device=dev)
    nb = torch.empty_like(na)
    for (N1, N2) in [(270, 320), (401, 203)]:
        torch.manual_seed(N1)
        A = torch.randn(1, N1, N2, device=dev, requires_grad=True)
        W = torch.randn(1, N1, N2, device=dev)

        def run():
            A.grad = None
            X = GF.sinkhorn_rpm(A, 20)
            (X * W).sum().backward()
            return X.detach().clone(), A.grad.clone()

        first = run()
        for _ in range(40):
            with torch.cuda.stream(side):
                nb.copy_(na)
            out = run()
            assert torch.equal(out[0], first[0]) and torch.equal(out[1], first[1])
    torch.cuda.synchronize()


@pytest.mark.timeout(600)
def test_grid_barrier_kernels_concurrent_and_replayed(dev):
    """Forward progress of the kernels whose workgroups meet at a device-scope barrier (rpm_coop_fwd/bwd_kernel, sd_fused_kernel
    with B > 1; launched through ge_launch_coresident = hipLaunchCooperativeKernel, ge_common.h): two of them in flight on two
    streams while a third stream keeps every CU busy with the step's largest convolution, then the batch-16 SinkhornDistance forward
    replayed from a HIP graph beside the same convolution stream.  Every repeat must equal the first result bit for bit; a lost
    workgroup would hang (pytest-timeout) rather than fail."""
    from graphecho_amd import functional as GF

    torch.manual_seed(5)
    s_rpm, s_sd, s_conv = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    A = torch.randn(1, 270, 320, device=dev, requires_grad=True)
    Wt = torch.randn(1, 270, 320, device=dev)
    xs, ys = torch.rand(16, 64, 256, device=dev), torch.rand(16, 64, 256, device=dev)
    cx = torch.randn(32, 256, 64, 64, device=dev)
    cw = torch.randn(256, 256, 3, 3, device=dev) * 0.02
    pack = GF.PackCache()
    assert GF.lib.ge_sinkhorn_rpm_coop_workspace(1, 270, 320) > 0 and GF.lib.ge_sinkhorn_distance_fused_ok(16, 64, 64) == 1
    torch.cuda.synchronize()

    def rpm():
        A.grad = None
        X = GF.sinkhorn_rpm(A, 20)
        (X * Wt).sum().backward()
        return X.detach().clone(), A.grad.clone()

    def sd():
        cost, pi, C, nits = GF.sinkhorn_distance(xs, ys, 0.1, 5)
        return cost.clone(), pi.clone()

    def busy(n):
        with torch.cuda.stream(s_conv), torch.no_grad():
            for _ in range(n):
                GF.conv2d(cx, cw, None, 1, 1, 1, pack)

    with torch.cuda.stream(s_rpm):
        first_rpm = rpm()
    with torch.cuda.stream(s_sd):
        first_sd = sd()
    torch.cuda.synchronize()
    for _ in range(25):
        busy(6)
        with torch.cuda.stream(s_rpm):
            r = rpm()
        with torch.cuda.stream(s_sd):
            d = sd()
        with torch.cuda.stream(s_rpm):
            r2 = rpm()
        torch.cuda.synchronize()
        assert all(torch.equal(u, v) for u, v in zip(r + r2 + d, first_rpm + first_rpm + first_sd))
    # the batch-16 one-launch SinkhornDistance captured into a graph (TGCN's recurrence replays it this way)
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap), torch.no_grad():
        GF.sinkhorn_distance(xs, ys, 0.1, 5)      # the stream's meeting point exists before the capture
        cap.synchronize()
        with torch.cuda.graph(g, stream=cap):
            gc, gp, _gC, _gn = GF.sinkhorn_distance(xs, ys, 0.1, 5)
    for _ in range(20):
        busy(4)
        with torch.cuda.stream(s_rpm):
            r = rpm()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(gc, first_sd[0]) and torch.equal(gp, first_sd[1])
        assert all(torch.equal(u, v) for u, v in zip(r, first_rpm))
