"""fp16 ACTIVATION STORAGE kernels (graphecho_amd/csrc/ge_half.hip, graphecho_amd/half.py) against plain PyTorch fp32 ops.

The kernels compute on fp16-rounded operands with fp32 accumulation and store fp16: references are computed in fp32 FROM THE
SAME fp16-rounded inputs, so what remains is the accumulation order (1e-5-ish) and ONE fp16 rounding of each stored result
(2^-11 relative): tolerance 2e-3 of the output scale for fp16 outputs, 1e-3 for fp32 outputs (weight gradients, sums).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fresh_loss_scale():
    """Every test starts from the initial loss scale (the device-resident scale follows the gradients of the previous backward)."""
    from graphecho_amd import functional as GF

    GF._H_SCALE.clear()
    GF._H_DIRTY.clear()
    yield


def blk(x):
    """fp32 NCHW -> blocked fp16 (torch)."""
    B, C, H, W = x.shape
    return x.view(B, C // 32, 32, H, W).permute(0, 1, 3, 4, 2).contiguous().half()


def unblk(h):
    B, CB, H, W, _ = h.shape
    return h.float().permute(0, 1, 4, 2, 3).reshape(B, CB * 32, H, W).contiguous()


def close(a, b, rtol, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, f"{what}: {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = max(b.abs().max().item(), 1e-12)
    err = (a - b).abs().max().item()
    assert err <= rtol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.3e}, rtol {rtol:g})"


def test_transpose_read_lane_mapping(dev):
    """ds_read_b64_tr_b16 as the weight-gradient kernel uses it: inside each 16-lane group, lane i's element j is element
    i % 4 of the 8-byte chunk that lane 4 j + i / 4 of the group addressed."""
    from graphecho_amd._lib import lib, check

    out = torch.empty(64, 4, device=dev)
    check(lib.ge_h_probe_tr(out.data_ptr(), None), "probe")
    torch.cuda.synchronize()
    exp = torch.empty(64, 4)
    for l in range(64):
        g, i = l >> 4, l & 15
        for j in range(4):
            src_lane = 16 * g + 4 * j + (i >> 2)
            exp[l, j] = src_lane * 4 + (i & 3)
    assert torch.equal(out.cpu(), exp), out.cpu()[:16]


def test_blocked_casts_round_trip(dev):
    from graphecho_amd import half as GH

    x = torch.randn(3, 64, 16, 32, device=dev)
    h = GH.to_blocked(x)
    assert h.shape == (3, 2, 16, 32, 32) and h.dtype == torch.float16
    assert torch.equal(h, blk(x))
    assert torch.equal(GH.from_blocked(h), unblk(h))


CONV_CASES = [
    # B, Cin, Cout, H, W
    (2, 64, 128, 32, 32),      # 128 x 128 tiles, 4 x 32 rectangles
    (2, 64, 64, 64, 64),       # 256-pixel x 64-channel tiles (Cout = 64)
    (1, 128, 256, 16, 16),     # 8 x 16 rectangles
    (1, 64, 128, 16, 128),     # W = 128: column halos come from the neighbouring rectangle
    (3, 64, 192, 32, 64),      # Cout = 192: 64-channel tiles, three of them
    (2, 512, 512, 16, 16),     # config 5's deepest layers
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_forward_backward(dev, case):
    from graphecho_amd import half as GH
    from graphecho_amd._lib import lib

    B, Cin, Cout, H, W = case
    assert GH.supported(B, Cin, Cout, H, W)
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) / (3.0 * Cin ** 0.5)).requires_grad_(True)
    bias = torch.randn(Cout, device=dev).requires_grad_(True)
    h = blk(x).requires_grad_(True)
    z, stats = GH.conv3x3(h, w, bias, None, bn_stats=True)
    # reference on the rounded operands
    xr = unblk(h.detach()).requires_grad_(True)
    wr = w.detach().half().float().requires_grad_(True)
    br = bias.detach().clone().requires_grad_(True)
    zr = F.conv2d(xr, wr, br, padding=1)
    close(unblk(z), zr, 2e-3, "forward")
    # moments of the fp32 results, merged by the BatchNorm finalize kernel
    mean = torch.empty(Cout, device=dev)
    invstd = torch.empty(Cout, device=dev)
    nb = stats.shape[1]
    assert nb == lib.ge_h_conv3x3_stat_parts(B, H, W) == B * H * W // 64
    from graphecho_amd._lib import check

    check(lib.ge_bn_finalize(stats.data_ptr(), nb * 3, 3, nb, Cout, 1e-5, 0.1, None, mean.data_ptr(), invstd.data_ptr(), None,
                             None, None), "finalize")
    close(mean, zr.mean(dim=(0, 2, 3)), 1e-3, "moments: mean")
    close(invstd, torch.rsqrt(zr.var(dim=(0, 2, 3), unbiased=False) + 1e-5), 1e-3, "moments: invstd")
    # backward: gradients carry the loss scale
    from graphecho_amd import functional as GF

    S = GF.h_scale_value(dev)               # the scale the kernels will divide out (device-resident, or the fixed one)
    g = torch.randn_like(zr) * 1e-4
    gh = blk(g * S)
    z.backward(gh)
    gr = unblk(gh) / S          # what the kernels saw, unscaled
    zr.backward(gr)
    close(unblk(h.grad) / S, xr.grad, 2e-3, "data gradient")
    close(w.grad, wr.grad, 1e-3, "weight gradient")
    close(bias.grad, br.grad, 1e-3, "bias gradient")


@pytest.mark.parametrize("segments", [None, (2, 1, 3)])
def test_batch_norm_relu_forward_backward(dev, segments):
    from graphecho_amd import half as GH
    from graphecho_amd import functional as GF

    torch.manual_seed(1)
    B, C, H, W = 6, 64, 32, 32
    Cin = 32
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(C, Cin, 3, 3, device=dev) / (3.0 * Cin ** 0.5)
    gamma = (torch.rand(C, device=dev) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device=dev) * 0.1).requires_grad_(True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    z, stats = GH.conv3x3(blk(x), w, None, None, bn_stats=True)
    z = z.detach().requires_grad_(True)
    a = GH.batch_norm(z, gamma, beta, rm, rv, True, 0.1, 1e-5, True, None, stats, segments)
    # reference: torch batch_norm per segment on the stored fp16 z
    zr = unblk(z.detach()).requires_grad_(True)
    gr_, br_ = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    rmr, rvr = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    outs, b0 = [], 0
    for n in (segments or (B,)):
        outs.append(F.relu(F.batch_norm(zr[b0:b0 + n], rmr, rvr, gr_, br_, True, 0.1, 1e-5)))
        b0 += n
    ar = torch.cat(outs)
    close(unblk(a), ar, 3e-3, "forward")
    close(rm, rmr, 2e-3, "running mean")
    close(rv, rvr, 2e-3, "running var")
    S = GF.h_scale_value(dev)
    g = torch.randn_like(ar) * 1e-5
    gh = blk(g * S)
    a.backward(gh)
    ar.backward(unblk(gh) / S)
    close(unblk(z.grad) / S, zr.grad, 4e-3, "dz")
    close(gamma.grad, gr_.grad, 2e-3, "dgamma")
    close(beta.grad, br_.grad, 2e-3, "dbeta")


def test_max_pool(dev):
    from graphecho_amd import half as GH

    torch.manual_seed(2)
    x = torch.randn(2, 64, 16, 32, device=dev)
    x[0, 0, 0, 0:2] = 1.5       # a tie inside a window: the first maximum takes the gradient
    h = blk(x).requires_grad_(True)
    y = GH.max_pool2(h)
    xr = unblk(h.detach()).requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    assert torch.equal(unblk(y), yr)
    g = torch.randn_like(yr)
    gh = blk(g)
    y.backward(gh)
    yr.backward(unblk(gh))
    assert torch.equal(unblk(h.grad), xr.grad)


@pytest.mark.parametrize("with_skip", [False, True])
def test_generic_conv3x3_through_blocked_kernels(dev, with_skip):
    """functional.conv2d under ACT_STORAGE = "f16": an fp32 NCHW 3x3 / s1 / p1 conv (FPN smoothing conv with its skip alias,
    discriminator tower, ...) runs on the blocked-fp16 kernels -- input and incoming gradient cast once, fp32 NCHW results."""
    from graphecho_amd import functional as GF

    if not GF.H_GENERIC:
        pytest.skip("GE_H_GENERIC=0")
    torch.manual_seed(4)
    B, Cin, Cout, H, W = 3, 256, 128, 32, 32
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (3.0 * Cin ** 0.5)
    bias = torch.randn(Cout, device=dev)
    g = torch.randn(B, Cout, H, W, device=dev) * 1e-5
    gs = torch.randn(B, Cin, H, W, device=dev) * 1e-5

    def run(storage):
        GF.ACT_STORAGE = storage
        try:
            xi, wi, bi = (t.clone().requires_grad_(True) for t in (x, w, bias))
            if with_skip:
                y, skip, stats = GF.conv2d_with_skip(xi, wi, bi, 1, 1, 1, None, True)
                torch.autograd.backward([y, skip], [g, gs])
            else:
                y, stats = GF.conv2d(xi, wi, bi, 1, 1, 1, None, True)
                y.backward(g)
            return y.detach(), stats, xi.grad, wi.grad, bi.grad
        finally:
            GF.ACT_STORAGE = "f32"

    y0, s0, dx0, dw0, db0 = run("f32")
    y1, s1, dx1, dw1, db1 = run("f16")
    assert s1.shape[1] == B * H * W // 64
    close(y1, y0, 3e-3, "forward")
    close(dx1, dx0, 3e-3, "data gradient")
    close(dw1, dw0, 3e-3, "weight gradient")
    close(db1, db0, 1e-4, "bias gradient")


def test_vgg_stack_fp16_storage_vs_fp32(dev):
    """A whole VGG16 backbone under ACT_STORAGE = "f16" against the fp32 kernels: features, input-side gradient and weight
    gradients agree to fp16-storage accuracy (relative L2)."""
    from graphecho_amd import functional as GF
    from graphecho_amd.models.fpnseg import VGG16

    torch.manual_seed(3)
    net = VGG16(1).to(dev).train()
    x = torch.randn(2, 1, 256, 256, device=dev)      # config 5's frame size: every 3x3 layer from the second on is covered

    # a fixed random linear functional of every feature map: (f * f).mean() would be a loss whose gradient is almost parallel
    # to the normalised activation, which BatchNorm's backward projects out -- a cancellation that amplifies any rounding
    gen = torch.Generator(device="cpu").manual_seed(5)
    proj = [None] * 5

    def run(storage):
        GF.ACT_STORAGE = storage
        try:
            for p in net.parameters():
                p.grad = None
            feats = net(x)
            for i, f in enumerate(feats):
                if proj[i] is None:
                    proj[i] = (torch.randn(f.shape, generator=gen) / f.numel() ** 0.5).to(dev)
            loss = sum((f * r).sum() for f, r in zip(feats, proj))
            loss.backward()
            return [f.detach().clone() for f in feats], {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        finally:
            GF.ACT_STORAGE = "f32"

    f32, g32 = run("f32")
    f16, g16 = run("f16")
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    # eval mode (running statistics, no tape): the validation loop's path
    net.eval()
    try:
        with torch.no_grad():
            e32 = net(x)
            GF.ACT_STORAGE = "f16"
            e16 = net(x)
    finally:
        GF.ACT_STORAGE = "f32"
        net.train()
    assert max(rel(a, b) for a, b in zip(e16, e32)) < 1e-2
    print([round(rel(a, b), 5) for a, b in zip(f16, f32)])
    for i, (a, b) in enumerate(zip(f16, f32)):
        assert rel(a, b) < 1e-2, f"feature {i}: {rel(a, b):.3e}"
    errs = {n: rel(g16[n], g32[n]) for n in g32 if n.endswith("weight") and g32[n].dim() == 4}
    print({n: round(v, 4) for n, v in errs.items()})
    # Weight gradients: right norm, but 0.1 - 0.2 relative L2 apart.  Measured cause (tools/debug_half_vgg.py): the fp16 run's
    # activations differ from the fp32 run's by the accumulated rounding (4e-4 after the first stack, 1e-2 after the fifth),
    # which swaps the two largest entries of ~0.5 % of the 2x2 max-pool windows; a swapped window hands its gradient to
    # another -- almost equivalent -- pixel, and that costs 2 g^2 of squared error per window (five pools deep).  The loss
    # barely notices; any fp16-storage implementation behaves so.  The tight check of the chain is the next test (no pool).
    assert max(errs.values()) < 0.3, errs
    # (a conv bias in front of a train-mode BatchNorm has a zero gradient up to rounding noise -- in both runs: skipped)
    for n, v in g32.items():
        if n.endswith(".bias") and isinstance(getattr(net, n.split(".")[0])[int(n.split(".")[1])], torch.nn.Conv2d):
            continue
        ratio = (g16[n].norm() / v.norm().clamp_min(1e-30)).item()
        assert 0.9 < ratio < 1.1, (n, ratio)


class _RoundSTE(torch.autograd.Function):
    """fp16 storage in a torch reference: the value is rounded to fp16 on the way forward, the gradient (times the loss scale)
    on the way back."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return (g * ctx.scale).half().float() / ctx.scale, None


def test_conv_stack_fp16_storage_vs_rounded_storage_reference(dev):
    """Three conv -> BatchNorm -> ReLU layers (a VGG16 stack without its max-pool, whose argmax ties are a property of fp16
    storage, see above) under ACT_STORAGE = "f16" against a plain-PyTorch fp32 computation of the SAME arithmetic: every tensor
    the kernels store as fp16 (conv results, BatchNorm + ReLU outputs, and their gradients times the loss scale) is rounded to
    fp16 at that point (straight-through), conv weights are rounded to fp16, batch statistics come from the un-rounded conv
    results, everything else is fp32.  Output 5e-3, input / weight / affine gradients 2e-2 relative L2."""
    from graphecho_amd import functional as GF
    from graphecho_amd import half as GH
    from graphecho_amd import nn as gnn
    from graphecho_amd.models.fpnseg import _ConvBNStack

    torch.manual_seed(6)
    chans = [64, 128, 128, 128]
    layers = []
    for ci, co in zip(chans[:-1], chans[1:]):
        layers += [gnn.Conv2d(ci, co, kernel_size=(3, 3), stride=(1, 1), padding=1), gnn.BatchNorm2d(co), gnn.ReLU()]
    stack = _ConvBNStack(*layers).to(dev).train()
    for m in stack:
        if isinstance(m, gnn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    x = torch.randn(4, 64, 32, 32, device=dev)
    proj = torch.randn(4, 128, 32, 32, device=dev) / (4 * 128 * 32 * 32) ** 0.5
    GF.h_scale_update()
    S = GF.h_scale_value(dev)
    rnd = lambda t: _RoundSTE.apply(t, S)

    xi = x.clone().requires_grad_(True)
    GF.ACT_STORAGE = "f16"
    try:
        out = GH.from_blocked(stack(xi))
        (out * proj).sum().backward()
    finally:
        GF.ACT_STORAGE = "f32"
    got = {n: p.grad.detach().clone() for n, p in stack.named_parameters()}

    params = {n: p.detach().clone().requires_grad_(True) for n, p in stack.named_parameters()}
    xr = x.clone().requires_grad_(True)
    h = rnd(xr)
    for i in range(0, 9, 3):
        w, bias, gamma, beta = params[f"{i}.weight"], params[f"{i}.bias"], params[f"{i + 1}.weight"], params[f"{i + 1}.bias"]
        w16 = w + (w.half().float() - w).detach()          # fp16 operand copy of the weights, identity gradient
        z32 = F.conv2d(h, w16, bias, padding=1)
        mean = z32.mean(dim=(0, 2, 3), keepdim=True)
        var = z32.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        z = rnd(z32)
        a = F.relu((z - mean) * torch.rsqrt(var + stack[i + 1].eps) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1))
        h = rnd(a)
    (h * proj).sum().backward()
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    e_out = rel(out.detach(), h.detach())
    ge = {n: rel(got[n], params[n].grad) for n in got if not (n.endswith(".bias") and isinstance(stack[int(n.split(".")[0])], gnn.Conv2d))}
    ge["input"] = rel(xi.grad, xr.grad)
    print(round(e_out, 5), {n: round(v, 4) for n, v in ge.items()})
    assert e_out < 5e-3, e_out
    assert max(ge.values()) < 2e-2, ge


def test_sync_batchnorm_world2_fp16_storage(dev, tmp_path):
    """SyncBN inside the fp16-storage stacks (config 5 runs on 8 GPUs): two ranks (gloo, both on cuda:0), each with half of a
    batch, against ONE process that sees the whole batch -- same features (the batch statistics are the whole batch's),
    the ranks' weight / affine gradients add up to the single process's, the running statistics agree."""
    import os
    import subprocess
    import sys

    from graphecho_amd import functional as GF
    from graphecho_amd.models.fpnseg import VGG16

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "ddp_half_worker.py")
    port = str(29900 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert a["sync"][0] >= 10 and a["sync"][1] >= 10, a["sync"]        # the fp16 layers exchanged their statistics
    torch.manual_seed(11)
    net = VGG16(1).to(dev).train()
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(4, 1, 128, 128, generator=gen).to(dev)
    GF.ACT_STORAGE = "f16"
    try:
        feats = net(x)
        proj = [(torch.randn(4, *f.shape[1:], generator=gen) / (4 * f[0].numel()) ** 0.5).to(dev) for f in feats]
        sum((f * r).sum() for f, r in zip(feats, proj)).backward()
    finally:
        GF.ACT_STORAGE = "f32"
    rel = lambda u, v: ((u - v).norm() / v.norm().clamp_min(1e-30)).item()
    fe = [rel(torch.cat([fa, fb]).to(dev), f.detach()) for fa, fb, f in zip(a["feats"], b["feats"], feats)]
    print([round(v, 5) for v in fe])
    assert max(fe) < 5e-3, fe
    close(a["rm"], net.block_2[1].running_mean, 1e-3, "running mean")
    is_conv_bias = lambda n: n.endswith(".bias") and isinstance(getattr(net, n.split(".")[0])[int(n.split(".")[1])], torch.nn.Conv2d)
    ge = {n: rel((a["grads"][n] + b["grads"][n]).to(dev), p.grad) for n, p in net.named_parameters() if not is_conv_bias(n)}
    print({n: round(v, 4) for n, v in ge.items()})
    # below the last BatchNorm the comparison inherits the max-pool argmax sensitivity measured in test_vgg_stack_fp16_storage_
    # vs_fp32 (the two runs' activations differ by 3e-3 after five stacks): tight where no pool lies in between, loose below
    assert ge["block_5.7.weight"] < 2e-2 and ge["block_5.7.bias"] < 1e-1, ge
    assert max(ge.values()) < 0.3, ge


def test_dynamic_loss_scale_follows_the_gradients(dev):
    """The device-resident loss scale: a backward whose gradients are 1e-9 and one whose gradients are 1e+4 both come out right
    (1e4: a scale below one; a fixed scale loses the first to underflow or the second to saturation), the scale after each is the power of two that
    puts the largest magnitude seen at 4096, and an update without a backward in between changes nothing."""
    from graphecho_amd import functional as GF
    from graphecho_amd import half as GH

    if not GF.H_DYNAMIC_SCALE:
        pytest.skip("GE_H_DYNAMIC_SCALE=0")
    torch.manual_seed(7)
    x = torch.randn(2, 64, 32, 32, device=dev)
    w = torch.randn(64, 64, 3, 3, device=dev) / 24.0
    for mag in (1e-9, 1e4, 1e-4):
        for rep in range(2):        # the first pass teaches the scale this magnitude, the second is checked
            xi, wi = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            g = torch.randn(2, 64, 32, 32, device=dev) * mag
            y = GH.from_blocked(GH.conv3x3(GH.to_blocked(xi), wi))
            y.backward(g)
        xr, wr = x.half().float().requires_grad_(True), w.half().float().requires_grad_(True)
        F.conv2d(xr, wr, padding=1).backward(g)
        close(xi.grad, xr.grad, 3e-3, f"data gradient at |g| ~ {mag:g}")
        close(wi.grad, wr.grad, 3e-3, f"weight gradient at |g| ~ {mag:g}")
        GF.h_scale_update()
        sc = GF.h_scale_value(dev)
        want = 2.0 ** torch.floor(torch.log2(GF.H_SCALE_TARGET / g.abs().max())).item()
        assert sc == min(max(want, GF.H_SCALE_MIN), GF.H_SCALE_MAX), (mag, sc, want)
        GF.h_scale_update()
        assert GF.h_scale_value(dev) == sc


def test_saturated_gradients_stay_finite_and_the_scale_recovers(dev, monkeypatch):
    """A loss scale that is far too large for the gradients (the first step of a run knows no magnitude; a spike later on):
    every fp16 store saturates, the gradients of a tensor with two consumers meet in fp32 (half.fork), so NO parameter
    gradient is inf / NaN -- a NaN BatchNorm weight would go unnoticed downstream, ReLU turns it into zeros -- and one update
    later the same backward is accurate again."""
    from graphecho_amd import functional as GF
    from graphecho_amd.models.fpnseg import VGG16

    monkeypatch.setattr(GF, "H_GRAD_SCALE", float(2 ** 24))
    torch.manual_seed(8)
    net = VGG16(1).to(dev).train()
    x = torch.randn(2, 1, 128, 128, device=dev)

    def run(storage):
        GF.ACT_STORAGE = storage
        try:
            for p in net.parameters():
                p.grad = None
            sum(f.sum() for f in net(x)).backward()          # gradients of magnitude 1 at every exit
            return {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        finally:
            GF.ACT_STORAGE = "f32"

    ref = run("f32")
    first = run("f16")            # 1 * 2^24 saturates every cast
    bad = [n for n, g in first.items() if not torch.isfinite(g).all()]
    assert not bad, bad[:4]
    if not GF.H_DYNAMIC_SCALE:
        return
    second = run("f16")           # the forward in front of it brought the scale down to the gradients
    assert GF.h_scale_value(dev) <= 4096.0
    for n, g in second.items():
        if g.dim() == 4:
            ratio = (g.norm() / ref[n].norm().clamp_min(1e-30)).item()
            assert 0.8 < ratio < 1.25, (n, ratio)


def test_fp16_storage_is_bit_reproducible(dev):
    """Two identical forward + backward passes of the VGG16 backbone under ACT_STORAGE = "f16" give identical bits: no float
    atomics anywhere (weight-gradient slabs and BatchNorm partials are folded in a fixed order; the loss-scale record is a MAX,
    which is order-independent)."""
    from graphecho_amd import functional as GF
    from graphecho_amd.models.fpnseg import VGG16

    torch.manual_seed(9)
    net = VGG16(1).to(dev).train()
    x = torch.randn(3, 1, 128, 128, device=dev)
    gen = torch.Generator().manual_seed(10)
    proj = None

    def run():
        nonlocal proj
        GF._H_SCALE.clear()
        GF._H_DIRTY.clear()
        for p in net.parameters():
            p.grad = None
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        GF.ACT_STORAGE = "f16"
        try:
            feats = net(x)
            if proj is None:
                proj = [(torch.randn(f.shape, generator=gen) / f.numel() ** 0.5).to(dev) for f in feats]
            sum((f * r).sum() for f, r in zip(feats, proj)).backward()
        finally:
            GF.ACT_STORAGE = "f32"
        return [f.detach().clone() for f in feats], [p.grad.detach().clone() for p in net.parameters()]

    fa, ga = run()
    fb, gb = run()
    assert all(torch.equal(u, v) for u, v in zip(fa, fb)), "features differ between two identical runs"
    assert all(torch.equal(u, v) for u, v in zip(ga, gb)), "gradients differ between two identical runs"


@pytest.mark.parametrize("Cin,H,W", [(1, 64, 128), (3, 32, 64)])
def test_stem_conv_forward_backward(dev, Cin, H, W):
    """The stem (fp32 image with 1 / 3 channels -> blocked fp16, fp32 FMAs): output and moments against torch, weight / bias
    gradient from a loss-scaled blocked gradient."""
    from graphecho_amd import functional as GF
    from graphecho_amd import half as GH
    from graphecho_amd._lib import lib, check

    torch.manual_seed(12)
    B, Cout = 3, 64
    x = torch.randn(B, Cin, H, W, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) / 3.0).requires_grad_(True)
    bias = torch.randn(Cout, device=dev).requires_grad_(True)
    z, stats = GH._StemConvHFn.apply(x, w, bias, True)
    wr, br = w.detach().clone().requires_grad_(True), bias.detach().clone().requires_grad_(True)
    zr = F.conv2d(x, wr, br, padding=1)
    close(unblk(z), zr, 1e-3, "forward")
    mean, invstd = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    nb = stats.shape[1]
    check(lib.ge_bn_finalize(stats.data_ptr(), nb * 3, 3, nb, Cout, 1e-5, 0.1, None, mean.data_ptr(), invstd.data_ptr(), None,
                             None, None), "finalize")
    close(mean, zr.mean(dim=(0, 2, 3)), 1e-4, "moments: mean")
    close(invstd, torch.rsqrt(zr.var(dim=(0, 2, 3), unbiased=False) + 1e-5), 1e-4, "moments: invstd")
    S = GF.h_scale_value(dev)
    gh = blk(torch.randn_like(zr) * 1e-4 * S)
    z.backward(gh)
    zr.backward(unblk(gh) / S)
    close(w.grad, wr.grad, 1e-4, "weight gradient")
    close(bias.grad, br.grad, 1e-4, "bias gradient")


@pytest.mark.parametrize("segments", [None, (1, 3)])
def test_batch_norm_relu_pool_fused_equals_two_passes(dev, segments):
    """BatchNorm + ReLU + 2x2 max-pool in one pass each way against the two-pass form (bnh_apply + poolh): the pooled map is
    bit-identical (same fp16-rounded activations, same first-maximum rule), dz / dgamma / dbeta agree to the summation order."""
    from graphecho_amd import functional as GF
    from graphecho_amd import half as GH

    torch.manual_seed(13)
    B, Cin, C, H, W = 4, 32, 64, 32, 64
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(C, Cin, 3, 3, device=dev) / (3.0 * Cin ** 0.5)
    z0, stats = GH.conv3x3(blk(x), w, None, None, bn_stats=True)
    S = GF.h_scale_value(dev)
    gp = blk(torch.randn(B, C, H // 2, W // 2, device=dev) * 1e-4 * S)
    res = []
    for fused in (False, True):
        z = z0.detach().clone().requires_grad_(True)
        gamma = (torch.linspace(0.5, 1.5, C, device=dev)).requires_grad_(True)
        beta = (torch.linspace(-0.3, 0.3, C, device=dev)).requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        if fused:
            y = GH.batch_norm(z, gamma, beta, rm, rv, True, 0.1, 1e-5, True, None, stats, segments, pool=True)
        else:
            y = GH.max_pool2(GH.batch_norm(z, gamma, beta, rm, rv, True, 0.1, 1e-5, True, None, stats, segments))
        y.backward(gp)
        res.append((y.detach(), z.grad, gamma.grad, beta.grad, rm, rv))
    (y0, dz0, dg0, db0, rm0, rv0), (y1, dz1, dg1, db1, rm1, rv1) = res
    assert torch.equal(y0, y1), "pooled maps differ"
    assert torch.equal(rm0, rm1) and torch.equal(rv0, rv1)
    close(dg1, dg0, 1e-5, "dgamma")
    close(db1, db0, 1e-5, "dbeta")
    close(unblk(dz1), unblk(dz0), 1e-3, "dz")
    assert (unblk(dz1) != unblk(dz0)).float().mean().item() < 0.01      # a differently rounded last bit at most, on few elements


def test_group_norm8_relu_forward_backward(dev):
    """GroupNorm with 8 channels per group (+ ReLU) on a blocked fp16 tensor (the discriminator towers' GroupNorm(32, 256))
    against torch on the stored fp16 z."""
    from graphecho_amd import functional as GF
    from graphecho_amd import half as GH

    torch.manual_seed(14)
    B, Cin, C, H, W = 3, 64, 256, 16, 32
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(C, Cin, 3, 3, device=dev) / (3.0 * Cin ** 0.5)
    gamma = (torch.rand(C, device=dev) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device=dev) * 0.2).requires_grad_(True)
    z, stats = GH.conv3x3(blk(x), w, None, None, bn_stats=True)
    z = z.detach().requires_grad_(True)
    a = GH._GroupNorm8HFn.apply(z, gamma, beta, 1e-5, True, stats)
    zr = unblk(z.detach()).requires_grad_(True)
    gr_, br_ = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    ar = F.relu(F.group_norm(zr, C // 8, gr_, br_, 1e-5))
    close(unblk(a), ar, 3e-3, "forward")
    S = GF.h_scale_value(dev)
    gh = blk(torch.randn_like(ar) * 1e-5 * S)
    a.backward(gh)
    ar.backward(unblk(gh) / S)
    close(unblk(z.grad) / S, zr.grad, 4e-3, "dz")
    close(gamma.grad, gr_.grad, 2e-3, "dgamma")
    close(beta.grad, br_.grad, 2e-3, "dbeta")


def test_discriminator_tower_fp16_storage_vs_fp32(dev):
    """The Discriminator (fpnseg.py:447-511) with its towers in the blocked fp16 domain against the exact-fp32 kernels: loss
    1e-3, parameter and input gradients 6e-2 relative L2 -- what fp16 OPERANDS cost through four layers with std-0.01 weights
    (measured 4.1e-2 on the first layer and the input, falling to 2e-4 at the last; the fp32-storage route of the same convs,
    GE_H_TOWERS=0, measures 3.9e-2: the storage format is not what this distance consists of)."""
    from graphecho_amd import functional as GF
    from graphecho_amd.models.fpnseg import Discriminator

    torch.manual_seed(15)
    dis = Discriminator().to(dev).train()
    fs = torch.randn(3, 256, 32, 32, device=dev)
    ft = torch.randn(3, 256, 32, 32, device=dev)

    def run(storage):
        GF.ACT_STORAGE = storage
        try:
            for p in dis.parameters():
                p.grad = None
            a, b = fs.clone().requires_grad_(True), ft.clone().requires_grad_(True)
            loss = dis((a, b))
            loss.backward()
            return loss.item(), a.grad, {n: p.grad.detach().clone() for n, p in dis.named_parameters()}
        finally:
            GF.ACT_STORAGE = "f32"

    l0, ga0, g0 = run("f32")
    l1, ga1, g1 = run("f16")
    rel = lambda u, v: ((u - v).norm() / v.norm().clamp_min(1e-30)).item()
    assert abs(l1 - l0) <= 1e-3 * abs(l0), (l1, l0)
    errs = {n: rel(g1[n], g0[n]) for n in g0}
    errs["input"] = rel(ga1, ga0)
    print({n: round(v, 4) for n, v in errs.items()})
    assert max(errs.values()) < 6e-2, errs
    assert errs["dis_tower.9.weight"] < 1e-2 and errs["cls_logits.weight"] < 2e-3, errs


def test_fpn_vgg16_f16s_forward_vs_rounded_storage_oracle(dev):
    """the whole FPN-VGG16 train-mode forward at 256 x 256 in BASELINE config 5's stated dtype (fp16 MFMA conv path,
    fp16 activation storage inside the VGG16 stacks, pool included) against oracle/fpn.py's ROUNDED-STORAGE mode -- the reference's
    arithmetic with an fp16 rounding at exactly the tensors the kernels round (operands of the fp16-MFMA convolutions; conv results and
    BatchNorm + ReLU outputs inside the stacks; batch statistics from the un-rounded conv results).  The routing plan (which conv
    rounds what) is read off the library's own `*_supported` predicates; the arithmetic is the oracle's.

    What the bar can be: thirteen conv -> round -> BatchNorm -> ReLU -> round layers amplify last-bit differences of the fp32
    accumulation into different fp16 roundings (and pool / ReLU decisions) downstream -- two equally valid evaluations of the SAME
    rounded arithmetic, the oracle with fp32- and with fp64-accumulated convolutions, end 5.8e-3 apart on these logits (max norm;
    4.9e-3 RMS).  north_star's 1e-3 is therefore not a property any implementation of this dtype can have; the criterion is "as close
    to the same-dtype oracle as that oracle is to itself": HIP vs either evaluation <= 2 x their spread.  Measured: 6.6e-3 vs a spread
    of 5.8e-3; the fp32 oracle is 1.2e-2 away from the HIP logits and 1.3e-2 from the rounded oracle -- the distance the bench line
    reports for this dtype is the dtype's, not the kernels'."""
    from graphecho_amd import functional as GF
    from graphecho_amd._lib import lib
    from graphecho_amd.models.fpnseg import FPN
    from graphecho_amd.trainer import synthetic_batch
    from oracle import fpn as O

    torch.manual_seed(3)
    net = FPN([2, 4, 23, 3], 4, 1, back_bone="VGG16").to(dev).train()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    x, _ = synthetic_batch(2, 1, 4, 256, "cpu", 4242)
    routed = {}

    def plan(name, xs, ws, stride, padding, groups):
        B, Cin, H, W = xs
        Cout, _, kh, kw = ws
        k3 = kh == 3 and kw == 3 and stride == 1 and padding == 1 and groups == 1
        mode = "f32"
        if "block_" in name and k3 and Cin < 32 and lib.ge_h_stem3x3_supported(B, Cin, Cout, H, W):
            mode = "stem"
        elif "block_" in name and k3 and Cin % 32 == 0 and lib.ge_h_conv3x3_supported(B, Cin, Cout, H, W):
            mode = "f16s"
        elif k3 and GF.H_GENERIC and Cin % 32 == 0 and Cout % 32 == 0 and lib.ge_h_conv3x3_supported(B, Cin, Cout, H, W):
            mode = "f16"
        elif lib.ge_conv2d_f16_supported(Cin, Cout, groups):
            mode = "f16"
        routed[name + str(xs[2:])] = mode
        return mode

    saved = (GF.CONV_PRECISION, GF.ACT_STORAGE)
    GF.CONV_PRECISION, GF.ACT_STORAGE = "f16", "f16"
    try:
        with torch.no_grad():
            logits = net(x.to(dev))[0].float().cpu()
    finally:
        GF.CONV_PRECISION, GF.ACT_STORAGE = saved
    conv32 = O.F.conv2d

    def conv64(x_, w_, b_=None, *args, **kw):      # the same products, accumulated in float64
        return conv32(x_.double(), w_.double(), None if b_ is None else b_.double(), *args, **kw).float()

    with torch.no_grad():
        ref32 = O.fpn_forward({k: v.clone() for k, v in sd.items()}, x, True)[0]
        O.HALF_PLAN = plan
        try:
            ref16 = O.fpn_forward({k: v.clone() for k, v in sd.items()}, x, True)[0]
            O.F.conv2d = conv64
            ref16_64 = O.fpn_forward({k: v.clone() for k, v in sd.items()}, x, True)[0]
        finally:
            O.HALF_PLAN = None
            O.F.conv2d = conv32
    # every VGG16 layer is inside the fp16 domain at this size (the case the bench line's config 5 rows run)
    assert sum(v == "stem" for v in routed.values()) == 1 and sum(v == "f16s" for v in routed.values()) == 12, routed
    scale = ref16.abs().max().item()
    d = lambda u, v: (u - v).abs().max().item() / scale
    err16, err16_64, err32 = d(logits, ref16), d(logits, ref16_64), d(logits, ref32)
    spread, dtype_gap = d(ref16, ref16_64), d(ref16, ref32)
    print(f"f16s logits vs rounded-storage oracle {err16:.2e} (fp64-accumulated: {err16_64:.2e}); the oracle's own spread {spread:.2e}; "
          f"vs fp32 oracle {err32:.2e}; rounded vs fp32 oracle {dtype_gap:.2e}")
    assert 1e-3 < spread < 2e-2, spread                      # the noise floor of the dtype at this depth (documented above)
    assert max(err16, err16_64) <= 2.0 * spread, (err16, err16_64, spread)
    assert err32 > 1.3 * min(err16, err16_64) and abs(err32 - dtype_gap) < 0.6 * dtype_gap, (err16, err32, dtype_gap)
    # thresholded prediction against the same-dtype oracle: no further from it than its second evaluation is
    flips = lambda u, v: ((u > 0) != (v > 0)).float().mean().item()
    assert flips(logits, ref16) <= 2.0 * flips(ref16, ref16_64) + 1e-4, (flips(logits, ref16), flips(ref16, ref16_64))
