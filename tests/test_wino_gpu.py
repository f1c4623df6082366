"""fp32 Winograd F(2x2, 3x3) kernels (graphecho_amd/csrc/ge_wino.hip) against an fp64 convolution: forward (+ bias), data
gradient (+ addend), the routing of functional.conv2d, and the property that makes the route safe -- the error against fp64 is
no larger than the direct kernels'.  Tolerance 5e-6 of the output scale (measured 3e-7 .. 6e-7; direct kernels 0.7 .. 1.7e-6)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cin, Cout, H, W
    (8, 64, 64, 64, 64),        # 2 x 16 tiles per workgroup
    (8, 256, 128, 32, 32),      # Cin != Cout: the data gradient swaps the roles
    (64, 128, 128, 16, 16),     # 4 x 8 tiles per workgroup (W = 16)
    (3, 64, 192, 36, 96),       # odd batch, H = 36 (multiple of 4, not of 8), three 64-channel tiles
    (16, 64, 64, 8, 16),        # 8-row maps: one block row of 4 x 8 tiles
    (8, 256, 256, 16, 16),      # a per-rank step's 16 x 16 level: 64 workgroups -> split over the input channels (slabs + ordered reduce)
    (2, 8, 64, 4, 32),          # ONE chunk of input channels, one workgroup per image
    (4, 72, 64, 8, 32),         # nine chunks (odd count: the pipelined loop runs chunks in pairs, one all-zero chunk more)
    (32, 128, 128, 64, 64),     # 8 192 workgroups, 32 waves of them per CU: the shape on which a mis-counted vmcnt showed (round 5)
    (32, 256, 128, 64, 64),
]


def rel(a, b):
    return ((a.double() - b).abs().max() / b.abs().max()).item()


@pytest.mark.parametrize("case", CASES)
def test_wino3x3_forward_and_data_gradient_vs_fp64(dev, case):
    from graphecho_amd._lib import lib, check

    B, Cin, Cout, H, W = case
    # (ge_wino3x3_supported also asks for a grid that fills the chip -- a routing decision; the kernels take any covered geometry)
    torch.manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5)
    bias = torch.randn(Cout, device=dev)
    dy = torch.randn(B, Cout, H, W, device=dev)
    add = torch.randn(B, Cin, H, W, device=dev)
    p = lambda t: t.data_ptr()
    u = torch.empty(lib.ge_wino3x3_weight_floats(Cin, Cout), device=dev)
    ut = torch.empty_like(u)
    check(lib.ge_wino3x3_pack_weight(p(w), p(u), Cout, Cin, 0, None), "pack")
    has_dgrad = bool(lib.ge_wino3x3_covered(B, Cout, Cin, H, W))      # (the data gradient's output channels are Cin: multiples of 64)
    if has_dgrad:
        check(lib.ge_wino3x3_pack_weight(p(w), p(ut), Cin, Cout, 1, None), "pack_t")
    y = torch.full((B, Cout, H, W), float("nan"), device=dev)
    dx = torch.full((B, Cin, H, W), float("nan"), device=dev)
    wsf = torch.full((max(1, lib.ge_wino3x3_workspace(B, Cin, Cout, H, W)),), float("nan"), device=dev)
    wsd = torch.full((max(1, lib.ge_wino3x3_workspace(B, Cout, Cin, H, W)),), float("nan"), device=dev)
    split = lib.ge_wino3x3_splits(B, Cin, Cout, H, W) > 1
    # BatchNorm moments from the epilogue (unsplit layers): one (count, mean, M2) triple per workgroup and channel
    parts = lib.ge_wino3x3_stat_parts(B, H, W)
    stats = None if split else torch.full((Cout, parts, 3), float("nan"), device=dev)
    check(lib.ge_wino3x3_fwd(p(x), p(u), p(bias), None, p(y), None if split else p(stats), p(wsf), B, Cin, Cout, H, W, None), "fwd")
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    assert rel(y, ref) < 5e-6, rel(y, ref)            # every output written (NaN fill), borders included
    if has_dgrad:
        check(lib.ge_wino3x3_fwd(p(dy), p(ut), None, p(add), p(dx), None, p(wsd), B, Cout, Cin, H, W, None), "dgrad")
        refd = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1) + add.double()
        assert rel(dx, refd) < 5e-6, rel(dx, refd)
    if stats is not None:
        # merged over the parts (Chan): count, mean, biased variance of every channel of the kernel's OWN output
        n, mean, m2 = stats[:, :, 0].double(), stats[:, :, 1].double(), stats[:, :, 2].double()
        assert torch.all(n == 128)
        tot = n.sum(1)
        gmean = (n * mean).sum(1) / tot
        gm2 = (m2 + n * (mean - gmean[:, None]) ** 2).sum(1)
        yy = y.double().transpose(0, 1).reshape(Cout, -1)
        assert torch.all(tot == B * H * W)
        assert (gmean - yy.mean(1)).abs().max() < 1e-5 * yy.abs().max()
        assert ((gm2 / tot) / yy.var(1, unbiased=False) - 1).abs().max() < 1e-4
        # a part covers 128 positions of ONE image, in image order (what lets BatchNorm split the moments by segment)
        per_img = parts // B
        for b in (0, B - 1):
            sl = slice(b * per_img, (b + 1) * per_img)
            mb = (n[:, sl] * mean[:, sl]).sum(1) / n[:, sl].sum(1)
            assert (mb - y[b].double().reshape(Cout, -1).mean(1)).abs().max() < 1e-5 * yy.abs().max()


def test_wino3x3_split_path_is_bit_reproducible_and_matches_unsplit(dev, monkeypatch):
    """The K-split path (slabs + ordered reduce) against the unsplit kernel on the same layer: same result to fp32 rounding of a
    differently associated sum, identical bits run to run."""
    import os
    import subprocess
    import sys

    code = (
        "import os, sys, torch; sys.path.insert(0, os.getcwd())\n"
        "from graphecho_amd._lib import lib, check\n"
        "dev = torch.device('cuda:0'); torch.manual_seed(1)\n"
        "B, C, M, S = 8, 256, 256, 16\n"
        "x = torch.randn(B, C, S, S, device=dev); w = torch.randn(M, C, 3, 3, device=dev) / 48; b = torch.randn(M, device=dev)\n"
        "u = torch.empty(16 * C * M, device=dev); check(lib.ge_wino3x3_pack_weight(w.data_ptr(), u.data_ptr(), M, C, 0, None), 'p')\n"
        "ws = torch.empty(max(1, lib.ge_wino3x3_workspace(B, C, M, S, S)), device=dev)\n"
        "outs = []\n"
        "for _ in range(3):\n"
        "    y = torch.empty(B, M, S, S, device=dev)\n"
        "    check(lib.ge_wino3x3_fwd(x.data_ptr(), u.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, ws.data_ptr(), B, C, M, S, S, None), 'f')\n"
        "    outs.append(y)\n"
        "assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])\n"
        "torch.save(outs[0].cpu(), sys.argv[1]); print('splits', lib.ge_wino3x3_splits(B, C, M, S, S))\n")
    import tempfile

    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, env in (("plan", {}), ("one", {"GE_WN_SPLITS": "1"}), ("eight", {"GE_WN_SPLITS": "8"})):
            out = os.path.join(tmp, tag + ".pt")
            r = subprocess.run([sys.executable, "-c", code, out], env={**os.environ, **env}, capture_output=True, text=True,
                               cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            assert r.returncode == 0, r.stdout + r.stderr
            res[tag] = (torch.load(out), r.stdout)
    assert "splits 1" in res["one"][1] and "splits 8" in res["eight"][1], (res["one"][1], res["eight"][1])
    assert "splits 1" not in res["plan"][1]          # 64 workgroups: the plan splits this layer
    scale = res["one"][0].abs().max()
    for tag in ("plan", "eight"):
        assert (res[tag][0] - res["one"][0]).abs().max() < 5e-6 * scale


def test_winograd_operands_are_repacked_by_the_model_packer_under_graph_replay(dev, monkeypatch):
    """a Winograd layer of a FlatParams model reads a PERSISTENT operand buffer that the model-wide packer refreshes
    after every optimizer step -- also when its forward / backward are replayed from HIP graphs that share the layer (two tags
    captured in one step).  Five optimizer steps graph-replayed == five steps eager (to fp32 rounding)."""
    from graphecho_amd import functional as GF
    from graphecho_amd import nn as gnn
    from graphecho_amd.graphs import GraphedModule
    from graphecho_amd.optim import FlatSGD

    monkeypatch.setattr(GF, "WINOGRAD_MIN_BLOCKS", 1)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = gnn.Conv2d(64, 64, 3, padding=1)
            self.c2 = gnn.Conv2d(64, 64, 3, padding=1)

        def forward(self, x):
            return self.c2(GF.relu(self.c1(x)))

    def run(graphed):
        torch.manual_seed(5)
        net = Net().to(dev).train()
        opt = FlatSGD(net, lr=0.05)
        gm = GraphedModule(net, [opt.fp], warmup=1)
        gm.enabled = graphed
        xa, xb = torch.randn(4, 64, 32, 32, device=dev), torch.randn(4, 64, 32, 32, device=dev)
        losses = []
        for step in range(5):
            opt.zero_grad()
            GF.DIRECT_GRAD_ACCUM = True
            try:
                la = gm(xa, tag="a").square().mean()
                lb = gm(xb, tag="b").square().mean()      # second slot of the same module in the same step: a pack-cache hit
                (la + lb).backward()
            finally:
                GF.DIRECT_GRAD_ACCUM = False
            opt.step()
            losses.append((la + lb).item())
        return losses, net.c1.weight.detach().clone(), gm

    le, we, _ = run(False)
    lg, wg, gm = run(True)
    assert gm.graphs()[0] == 2 and gm.graphs()[1] >= 1
    assert all(abs(a - b) <= 1e-6 * abs(a) for a, b in zip(le, lg)), (le, lg)      # (a stale operand shows up at the 1e-2 level)
    assert (we - wg).abs().max() <= 1e-6 * we.abs().max()
    assert le[-1] < le[0]          # the weights the graphs read DID change from step to step


def test_fpn_forward_backward_on_winograd_kernels_vs_oracle(dev, monkeypatch):
    """The ResNet FPN of config 1 (2 frames, 256 x 256) with every covered 3x3 layer FORCED onto the Winograd kernels (two frames
    are below their routing threshold) against the fp32 CPU oracle: logits / pyramid 1e-3 (north_star), parameter gradients
    within the fp32-vs-fp64 yardstick of tests/test_models_gpu.py."""
    from graphecho_amd import functional as GF
    from graphecho_amd.models.fpnseg import FPN
    from oracle.fpn import fpn_forward
    from oracle.misc import seg_loss_cardiac
    from oracle.weights import fill_state_dict
    from test_models_gpu import _check_grads, _relerr

    monkeypatch.setattr(GF, "WINOGRAD_MIN_BLOCKS", 1)
    torch.manual_seed(0)
    net = FPN([2, 4, 23, 3], 4, 3)
    sd = fill_state_dict(net.state_dict(), seed=1)
    net.load_state_dict(sd)
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 256, 256, generator=gen)
    t = (torch.rand(2, 4, 256, 256, generator=gen) > 0.6).float()

    def run_oracle(dtype, perturb=0.0):
        params = {k: (v.detach().to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        params = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in params.items()}
        xin = x.to(dtype)
        if perturb:
            xin = xin * (1 + perturb * torch.randn(x.shape, generator=torch.Generator().manual_seed(11)).to(dtype))
        lg, pyr = fpn_forward(params, xin, True)
        seg_loss_cardiac(lg, t.to(dtype)).backward()
        return params, lg, pyr

    p32, ref_logits, ref_pyr = run_oracle(torch.float32)
    p64, _, _ = run_oracle(torch.float64)
    p32p, _, _ = run_oracle(torch.float32, 1e-6)
    net = net.to(dev).train()
    GF.KERNEL_TIMER = GF.KernelTimer()      # records the kernel instantiation of every conv launch
    try:
        logits, pyr = net(x.to(dev))
        loss = GF.dice_loss(logits, t.to(dev)) + GF.bce_with_logits(logits, t.to(dev))
        loss.backward()
        names = [(r[0], r[1]) for r in GF.KERNEL_TIMER.records]
    finally:
        GF.KERNEL_TIMER = None
    wino = [k for k, n in names if "wino3x3" in n]
    assert sum(k.startswith("conv_fwd") for k in wino) >= 10 and sum(k.startswith("conv_dgrad") for k in wino) >= 10, wino
    assert _relerr(logits, ref_logits) < 1e-3
    for a, b in zip(pyr, ref_pyr):
        assert _relerr(a, b) < 1e-3
    _check_grads([(n, p.grad) for n, p in net.named_parameters()], p32, p64, "resnet/winograd", p32p)


def test_conv2d_routes_large_3x3_layers_through_winograd(dev, monkeypatch):
    """functional.conv2d (fp32): a covered layer runs on wino3x3_kernel forward and backward, agrees with the direct kernels to
    fp32 rounding, and is no further from fp64 than they are; GE_WINOGRAD=0 / functional.WINOGRAD = False restores them."""
    from graphecho_amd import functional as GF
    from graphecho_amd._lib import lib

    torch.manual_seed(4)
    B, Cin, Cout, S = 32, 128, 128, 32      # 512 workgroups: the routing threshold
    x0 = torch.randn(B, Cin, S, S, device=dev)
    w0 = torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5)
    b0 = torch.randn(Cout, device=dev)
    g = torch.randn(B, Cout, S, S, device=dev)

    def run(flag):
        monkeypatch.setattr(GF, "WINOGRAD", flag)
        x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = GF.conv2d(x, w, b, 1, 1, 1)
        k_f = lib.ge_last_conv_kernel().decode()
        y.backward(g)
        return y.detach(), x.grad, w.grad, b.grad, k_f

    yw, dxw, dww, dbw, kw = run(True)
    yd, dxd, dwd, dbd, kd = run(False)
    assert "wino3x3" in kw and "wino3x3" not in kd, (kw, kd)
    ref = F.conv2d(x0.double(), w0.double(), b0.double(), padding=1)
    refd = torch.nn.grad.conv2d_input(x0.shape, w0.double(), g.double(), padding=1)
    assert rel(yw, ref) <= max(rel(yd, ref), 1e-6) and rel(dxw, refd) <= max(rel(dxd, refd), 1e-6)
    assert rel(yw, yd.double()) < 5e-6 and rel(dxw, dxd.double()) < 5e-6
    # the weight gradient takes the F(3x3, 2x2) kernel with the route on, the direct one with it off: both against fp64
    refw = torch.nn.grad.conv2d_weight(x0.double(), w0.shape, g.double(), padding=1)
    assert rel(dww, refw) <= max(2 * rel(dwd, refw), 2e-6), (rel(dww, refw), rel(dwd, refw))
    # (the bias gradient comes out of the same pass: row sums of dy in the F(3x3, 2x2) kernel's tile order, or ge_channel_sum's)
    refb = g.double().sum(dim=(0, 2, 3))
    assert rel(dww, dwd.double()) < 1e-5 and rel(dbw, refb) < 1e-5 and rel(dbd, refb) < 1e-5


WGRAD_CASES = [
    # B, Cin, Cout, H, W
    (8, 64, 64, 64, 64),        # four strips per tile row
    (4, 32, 128, 32, 32),       # Cin = one channel tile, two output tiles
    (6, 96, 64, 16, 16),        # ONE strip per tile row: every strip touches the left and the right border
    (3, 32, 64, 6, 48),         # odd batch, three tile rows
    (32, 128, 128, 32, 32),     # a layer of the timed step (Bottleneck.conv2 of layer2)
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wino3x3_weight_gradient_vs_fp64(dev, case, monkeypatch):
    """F(3x3, 2x2) weight gradient (ge_wino_wgrad.hip) against an fp64 correlation: every tap of every (m, c) pair, borders
    included (NaN-filled destination), the accumulate form, and the same bits on a second launch.  Tolerance 1e-5 of the largest
    gradient element (K = B H W / 4 products per plane and element; the direct kernel's error on the same data is printed)."""
    from graphecho_amd._lib import lib, check

    B, Cin, Cout, H, W = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, device=dev)
    dy = torch.randn(B, Cout, H, W, device=dev)
    p = lambda t: t.data_ptr()
    n_ws = lib.ge_wino3x3_wgrad_workspace(B, Cin, Cout, H, W)
    splits = lib.ge_wino3x3_wgrad_splits(B, Cin, Cout, H, W)
    assert splits >= 1 and n_ws == splits * Cout * Cin * 9 + splits * Cout, (splits, n_ws)      # weight slabs + bias slabs
    ws = torch.full((n_ws,), float("nan"), device=dev)
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=dev)
    check(lib.ge_wino3x3_wgrad(p(x), p(dy), p(dw), p(ws), B, Cin, Cout, H, W, 0, None), "wgrad")
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), padding=1)
    e = rel(dw, ref)
    # the direct kernel on the same data
    wsd = torch.empty(lib.ge_conv2d_wgrad_workspace(B, Cin, Cout, H, W, 3, 3, 1), device=dev)
    dwd = torch.empty_like(dw)
    check(lib.ge_conv2d_wgrad(p(x), p(dy), p(dwd), p(wsd), B, Cin, H, W, Cout, H, W, 3, 3, 1, 1, 1, 0, None), "direct")
    print(f"{case}: {splits} splits, error vs fp64: winograd {e:.2e}, direct {rel(dwd, ref):.2e}")
    assert e < 1e-5, e
    dw2 = torch.full_like(dw, 0.25)
    check(lib.ge_wino3x3_wgrad(p(x), p(dy), p(dw2), p(ws), B, Cin, Cout, H, W, 1, None), "wgrad accumulate")
    assert (dw2 - (0.25 + dw)).abs().max().item() <= 2e-6 * dw.abs().max().item()      # the same slabs, added to what was there
    dw3 = torch.empty_like(dw)
    check(lib.ge_wino3x3_wgrad(p(x), p(dy), p(dw3), p(ws), B, Cin, Cout, H, W, 0, None), "wgrad again")
    assert torch.equal(dw3, dw)
    # the form that also produces the bias gradient (row sums of dy from the c-tile-0 workgroups): same dw bits, db vs fp64,
    # accumulate adds to both
    ws.fill_(float("nan"))
    dw4, db = torch.full_like(dw, float("nan")), torch.full((Cout,), float("nan"), device=dev)
    check(lib.ge_wino3x3_wgrad_bias(p(x), p(dy), p(dw4), p(db), p(ws), B, Cin, Cout, H, W, 0, None), "wgrad_bias")
    refb = dy.double().sum(dim=(0, 2, 3))
    assert torch.equal(dw4, dw) and rel(db, refb) < 1e-5, rel(db, refb)
    db2 = torch.full_like(db, 2.0)
    check(lib.ge_wino3x3_wgrad_bias(p(x), p(dy), p(dw4), p(db2), p(ws), B, Cin, Cout, H, W, 1, None), "wgrad_bias accumulate")
    assert (db2 - (2.0 + db)).abs().max().item() <= 2e-6 * max(1.0, db.abs().max().item())


def test_wino3x3_weight_gradient_adjoint_full_size(dev):
    """At the timed step's largest 3x3 layer (256 -> 256 @ 64 x 64, batch 32), where an fp64 reference would take minutes:
    <dw, w> == <conv(x; w), dy> for a random w (fp64 sums of fp32 results; the forward through functional.conv2d)."""
    from graphecho_amd import functional as GF
    from graphecho_amd._lib import lib, check

    B, C, M, S = 32, 256, 256, 64
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B, C, S, S, device=dev, generator=gen)
    dy = torch.randn(B, M, S, S, device=dev, generator=gen)
    w = torch.randn(M, C, 3, 3, device=dev, generator=gen) / 48
    ws = torch.empty(lib.ge_wino3x3_wgrad_workspace(B, C, M, S, S), device=dev)
    dw = torch.empty(M, C, 3, 3, device=dev)
    check(lib.ge_wino3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, C, M, S, S, 0, None), "wgrad")
    y = GF.conv2d(x, w, None, 1, 1)
    ip_y = (y.double() * dy.double()).sum().item()
    ip_w = (dw.double() * w.double()).sum().item()
    scale = (y.double().norm() * dy.double().norm()).item()
    assert abs(ip_y - ip_w) <= 1e-5 * scale, (ip_y, ip_w)


@pytest.mark.parametrize("ws", ["0", "1"])
def test_wino3x3_weight_gradient_both_kernels(dev, ws):
    """The plan picks wino3x3_wgrad_kernel or the warp-specialised wino3x3_wgrad_ws_kernel per layer (read once per process):
    GE_WNW_WS forces one of them in a child process, which runs every case of WGRAD_CASES plus an odd number of chunks per split
    (the specialised kernel's loop is unrolled by two) against the fp64 correlation, twice (same bits)."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, torch
sys.path.insert(0, %r)
from graphecho_amd._lib import lib, check
dev = torch.device("cuda:0")
cases = %r + [(1, 32, 64, 6, 16), (5, 32, 64, 18, 16)]
for case in cases:
    B, Cin, Cout, H, W = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, device=dev); dy = torch.randn(B, Cout, H, W, device=dev)
    ws = torch.full((lib.ge_wino3x3_wgrad_workspace(B, Cin, Cout, H, W),), float("nan"), device=dev)
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=dev); dw2 = dw.clone()
    check(lib.ge_wino3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, Cin, Cout, H, W, 0, None), "wgrad")
    db = torch.full((Cout,), float("nan"), device=dev)
    check(lib.ge_wino3x3_wgrad_bias(x.data_ptr(), dy.data_ptr(), dw2.data_ptr(), db.data_ptr(), ws.data_ptr(), B, Cin, Cout, H, W, 0, None), "wgrad_bias")
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), padding=1)
    e = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
    refb = dy.double().sum(dim=(0, 2, 3))
    eb = ((db.double() - refb).abs().max() / refb.abs().max()).item()
    assert e < 1e-5 and eb < 1e-5 and torch.equal(dw, dw2), (case, e, eb)
    print(case, lib.ge_wino3x3_wgrad_splits(B, Cin, Cout, H, W), "splits", "%%.1e" %% e)
print("both ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), WGRAD_CASES)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "GE_WNW_WS": ws}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "both ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
