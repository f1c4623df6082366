"""Module-level parity on the MI355X: HIP modules vs the CPU oracle driven by the same state_dict."""
import pytest
import torch

# L2-relative bound on the FPN input gradient against the reference fixture (it runs back through all 50 train-mode BatchNorm
# layers; measured 1.4e-2 .. 2.4e-2 over the three fixture cases, the fp32 CPU oracle is as far from fp64)
GX_TOL = 3.5e-2

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def _check_grads(named_grads, g32, g64, what, g32p=None):
    """HIP gradients must be as accurate as the fp32 CPU oracle when both are measured against an fp64 oracle.
    g32p (optional): the fp32 oracle's gradients from an input perturbed by 1e-6 relative -- less than what ONE fp32
    convolution layer injects (measured against fp64: direct kernels 1.4e-6, Winograd kernels 5e-7, torch CPU 1e-6 of the output
    scale): a second, equally valid fp32 run.  Where max-pool argmax / ReLU flips make a tensor's gradient jump under such a
    perturbation (VGG16's deep BatchNorm weights: 1e-2), the larger of the two fp32 errors is the yardstick.

    Train-mode BatchNorm chains with O(1) random weights are ill-conditioned (ReLU masks flip under 1e-7
    perturbations), so fp32-vs-fp32 differences of 1e-1 on single elements are expected even between two
    correct fp32 implementations.  Criterion (relative L2 errors against the fp64 gradients):
      * whole-model: err_hip <= 2 * err_cpu32 + 1e-3
      * per tensor : err_hip <= 10 * err_cpu32 + 5e-3
    Structurally-zero gradients (a bias in front of a per-channel GroupNorm) are compared absolutely."""
    num_h = num_c = den = 0.0
    worst = []
    for name, g in named_grads:
        t = g64[name].grad
        nt = t.norm().item()
        e_hip = (g.detach().cpu().double() - t).norm().item()
        e_cpu = (g32[name].grad.double() - t).norm().item()
        if g32p is not None:
            e_p = (g32p[name].grad.double() - t).norm().item()
            if t.abs().max().item() >= 1e-9:
                worst.append((e_hip / nt, e_cpu / nt, e_p / nt, name))
            e_cpu = max(e_cpu, e_p)
        if t.abs().max().item() < 1e-9:
            assert e_hip < 1e-4, f"{what}:{name} should be ~0, got {e_hip:.2e}"
            continue
        assert e_hip <= 10 * e_cpu + 5e-3 * nt, f"{what}:{name}: hip {e_hip/nt:.2e} vs cpu32 {e_cpu/nt:.2e}"
        num_h += e_hip ** 2
        num_c += e_cpu ** 2
        den += nt ** 2
    for w in sorted(worst, reverse=True)[:3]:
        print(f"{what}: {w[3]}: hip {w[0]:.2e}, cpu32 {w[1]:.2e}, cpu32 on the perturbed input {w[2]:.2e}")
    eh, ec = (num_h / den) ** 0.5, (num_c / den) ** 0.5
    assert eh <= 2 * ec + 1e-3, f"{what}: whole-model gradient error hip {eh:.2e} vs cpu32 {ec:.2e}"


@pytest.mark.parametrize("bb,cin,nc,hw", [("resnet", 3, 4, 128), ("VGG16", 1, 1, 128), ("resnet", 1, 3, 96),
                                         ("VGG16", 1, 4, 256)])
def test_fpn_forward_backward_vs_oracle(dev, bb, cin, nc, hw):
    """Logits / pyramid within 1e-3 rel of the fp32 oracle (north_star tolerance); gradients as accurate as it.
    Last case: config 5's FPN as the reference builds it (train_cardiac_uda.py:73: VGG16, one input channel, four
    classes, 256 x 256) against the fp64 yardstick."""
    from graphecho_amd.models.fpnseg import FPN
    from graphecho_amd import functional as GF
    from oracle.fpn import fpn_forward
    from oracle.misc import seg_loss_cardiac
    from oracle.weights import fill_state_dict

    torch.manual_seed(0)
    net = FPN([2, 4, 23, 3], nc, cin, back_bone=bb)
    sd = fill_state_dict(net.state_dict(), seed=1)
    net.load_state_dict(sd)
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(2, cin, hw, hw, generator=gen)
    t = (torch.rand(2, nc, hw, hw, generator=gen) > 0.6).float()

    def run_oracle(dtype, perturb=0.0):
        params = {k: (v.detach().to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        params = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v)
                  for k, v in params.items()}
        xin = x.to(dtype)
        if perturb:
            xin = xin * (1 + perturb * torch.randn(x.shape, generator=torch.Generator().manual_seed(11)).to(dtype))
        lg, pyr = fpn_forward(params, xin, True)
        ls = seg_loss_cardiac(lg, t.to(dtype))
        ls.backward()
        return params, lg, pyr, ls

    p32, ref_logits, ref_pyr, ref_loss = run_oracle(torch.float32)
    p64, _, _, _ = run_oracle(torch.float64)
    p32p, _, _, _ = run_oracle(torch.float32, perturb=1e-6)

    net = net.to(dev).train()
    logits, pyr = net(x.to(dev))
    loss = GF.dice_loss(logits, t.to(dev)) + GF.bce_with_logits(logits, t.to(dev))
    loss.backward()

    assert _relerr(logits, ref_logits) < 1e-3
    for a, b in zip(pyr, ref_pyr):
        assert _relerr(a, b) < 1e-3
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    # Dice of the thresholded prediction against the oracle's (sigmoid > 0.5; (2TP+e)/(2TP+FP+FN+e))
    p, r = logits.detach().cpu() > 0, ref_logits.detach() > 0
    tp, fp, fn = (p & r).sum().item(), (p & ~r).sum().item(), (~p & r).sum().item()
    assert (2 * tp + 1e-5) / (2 * tp + fp + fn + 1e-5) > 0.999, (tp, fp, fn)
    _check_grads([(n, p.grad) for n, p in net.named_parameters()], p32, p64, bb, p32p)
    sd_after = net.state_dict()
    key = next(k for k in sd_after if k.endswith("running_mean"))
    assert not torch.equal(sd_after[key].cpu(), sd[key])
    assert _relerr(sd_after[key], 0.9 * sd[key] + 0.1 * _batch_mean_of_first_bn(sd, x, bb)) < 1e-3


def _batch_mean_of_first_bn(sd, x, bb):
    import torch.nn.functional as F

    if bb == "resnet":
        y = F.conv2d(x, sd["back_bone.conv1.weight"], None, 2, 3)
    else:
        y = F.conv2d(x, sd["back_bone.block_1.0.weight"], sd["back_bone.block_1.0.bias"], 1, 1)
    return y.mean((0, 2, 3))


def test_discriminator_vs_oracle(dev):
    from graphecho_amd.models.fpnseg import Discriminator
    from oracle.fpn import discriminator_forward
    from oracle.weights import fill_state_dict

    torch.manual_seed(0)
    dis = Discriminator(grad_reverse_lambda=0.02)
    sd = fill_state_dict(dis.state_dict(), seed=2)
    dis.load_state_dict(sd)
    gen = torch.Generator().manual_seed(4)
    fs = torch.randn(2, 256, 16, 16, generator=gen).requires_grad_(True)
    ft = torch.randn(2, 256, 16, 16, generator=gen).requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = discriminator_forward(params, (fs, ft), 0.02)
    ref.backward()
    dis = dis.to(dev)
    gs, gt = fs.detach().to(dev).requires_grad_(True), ft.detach().to(dev).requires_grad_(True)
    out = dis((gs, gt))
    out.backward()
    assert abs(out.item() - ref.item()) < 1e-4
    assert _relerr(gs.grad, fs.grad) < 5e-3 and _relerr(gt.grad, ft.grad) < 5e-3
    for name, p in dis.named_parameters():
        assert _relerr(p.grad, params[name].grad) < 5e-3, name


# ----------------------------------------------------------------------------------------------------------
# Grapher / TGCN / GModule / attention / affinity against the oracle AND the reference-generated fixtures
# ----------------------------------------------------------------------------------------------------------
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def _close(a, b, rtol, what):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    assert a.shape == b.shape, f"{what}: {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = max(b.abs().max().item(), 1e-8)
    err = (a - b).abs().max().item()
    assert err <= rtol * scale + 1e-9, f"{what}: err {err:.3e} scale {scale:.3e}"


def _no_dropout(m):
    for s in m.modules():
        if isinstance(s, (torch.nn.Dropout, torch.nn.Dropout2d)):
            s.p = 0.0
        if hasattr(s, "p") and s.__class__.__name__ == "dot_attention":
            s.p = 0.0
    return m


@pytest.mark.parametrize("tag,C,hw,r", [("c64_r2", 64, 16, 2), ("c256_r1", 256, 8, 1), ("c256_r4_64", 256, 64, 4)])
def test_grapher_vs_reference_fixture(dev, tag, C, hw, r):
    """"c256_r4_64" is config 2's p2 block at its own size (survey F4): 4096 nodes against 256 pooled candidates."""
    from graphecho_amd.models.vig import Grapher
    from oracle.weights import det_tensor, fill_state_dict

    g = _gold("grapher_" + tag)
    mod = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.0, r, n=hw * hw)
    mod.load_state_dict(fill_state_dict(mod.state_dict(), seed=3))
    mod = mod.to(dev).train()
    x = det_tensor(f"grapher.{tag}.x", (2, C, hw, hw)).to(dev).requires_grad_(True)
    y = mod(x)
    (y * det_tensor(f"grapher.{tag}.g", tuple(y.shape)).to(dev)).sum().backward()
    sp = 4 if hw >= 64 else 1
    _close(y[:, ::8, ::sp, ::sp], g["y"], 1e-3, "grapher out")
    _close(x.grad[:, ::8, ::sp, ::sp], g["g_x"], 5e-3, "grapher d x")
    _close(mod.fc1[0].weight.grad[:8, :8, 0, 0], g["g_fc1"], 5e-3, "d fc1")
    _close(mod.graph_conv.gconv.nn[0].weight.grad[:8, :8, 0, 0], g["g_gconv"], 5e-3, "d gconv")


GRAPHCONV_CASES = {  # tag: (conv, act, norm, C_in, C_out, N, M or None) -- tools/gen_golden.py:GRAPHCONV_CASES
    "edge_relu_batch": ("edge", "relu", "batch", 32, 64, 49, None),
    "edge_leaky_batch_xy": ("edge", "leakyrelu", "batch", 32, 48, 50, 16),
    "sage_prelu_batch_xy": ("sage", "prelu", "batch", 32, 64, 50, 16),
    "gin_hswish_none": ("gin", "hswish", None, 32, 64, 49, None),
    "mr_relu_none_xy": ("mr", "relu", None, 32, 64, 50, 16),
}


@pytest.mark.parametrize("tag", list(GRAPHCONV_CASES))
def test_graphconv_variants_vs_reference_fixture(dev, tag):
    """GraphConv2d with each aggregator / activation / norm (vig.py:88-181,433-500) on the reference's own edges."""
    from graphecho_amd.models.vig import DenseDilatedKnnGraph, GraphConv2d
    from oracle.weights import det_tensor, fill_state_dict

    conv, act, norm, ci, co, N, M = GRAPHCONV_CASES[tag]
    g = _gold("graphconv")
    mod = GraphConv2d(ci, co, conv, act, norm, True)
    mod.load_state_dict(fill_state_dict(mod.state_dict(), seed=7))
    mod = mod.to(dev).train()
    x = det_tensor(f"gconv.{tag}.x", (2, ci, N, 1)).to(dev).requires_grad_(True)
    y = det_tensor(f"gconv.{tag}.y", (2, ci, M, 1)).to(dev).requires_grad_(True) if M else None
    mine = DenseDilatedKnnGraph(9, 1)(x, y)
    assert (mine.cpu().numpy() == g[tag + ".edge"]).mean() > 0.995
    edge = torch.from_numpy(g[tag + ".edge"]).to(dev)
    out = mod(x, edge, y)
    (out * det_tensor(f"gconv.{tag}.g", tuple(out.shape)).to(dev)).sum().backward()
    _close(out, g[tag + ".out"], 1e-4, "out")
    _close(x.grad, g[tag + ".g_x"], 1e-3, "d x")
    if M:
        _close(y.grad, g[tag + ".g_y"], 1e-3, "d y")
    w = mod.gconv.nn1[0].weight if conv == "sage" else mod.gconv.nn[0].weight
    _close(w.grad, g[tag + ".g_w"], 1e-3, "d w")
    if conv == "gin":
        _close(mod.gconv.eps.grad, g[tag + ".g_eps"], 1e-3, "d eps")


def test_pvig_vs_oracle_stage_by_stage(dev):
    """Pyramid ViG tiny at 224x224 -- relative-position k-NN with dilation 1..3 (K up to 27), odd
    channel widths (48/96/240/384), stride-2 Downsample convs, fused conv+BN(+residual) in Grapher/FFN.

    The oracle is pinned to the reference's fixture end to end (tests/test_oracle_golden.py).  The HIP path is compared
    with the oracle one stage at a time ON THE ORACLE'S INPUT, forward and backward: end-to-end comparison is not a
    usable criterion for this network because a dilated k-NN over 27 of 49 candidates flips a neighbour under 1e-6
    input noise (measured: 3 flips per forward) and every later block amplifies the flip."""
    from graphecho_amd import functional as GF
    from graphecho_amd.models.vig import pvig_ti_224_gelu
    from oracle.vig import deepgcn_forward, deepgcn_stages
    from oracle.weights import det_tensor, fill_state_dict

    mod = pvig_ti_224_gelu(num_classes=10)
    sd = mod.state_dict()
    filled = fill_state_dict(sd, seed=5)
    for k in sd:
        if "relative_pos" in k:
            filled[k] = sd[k].clone()
    mod.load_state_dict(filled)
    x = det_tensor("pvig.x", (2, 3, 224, 224), "uniform")
    taps = []
    with torch.no_grad():
        y_ref = deepgcn_forward(filled, x, [2, 2, 6, 2], taps=taps)
    _close(y_ref, _gold("pvig_ti")["y"], 1e-3, "oracle logits vs reference fixture")
    mod = mod.to(dev).train()

    def hip_stage(tag):
        if tag == "stem+pos":
            return lambda h: mod.stem(h) + mod.pos_embed
        if tag == "prediction":
            return lambda h: mod.prediction(GF.adaptive_avg_pool2d_1(h)).squeeze(-1).squeeze(-1)
        parts = tag.split(".")
        blk = mod.backbone[int(parts[1])]
        return blk if len(parts) == 2 else blk[int(parts[2])]

    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
                  and "relative_pos" not in k else v.clone()) for k, v in filled.items()}
    ref_stages = dict(deepgcn_stages(params, [2, 2, 6, 2]))
    grad_tags = {"stem+pos", "backbone.0.0", "backbone.2", "backbone.3.1", "backbone.6.0", "backbone.11.0",
                 "backbone.14.0", "backbone.14.1", "prediction"}       # dilation 1, 2 and 3 Graphers among them
    flips = 0
    for tag, xin, want in taps:
        need_grad = tag in grad_tags
        xg = xin.to(dev).requires_grad_(need_grad)
        with torch.enable_grad() if need_grad else torch.no_grad():
            out = hip_stage(tag)(xg)
        scale = want.abs().max()
        err = (out.detach().cpu() - want).abs() / scale
        node_err = err.amax(dim=1) if err.dim() == 4 else err
        bad = int((node_err > 1e-4).sum())
        flips += bad
        assert bad <= max(1, node_err.numel() // 200), f"{tag}: {bad} of {node_err.numel()} nodes differ"
        assert node_err.median().item() < 1e-5, f"{tag}: median node error {node_err.median():.2e}"
        if not need_grad or bad:
            continue
        g = det_tensor("pvig.g." + tag, tuple(want.shape))
        mod.zero_grad(set_to_none=True)
        (out * g.to(dev)).sum().backward()
        xr = xin.clone().requires_grad_(True)
        for p in params.values():
            p.grad = None
        (ref_stages[tag](xr) * g).sum().backward()
        _close(xg.grad, xr.grad, 2e-3, f"{tag}: d input")
        checked = 0
        stage_scale = max(p.grad.abs().max().item() for p in params.values() if p.grad is not None)
        for name, p in mod.named_parameters():
            ref_p = params[name]
            if p.grad is None or ref_p.grad is None:
                assert (p.grad is None or not p.grad.any()) and (ref_p.grad is None or not ref_p.grad.any()), name
                continue
            if ref_p.grad.abs().max().item() < 1e-3 * stage_scale:
                # rounding noise on both sides: a conv bias under train-mode BN has an exactly-zero true gradient
                assert p.grad.abs().max().item() < 1e-2 * stage_scale, f"{tag}: d {name} should be ~0"
                continue
            _close(p.grad, ref_p.grad, 5e-3, f"{tag}: d {name}")
            checked += 1
        assert checked >= 2, f"{tag}: only {checked} parameter gradients compared"
    assert flips <= 4, f"{flips} nodes differ across the network on oracle-exact inputs"


def test_knn_vs_reference_fixture(dev):
    """k-NN indices of the HIP kernel equal the reference's on every stable row (bit-exact), C oracle everywhere."""
    from graphecho_amd.models.vig import DenseDilatedKnnGraph
    from oracle.knn import knn_graph
    from oracle.weights import det_tensor

    g = _gold("knn")
    for tag, (B, C, N, M, d) in {"n64_m64": (2, 256, 64, 64, 1), "self256": (2, 64, 256, None, 1),
                                 "n1024_m256_d2": (1, 128, 1024, 256, 2),
                                 "n4096_m256": (1, 256, 4096, 256, 1)}.items():      # config 2's p2 graph (survey F3)
        x = det_tensor(f"knn.{tag}.x", (B, C, N, 1))
        y = None if M is None else det_tensor(f"knn.{tag}.y", (B, C, M, 1))
        idx = DenseDilatedKnnGraph(9, d)(x.to(dev), None if y is None else y.to(dev)).cpu().numpy()
        ref, stable = g[tag + "_idx"].astype(np.int64), g[tag + "_stable"]
        assert np.array_equal(idx[1], ref[1])
        assert np.array_equal(idx[0][stable], ref[0][stable])
        assert np.array_equal(idx, knn_graph(x.numpy(), None if y is None else y.numpy(), 9, d))


def test_small_modules_vs_reference_fixture(dev):
    from graphecho_amd.models.affinity_layer import Affinity
    from graphecho_amd.models.transformer import MultiHeadAttention
    from graphecho_amd.utils.sinkhorn_distance import SinkhornDistance
    from oracle.weights import det_tensor, fill_state_dict

    g = _gold("small_ops")
    mha = MultiHeadAttention(256, 1, dropout=0.0, version="v2")
    mha.load_state_dict(fill_state_dict(mha.state_dict(), seed=4))
    mha = mha.to(dev)
    kv, q = det_tensor("mha.kv", (70, 256)).to(dev), det_tensor("mha.q", (50, 256)).to(dev)
    o, a = mha(kv, kv, q)
    _close(o, g["mha_out"], 1e-3, "mha out")
    _close(a[::5, ::5], g["mha_att"], 1e-3, "mha attention")
    aff = Affinity(256)
    aff.load_state_dict(fill_state_dict(aff.state_dict(), seed=5))
    aff = aff.to(dev)
    M = aff(det_tensor("aff.x", (37, 256)).to(dev), det_tensor("aff.y", (45, 256)).to(dev))
    _close(M, g["aff_M"], 1e-3, "affinity")
    sd = SinkhornDistance(eps=0.1, max_iter=5, reduction="mean")
    c3, p3, C3 = sd(det_tensor("sd.x", (2, 64, 256), "uniform").to(dev), det_tensor("sd.y", (2, 64, 256), "uniform").to(dev))
    _close(c3, g["sd3_cost"], 1e-3, "sinkhorn cost")
    _close(p3[:, ::4, ::4], g["sd3_pi"], 1e-3, "sinkhorn plan")
    c2, p2, _ = sd(det_tensor("sd2.x", (64, 32), "uniform").to(dev), det_tensor("sd2.y", (50, 32), "uniform").to(dev))
    _close(c2, g["sd2_cost"], 1e-3, "sinkhorn 2-D cost")
    _close(p2[::4, ::4], g["sd2_pi"], 1e-3, "sinkhorn 2-D plan")


@pytest.mark.parametrize("cluster", [0, 1])
def test_gmodule_vs_reference_fixture(dev, cluster):
    from graphecho_amd.models.graph_matching import GModule
    from oracle.weights import det_tensor, fill_state_dict, rect_masks

    g = _gold(f"gmodule_cluster{cluster}")
    gm = GModule(256, 4, dev)
    gm.load_state_dict(fill_state_dict(gm.state_dict(), seed=6))
    _no_dropout(gm)
    gm = gm.to(dev).train()
    gm.with_cluster_update = bool(cluster)
    sizes = (64, 32, 16, 8)
    fs = [det_tensor(f"gm.fs{l}", (2, 256, s, s)).to(dev).requires_grad_(True) for l, s in enumerate(sizes)]
    ft = [det_tensor(f"gm.ft{l}", (2, 256, s, s)).to(dev).requires_grad_(True) for l, s in enumerate(sizes)]
    tgt, sm = rect_masks(2, 4, 256, 256, seed=1).to(dev), rect_masks(2, 4, 256, 256, seed=2).to(dev)
    _, (n1, n2), losses = gm(None, (fs, ft), targets=tgt, score_maps=sm)
    sum(losses.values()).backward()
    assert [len(n1), len(n2)] == list(g["n_nodes"])
    _close(n1[::7, ::16], g["n1"], 1e-3, "nodes_1")
    _close(n2[::7, ::16], g["n2"], 1e-3, "nodes_2")
    for k in ("dis_loss", "node_loss", "mat_loss_aff", "mat_loss_qu"):
        _close(losses[k], g[k], 1e-3, k)
    _close(gm.sr_seed, g["sr_seed"], 1e-3, "sr_seed")
    _close(gm.tg_seed, g["tg_seed"], 1e-3, "tg_seed")
    _close(fs[0].grad[:, ::32, ::8, ::8], g["g_fs0"], 1e-2, "d p2")
    _close(gm.node_affinity.fc_M[0].weight.grad[:8, :8], g["g_aff"], 1e-2, "d fc_M.0")


@pytest.mark.parametrize("method", ["node_discriminate", "sinkhorn_distance"])
def test_tgcn_vs_reference_fixture(dev, method):
    from graphecho_amd.models.TGCN import TGCN
    from graphecho_amd.utils.sinkhorn_distance import SinkhornDistance
    from oracle.weights import det_tensor, fill_state_dict

    g = _gold("tgcn_" + method)
    m = TGCN(256, 256, (3, 8, 8), 10, 10, transport_method=method)
    m.load_state_dict(fill_state_dict(m.state_dict(), seed=7))
    _no_dropout(m)
    m = m.to(dev).train()
    feats = [det_tensor(f"tgcn.f{l}", (2, 3, 256, s, s)).to(dev) for l, s in enumerate((64, 32, 16, 8))]
    nodes = (det_tensor("tgcn.ns", (33, 256)).to(dev), det_tensor("tgcn.nt", (34, 256)).to(dev))
    graphs = []
    attend = m.grapher.attend          # the recurrent half of a time step (the pooling + MLP half runs batched over the steps)

    def probe(*a, **k):
        out = attend(*a, **k)
        graphs.append(out[0].detach())
        return out

    m.grapher.attend = probe
    upd = (torch.zeros(1, dtype=torch.long, device=dev), torch.zeros(1, dtype=torch.long, device=dev))
    losses = m(feats, nodes, SinkhornDistance(eps=0.1, max_iter=5, reduction="mean"), torch.nn.CrossEntropyLoss(),
               upd, r=[8, 4, 2, 1])
    del m.grapher.attend
    sum(losses.values()).backward()
    _close(graphs[0][:, ::16, ::4], g["graph0"], 1e-3, "graph after step 0 (all-ties k-NN)")
    _close(graphs[-1][:, ::16, ::4], g["graph"], 1e-3, "current_graph")
    for k, v in losses.items():
        _close(v, g[k], 1e-3, k)
    _close(m.pos_embed.grad[:, 0, ::32], g["g_pos"], 1e-2, "d pos_embed")
    _close(m.grapher.MLP[0].weight.grad[:8, :8, 0, 0], g["g_mlp"], 1e-2, "d MLP.0")


def test_tgcn_backward_per_time_step_on_oracle_inputs(dev):
    """The 16-step recurrence of config 5's TGCN (TGCN.py:224-285; clip 16 x 8 x 8, two clips), backward checked PER TIME
    STEP on oracle-exact inputs -- the pattern of test_pvig_vs_oracle_stage_by_stage.  End to end, two correct fp32
    implementations differ by ~1e-1 on the gradient probes downstream of the recurrence (every step rebuilds a k-NN graph
    from the previous step's output; a flipped 9th neighbour re-routes gradient -- test_temporal_step_c5_vs_reference_
    fixture bounds those probes at 0.25).  Here the oracle runs the whole recurrence once (fp32), which fixes every step's
    inputs (pyramid slices, previous graph) and the gradient arriving at every step's output; each step is then run on
    its own by the HIP module and by the oracle in fp32 and fp64 on those SAME inputs and upstream gradient: outputs
    1e-3, every input / parameter gradient of the step <= 2e-2 L2-relative (plus the fp32-vs-fp64 distance of the oracle
    itself where THAT already contains a flipped neighbour) and as accurate as the fp32 oracle against fp64."""
    from graphecho_amd.models.TGCN import TGCN
    from oracle.tgcn import grapher_step
    from oracle.weights import det_tensor, fill_state_dict

    L, B, rs = 16, 2, [8, 4, 2, 1]
    m = TGCN(256, 256, (L, 8, 8), 10, 10, transport_method="sinkhorn_distance")
    sd = fill_state_dict(m.state_dict(), seed=11)
    m.load_state_dict(sd)
    _no_dropout(m)
    m = m.to(dev).train()
    feats = [det_tensor(f"tgcnstep.f{l}", (B, L, 256, s, s)) for l, s in enumerate((64, 32, 16, 8))]
    wout = det_tensor("tgcnstep.w", (B, 256, 64))
    pnames = ["grapher.MLP.0.weight", "grapher.MLP.0.bias", "grapher.MLP.1.weight", "grapher.MLP.1.bias",
              "grapher.MLP.4.weight", "grapher.MLP.4.bias", "grapher.gconv.nn.0.weight", "grapher.gconv.nn.0.bias"]

    def oracle_params(dtype):
        return {k: (v.to(dtype).clone().requires_grad_(k in pnames or k == "pos_embed") if v.is_floating_point()
                    else v.clone()) for k, v in sd.items()}

    # 1. the whole recurrence once: every step's inputs and the gradient arriving at its output
    p = oracle_params(torch.float32)
    hidden = torch.zeros(B, 256, 64)
    hiddens, outs = [], []
    for i in range(L):
        hiddens.append(hidden)
        hidden, _, _ = grapher_step(p, [f[:, i] for f in feats], rs, hidden, p["pos_embed"][i], True)
        hidden.retain_grad()
        outs.append(hidden)
    (outs[-1] * wout).sum().backward()
    ups = [o.grad.clone() for o in outs]
    assert all(torch.isfinite(u).all() and u.abs().max() > 0 for u in ups)

    def oracle_step(i, dtype):
        q = oracle_params(dtype)
        fi = [f[:, i].to(dtype).clone().requires_grad_(True) for f in feats]
        h = hiddens[i].detach().to(dtype).clone().requires_grad_(i > 0)
        out, _, _ = grapher_step(q, fi, rs, h, q["pos_embed"][i], True)
        out.backward(ups[i].to(dtype))
        grads = {f"feat{l}": t.grad for l, t in enumerate(fi)}
        grads["hidden"] = h.grad if i > 0 else None
        grads["pos"] = q["pos_embed"].grad[i]
        grads.update({k: q[k].grad for k in pnames})
        return out.detach(), grads

    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    worst = 0.0
    for i in (0, 1, 2, 7, 15):            # step 0: all-ties k-NN against the zero graph; 15: the step the loss reads
        o32, g32 = oracle_step(i, torch.float32)
        _, g64 = oracle_step(i, torch.float64)
        for mod in m.modules():
            mod.zero_grad(set_to_none=True)
        fi = [f[:, i].to(dev).clone().requires_grad_(True) for f in feats]
        h = hiddens[i].detach().to(dev).clone().requires_grad_(i > 0)
        out, _, _ = m.grapher(fi, rs, h, m.pos_embed[i])
        out.backward(ups[i].to(dev))
        _close(out, o32, 1e-3, f"step {i} graph")
        got = {f"feat{l}": t.grad for l, t in enumerate(fi)}
        got["hidden"] = h.grad if i > 0 else None
        got["pos"] = m.pos_embed.grad[i]
        named = dict(m.named_parameters())
        got.update({k: named[k].grad for k in pnames})
        for k, ref in g32.items():
            if ref is None:
                continue
            if k == "grapher.MLP.0.bias":      # a bias in front of a train-mode BatchNorm: structurally zero gradient
                assert got[k].abs().max().item() <= 1e-4 * max(1.0, ups[i].abs().max().item()), f"step {i} d {k}"
                continue
            e_hip, e_cpu = l2(got[k].cpu(), g64[k]), l2(ref, g64[k])
            e = l2(got[k].cpu(), ref)
            worst = max(worst, e)
            # e_cpu > 1e-3 means the fp32 and fp64 ORACLES already pick a different 9th neighbour somewhere in this step
            # (a near-tie in the k-NN; measured 1.7e-2 on d hidden of step 15): that distance is granted on top
            assert e <= 2e-2 + (e_cpu if e_cpu > 1e-3 else 0.0), f"step {i} d {k}: {e:.2e} against the fp32 oracle"
            assert e_hip <= 10 * e_cpu + 2e-2, f"step {i} d {k}: hip {e_hip:.2e} vs cpu32 {e_cpu:.2e} against fp64"
    print(f"TGCN per-step gradients, worst L2-relative error against the fp32 oracle: {worst:.2e}")


# ----------------------------------------------------------------------------------------------------------
# distributed code paths exercised on ONE GPU (the 2/4/8-GPU runs belong to the driver)
# ----------------------------------------------------------------------------------------------------------
def test_syncbn_moment_merge_equals_full_batch(dev):
    """Per-rank (count, mean, M2) triples gathered as [world][C][3] and merged by ge_bn_finalize reproduce the
    statistics of the concatenated batch (what SyncBatchNorm must compute)."""
    from graphecho_amd._lib import lib, check

    gen = torch.Generator().manual_seed(5)
    B, C, H, W = 6, 20, 12, 12
    x = (torch.randn(B, C, H, W, generator=gen) * 3 + 1).to(dev)
    parts = []
    for lo, hi in ((0, 2), (2, 6)):                       # uneven "ranks"
        xs = x[lo:hi].contiguous()
        nb = lib.ge_bn_num_partials(hi - lo, H * W)
        partial = torch.empty(C * nb * 3, device=dev)
        stats = torch.empty(C * 3, device=dev)
        check(lib.ge_bn_stats_partial(xs.data_ptr(), partial.data_ptr(), hi - lo, C, H * W, None))
        check(lib.ge_bn_finalize(partial.data_ptr(), nb * 3, 3, nb, C, 1e-5, 0.1, stats.data_ptr(), None, None, None,
                                 None, None))
        parts.append(stats)
    gathered = torch.cat(parts)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    check(lib.ge_bn_finalize(gathered.data_ptr(), 3, C * 3, 2, C, 1e-5, 0.1, None, mean.data_ptr(), invstd.data_ptr(),
                             rm.data_ptr(), rv.data_ptr(), None))
    xc = x.cpu()
    assert _relerr(mean, xc.mean((0, 2, 3))) < 1e-5
    assert _relerr(invstd, (xc.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()) < 1e-5
    assert _relerr(rv, 0.9 + 0.1 * xc.var((0, 2, 3), unbiased=True)) < 1e-5


def test_distributed_step_world1_matches_local_step(dev):
    """SyncBN + bucketed async all-reduce + synced 'used' map, forced on at world size 1 over RCCL, must give the
    same update as the plain single-GPU step."""
    import os
    import torch.distributed as dist
    from graphecho_amd import nn as gnn
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    x, m = synthetic_batch(2, 3, 4, 128, dev, 3)
    ref = GraphEchoTrainer(dev, workload="fpn_grapher", image_size=128, seed=1)
    loss_ref = ref.step(x, m)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        tr = GraphEchoTrainer(dev, workload="fpn_grapher", image_size=128, distributed=True, seed=1)
        tr.sync.force = True
        for mod in tr.network.modules():
            if isinstance(mod, gnn.BatchNorm2d):
                mod.force_sync = True
        loss = tr.step(x, m)
        assert abs(loss.item() - loss_ref.item()) < 1e-5 * max(1.0, abs(loss_ref.item()))
        a, b = tr.optimizers["Net"].fp.flat, ref.optimizers["Net"].fp.flat
        assert (a - b).abs().max().item() <= 2.1e-4 and (a - b).abs().mean().item() < 5e-6
        assert len(tr.sync.buckets) >= 4 and all(tr.sync._launched)
    finally:
        if created:
            dist.destroy_process_group()


def test_bucket_exchange_waits_for_weight_gradient_side_stream(dev):
    """buckets launched by mark_complete() -- between autograd calls, when the trainer has already
    reset GF.WGRAD_STREAM -- must still wait for the conv weight-gradient side stream.  A late bucket whose gradients
    are still being written there (delayed by a long sleep) is exchanged in rs_ag mode over a one-rank RCCL group: the
    reduced shard must hold the values the side stream wrote, not what the buffer held before."""
    import os
    import torch.distributed as dist
    from graphecho_amd import functional as GF
    from graphecho_amd.ddp import GradSynchronizer
    from graphecho_amd.optim import FlatSGD

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29575")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        lin = torch.nn.Linear(256, 256).to(dev)
        opt = FlatSGD([lin], lr=0.1)
        sync = GradSynchronizer([opt], mode="rs_ag")
        sync.force = True
        side = torch.cuda.Stream(device=dev)
        sync.side_stream = side
        opt.zero_grad()
        sync.reset()
        torch.cuda.synchronize()
        assert GF.WGRAD_STREAM is None        # the situation of trainer._step_phased at mark_complete()
        with torch.cuda.stream(side):
            torch.cuda._sleep(200_000_000)    # ~0.1 s: the exchange below is enqueued long before this finishes
            opt.fp.grad.fill_(3.0)
        sync.mark_complete([opt])
        assert all(sync._launched)
        sync.finish()
        torch.cuda.synchronize()
        for buf in sync._shard_buf.values():
            assert torch.all(buf[:8] == 3.0) and float(buf.min()) >= 0.0
        n = sum(p.numel() for p in lin.parameters())
        got = torch.cat([b for _, b in sorted(sync._shard_buf.items())])[:n]
        assert torch.all(got == 3.0), "a bucket was exchanged before the side stream's gradients had landed"
    finally:
        if created:
            dist.destroy_process_group()


def test_distributed_temporal_step_world1_over_rccl(dev):
    """The config-5-shaped step (merged FPN pass with three BatchNorm segments, GModule, discriminators, TGCN, Sinkhorn)
    with SyncBN and the gradient synchroniser forced on over a one-rank RCCL group: exercises every collective call of
    the full model set on the real backend (segmented statistics travel in ONE all-gather per layer) and must
    reproduce the local step's loss."""
    import os
    import torch.distributed as dist
    from graphecho_amd import nn as gnn
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    def inputs():
        xs, ms = synthetic_batch(2, 3, 4, 128, dev, 11)
        xt, _ = synthetic_batch(2, 3, 4, 128, dev, 12)

        def clip(seed, t=8):
            f, mk = synthetic_batch(t, 3, 4, 128, dev, seed)
            return (f.reshape(1, t, 3, 128, 128).permute(0, 2, 3, 4, 1).contiguous(),
                    mk.reshape(1, t, 4, 128, 128).permute(0, 2, 3, 4, 1).contiguous())

        cs, cm = clip(13)
        ct, _ = clip(14)
        return xs, ms, xt, {"source": cs, "target": ct, "masks": cm}

    ref = GraphEchoTrainer(dev, workload="temporal", image_size=128, seed=2, clip_len=8)
    ref.graph_model.async_seed_update = False
    loss_ref = ref.step(*inputs())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29573")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        tr = GraphEchoTrainer(dev, workload="temporal", image_size=128, distributed=True, seed=2, clip_len=8)
        tr.graph_model.async_seed_update = False
        tr.sync.force = True
        nsync = 0
        for model in [tr.network, tr.tgcn]:
            for mod in model.modules():
                if isinstance(mod, gnn.BatchNorm2d):
                    mod.force_sync = True
                    nsync += 1
        assert nsync > 50
        loss = tr.step(*inputs())
        assert torch.isfinite(loss) and abs(loss.item() - loss_ref.item()) < 1e-4 * max(1.0, abs(loss_ref.item()))
        assert all(tr.sync._launched) and len(tr.sync.buckets) >= 6
    finally:
        if created:
            dist.destroy_process_group()


def test_distributed_full_step_world1_replays_the_syncbn_backbone(dev):
    """Round 6: under data parallelism over RCCL graphs="auto" replays EVERY static piece -- the SyncBN backbone with its
    all-gathers / all-reduces captured inside the graphs -- not only the collective-free head and discriminators.  One-rank RCCL
    group with SyncBN and the gradient synchroniser forced on: the replayed steps must reproduce the eager distributed steps'
    losses, and the exchange counters must advance by the same 50 + 50 collectives per step."""
    import os
    import torch.distributed as dist
    from graphecho_amd import functional as GF
    from graphecho_amd import nn as gnn
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        xs, ms = synthetic_batch(4, 3, 4, 128, dev, 21)
        xt, _ = synthetic_batch(4, 3, 4, 128, dev, 22)
        runs = {}
        for mode in (False, "auto"):
            tr = GraphEchoTrainer(dev, workload="full", image_size=128, distributed=True, seed=4, graphs=mode)
            tr.graph_model.async_seed_update = False
            tr.sync.force = True
            for mod in tr.network.modules():
                if isinstance(mod, gnn.BatchNorm2d):
                    mod.force_sync = True
            losses, counts = [], []
            for _ in range(6):
                GF.SYNC_BN_STATS[:] = [0, 0, 0]
                losses.append(float(tr.step(xs, ms, xt)))
                counts.append(tuple(GF.SYNC_BN_STATS[:2]))
            runs[mode] = (losses, counts, tr.graphs_in_use())
            del tr
            torch.cuda.synchronize()
        assert runs[False][2] is False and runs["auto"][2] == "all", (runs[False][2], runs["auto"][2])
        for a, b in zip(runs[False][0], runs["auto"][0]):
            assert np.isfinite(a) and abs(a - b) <= 1e-5 * max(1.0, abs(a)), (runs[False][0], runs["auto"][0])
        assert runs[False][1] == runs["auto"][1] and runs["auto"][1][-1][0] > 0, (runs[False][1], runs["auto"][1])
    finally:
        if created:
            dist.destroy_process_group()


def test_syncbn_all_segments_per_launch_equals_per_segment_launches(dev):
    """SyncBN over the merged source + target pass: the layer's launches on either side of the exchange take both segments at
    once (ge_bn_finalize_segs or ge_bn_stats_channel_segs / ge_bn_fwd_channel_segs_sync forward, ge_bn_bwd_reduce_channel_segs /
    ge_bn_bwd_apply_channel_segs backward) -- 3 + 3 launches per layer instead of 5 + 5 (7 + 5 where the convolution left no
    moments).  One step of the distributed trainer (one-rank RCCL group, collectives forced) against GE_SYNCBN_SEGS=0: the kernels
    that merge conv-epilogue moments keep the per-segment kernels' order and expressions; the layers that take their moments from
    x sum in another order, so the step agrees to rounding: loss to 1e-6, running statistics to 1e-5."""
    import os
    import torch.distributed as dist
    from graphecho_amd import functional as GF
    from graphecho_amd import nn as gnn
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    saved = GF.SYNC_BN_SEGS
    try:
        xs, ms = synthetic_batch(4, 3, 4, 128, dev, 31)
        xt, _ = synthetic_batch(4, 3, 4, 128, dev, 32)
        runs = {}
        for segs in (False, True):
            GF.SYNC_BN_SEGS = segs
            tr = GraphEchoTrainer(dev, workload="full", image_size=128, distributed=True, seed=5, graphs=False)
            tr.graph_model.async_seed_update = False
            tr.sync.force = True
            for mod in tr.network.modules():
                if isinstance(mod, gnn.BatchNorm2d):
                    mod.force_sync = True
            loss = float(tr.step(xs, ms, xt))
            torch.cuda.synchronize()
            runs[segs] = (loss, {n: o.fp.grad.clone() for n, o in tr.optimizers.items()},
                          {k: v.clone() for k, v in tr.network.state_dict().items() if "running" in k})
            del tr
        assert abs(runs[False][0] - runs[True][0]) <= 1e-6 * max(1.0, abs(runs[False][0])), (runs[False][0], runs[True][0])
        # (the gradients of this network are not a rounding-stable function of its statistics -- ReLU masks, GModule's discrete
        # node choices: a sanity bound only; the kernels themselves are held to float64 in tests/test_ops_gpu.py)
        for n in runs[False][1]:
            a, b = runs[True][1][n].double(), runs[False][1][n].double()
            assert ((a - b).norm() / b.norm().clamp_min(1e-30)).item() < 0.1, f"gradients of {n}"
        assert len(runs[False][2]) > 50
        for k in runs[False][2]:
            _close(runs[True][2][k], runs[False][2][k], 1e-5, k)
    finally:
        GF.SYNC_BN_SEGS = saved
        if created:
            dist.destroy_process_group()


def test_syncbn_big_layers_merge_and_apply_in_one_launch(dev):
    """SyncBN layers too big for the one-workgroup-per-channel kernels (the stem at 256 x 256): the segments' local finalize is one
    launch (ge_bn_finalize_segs) and every segment's merge of the gathered moments + apply is one (ge_bn_fwd_merge_apply_sync: the
    wave merge order instead of ge_bn_finalize's sequential one -- not the same bits, the same statistics).  Against
    GE_SYNCBN_SEGS=0 on a one-rank RCCL group with the collectives forced: first-step loss to 1e-6, running statistics to 1e-5."""
    import os
    import torch.distributed as dist
    from graphecho_amd import functional as GF
    from graphecho_amd import nn as gnn
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29579")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    saved = GF.SYNC_BN_SEGS
    try:
        xs, ms = synthetic_batch(4, 3, 4, 256, dev, 41)
        xt, _ = synthetic_batch(4, 3, 4, 256, dev, 42)
        runs = {}
        for segs in (False, True):
            GF.SYNC_BN_SEGS = segs
            tr = GraphEchoTrainer(dev, workload="full", image_size=256, distributed=True, seed=6, graphs=False)
            tr.graph_model.async_seed_update = False
            tr.sync.force = True
            for mod in tr.network.modules():
                if isinstance(mod, gnn.BatchNorm2d):
                    mod.force_sync = True
            loss = float(tr.step(xs, ms, xt))
            torch.cuda.synchronize()
            runs[segs] = (loss, {k: v.clone() for k, v in tr.network.state_dict().items() if "running" in k})
            del tr
        assert abs(runs[False][0] - runs[True][0]) <= 1e-6 * max(1.0, abs(runs[False][0])), (runs[False][0], runs[True][0])
        for k in runs[False][1]:
            _close(runs[True][1][k], runs[False][1][k], 1e-5, k)
    finally:
        GF.SYNC_BN_SEGS = saved
        if created:
            dist.destroy_process_group()


def test_ddp_world2_gloo_on_one_gpu(dev, tmp_path):
    """Two ranks (gloo, both on cuda:0) run the real distributed trainer: SyncBN all-gather/all-reduce, bucketed
    gradient all-reduce from the autograd hooks, flat optimizers.  Replicas must stay bit-identical, and the SyncBN
    running statistics must equal those of a single process that sees the whole batch."""
    import subprocess
    import sys
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "ddp_gpu_worker.py")
    port = str(29600 + os.getpid() % 300)
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert torch.equal(a["flat"], b["flat"]) and torch.equal(a["gflat"], b["gflat"]), "replicas diverged"
    assert torch.equal(a["rm"], b["rm"]) and torch.equal(a["rv"], b["rv"]) and a["nbt"] == b["nbt"] == 2
    assert a["buckets"] >= 4
    assert all(np.isfinite(a["losses"])) and all(np.isfinite(b["losses"]))
    # single process, whole batch: the first BatchNorm sees the same conv1 weights and, through SyncBN, the same
    # batch statistics on the first step => the same running statistics after it
    x, m = synthetic_batch(4, 3, 4, 128, dev, 7)
    ref = GraphEchoTrainer(dev, workload="fpn_grapher", image_size=128, seed=1)
    ref.step(x, m)
    bn0 = ref.network.back_bone.bn1
    _close(a["rm1"].to(dev), bn0.running_mean, 1e-5, "syncbn running_mean vs whole-batch")
    _close(a["rv1"].to(dev), bn0.running_var, 1e-4, "syncbn running_var vs whole-batch")


def test_train_loop_synthetic_raw_data(dev, tmp_path):
    """graphecho_amd.train.run: raw uint8 batches -> GPU formatting -> steps -> validation Dice -> checkpoint in the
    reference's format, reloadable into a fresh FPN (and by load())."""
    from graphecho_amd import train as gtrain
    from graphecho_amd.models.fpnseg import FPN

    cfg = {"train": {"num_epochs": 2, "batch_size": 2, "save_dir": str(tmp_path), "spatial_size": 160, "crop_size": 128,
                     "graph_matching": False, "discriminator": False}}
    src = gtrain.SyntheticRawSet(2, 2, 3, 4, hw=(150, 200), seed=1, device=dev)
    val = gtrain.SyntheticRawSet(1, 2, 3, 4, hw=(150, 200), seed=3, device=dev)
    logs = []
    trainer, hist = gtrain.run(cfg, src, None, val, device=dev, log=logs.append)
    assert len(hist) == 2 and all(np.isfinite(h["loss"]) for h in hist) and len(hist[0]["dice"]) == 4
    assert all(0.0 <= d <= 1.0 for d in hist[1]["dice"])
    assert sorted(os.listdir(tmp_path)) == ["latest.ckpt", "net_00000.pth", "net_00001.pth"]
    assert open(tmp_path / "latest.ckpt").read() == "00001\n"      # the id only: train_camus_echo.py:449-459,487
    sd = torch.load(tmp_path / "net_00001.pth")["network"]
    fresh = FPN([2, 4, 23, 3], 4, 3)
    fresh.load_state_dict(sd)       # reference checkpoint format: {'network': state_dict}
    for k, v in trainer.network.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k


def test_training_learns_a_synthetic_segmentation_task(dev, tmp_path):
    """End-to-end sanity of forward + losses + backward + Adam + validation: on frames whose brightness depends on the
    class (speckle/3 + 50 grey levels per class id) the validation Dice of every class goes from ~0.1 to > 0.7 within 75
    steps; with pure-noise frames it stays at the class prior."""
    from graphecho_amd import train as gtrain

    cfg = {"train": {"num_epochs": 3, "batch_size": 8, "save_dir": str(tmp_path), "spatial_size": 144, "crop_size": 128,
                     "graph_matching": False, "discriminator": False, "seg_loss": "cardiac"}}
    src = gtrain.SyntheticRawSet(25, 8, 3, 4, hw=(150, 200), seed=1, device=dev, contrast=50)
    val = gtrain.SyntheticRawSet(2, 8, 3, 4, hw=(150, 200), seed=3, device=dev, contrast=50)
    _, hist = gtrain.run(cfg, src, None, val, device=dev, log=lambda s: None)
    assert hist[-1]["loss"] < hist[0]["loss"]
    assert min(hist[-1]["dice"]) > 0.7, hist[-1]["dice"]
    assert min(hist[-1]["dice"][1:]) > min(hist[0]["dice"][1:]) + 0.4, (hist[0]["dice"], hist[-1]["dice"])


@pytest.mark.parametrize("bb,hw,sizes,gtol,wino", [("resnet", 128, (3, 5), 5e-3, False), ("resnet", 128, (3, 5), 3e-2, True),
                                                   ("resnet", 64, (2, 3, 1), 5e-2, True), ("VGG16", 64, (1, 2), 5e-2, True)])
def test_bn_segments_equal_separate_passes(dev, bb, hw, sizes, gtol, wino, monkeypatch):
    """One FPN pass over [a; b; c] under GF.bn_segments == the passes net(a), net(b), net(c) in that order: logits,
    pyramids, running statistics (sequential momentum updates), num_batches_tracked and every parameter gradient.
    At hw = 64 the deepest maps are 2x2: segment boundaries fall inside a conv-epilogue statistics run there and the
    layer falls back to its own statistics pass (maps of 1x1 with 2-sample segments are left out: BatchNorm over two
    values is a sign function, any two fp32 evaluation orders disagree on it)."""
    import copy
    from graphecho_amd import functional as GF
    from graphecho_amd.models.fpnseg import FPN

    # wino = False: one conv algorithm for every batch size, the tight gradient bound.  With the Winograd route on, the merged
    # pass (8 frames) and the separate ones (3 / 5 frames) take different kernels for the same layer (the routing plan looks at
    # the grid size): two valid fp32 roundings that 50 train-mode BatchNorm layers amplify to 1e-2 on the gradient (measured);
    # a wrong segment would be O(1) off on the logits already.
    monkeypatch.setattr(GF, "WINOGRAD", wino)
    torch.manual_seed(0)
    net_a = FPN([2, 4, 23, 3], 3, 3, back_bone=bb).to(dev).train()
    net_b = copy.deepcopy(net_a)
    gen = torch.Generator().manual_seed(5)
    xs = [torch.rand(n, 3, hw, hw, generator=gen).to(dev) for n in sizes]
    gs = [torch.randn(n, 3, hw, hw, generator=gen).to(dev) for n in sizes]
    sep = [net_a(x) for x in xs]
    sum((o[0] * g).sum() + sum(p.square().mean() for p in o[1]) for o, g in zip(sep, gs)).backward()
    with GF.bn_segments(sizes):
        logits, pyr = net_b(torch.cat(xs))
    parts = torch.split(logits, sizes)
    pparts = [torch.split(p, sizes) for p in pyr]
    sum((parts[i] * gs[i]).sum() + sum(pp[i].square().mean() for pp in pparts) for i in range(len(sizes))).backward()
    for i, o in enumerate(sep):
        # a 1-sample pass normalises 2x2 maps over four values: rounding of the moments (conv-epilogue runs vs a
        # separate statistics pass) is amplified there, hence 5e-4 and not 1e-5; a wrong segment would be O(1) off
        _close(parts[i], o[0], 5e-4, f"logits of pass {i}")
        for lvl in range(4):
            _close(pparts[lvl][i], o[1][lvl], 5e-4, f"p{lvl + 2} of pass {i}")
    sa, sb = net_a.state_dict(), net_b.state_dict()
    for k in sa:
        if "running" in k:
            _close(sb[k], sa[k], 1e-4, k)
        elif "num_batches" in k:
            assert int(sa[k]) == int(sb[k]) == len(sizes), k
    ga = torch.cat([p.grad.reshape(-1) for p in net_a.parameters()])
    gb = torch.cat([p.grad.reshape(-1) for p in net_b.parameters()])
    # gradients through 1-sample / 1x1-map BatchNorms are ill-conditioned (norms ~1e7): tight only on the regular case
    assert ((ga - gb).norm() / ga.norm()).item() < gtol
    with pytest.raises(RuntimeError):
        with GF.bn_segments((1, 1)):
            net_b(torch.cat(xs))


def test_train_loop_from_camus_tree(dev, tmp_path):
    """On-disk CAMUS-shaped tree (.mhd/.zraw, frames of different sizes) -> CamusSet -> RawBatches -> GPU formatting ->
    training steps + validation (SURVEY.md 8f rank 3 in front of rank 2 and 1)."""
    from graphecho_amd import train as gtrain
    from graphecho_amd.datasets import CamusSet, RawBatches, write_mhd

    rng = np.random.default_rng(0)
    for i in range(12):
        pid = f"patient{i:04d}"
        d = tmp_path / "camus" / "training" / pid
        d.mkdir(parents=True)
        h, w = 150 + 3 * i, 180 + 5 * i
        lab = np.zeros((1, h, w), np.uint8)
        lab[0, 20:90, 30:100] = 1
        lab[0, 95:140, 60:150] = 3
        lab[0, 60:80, 110:160] = 2                     # myocardium id: present in the file, not a training class
        write_mhd(str(d / f"{pid}_2CH_ED.mhd"), rng.integers(0, 256, (1, h, w)).astype(np.uint8), compressed=True)
        write_mhd(str(d / f"{pid}_2CH_ED_gt.mhd"), lab)
    root = str(tmp_path / "camus")
    train_set = CamusSet(root, "2CH_ED", "2CH_ED_gt", "train")
    valid_set = CamusSet(root, "2CH_ED", "2CH_ED_gt", "valid")
    assert len(train_set) == 10 and len(valid_set) == 1
    cfg = {"train": {"num_epochs": 1, "batch_size": 4, "save_dir": str(tmp_path / "ckpt"), "spatial_size": 144,
                     "crop_size": 128, "graph_matching": False, "discriminator": False, "in_channel": 1,
                     "class_values": train_set.class_values}}
    src = RawBatches(train_set, 4, dev, shuffle=True, drop_last=True, seed=1)
    frames, labels = next(iter(src))
    assert len(frames) == 4 and frames[0].dtype == torch.uint8 and frames[0].is_cuda and frames[0].dim() == 4
    x, m = gtrain._format(frames, labels, cfg, False, None)
    assert x.shape == (4, 1, 128, 128) and m.shape == (4, 2, 128, 128)
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0 and set(m.unique().tolist()) <= {0.0, 1.0}
    assert 0.05 < float(m[:, 0].mean()) < 0.6 and 0.02 < float(m[:, 1].mean()) < 0.6     # LV and LA planes populated
    trainer, hist = gtrain.run(cfg, src, None, RawBatches(valid_set, 4, dev), device=dev, log=lambda s: None)
    assert len(hist) == 1 and np.isfinite(hist[0]["loss"]) and len(hist[0]["dice"]) == 2


@pytest.mark.parametrize("prec", ["f16", "f16s"])
def test_fpn_f16_conv_path_tracks_fp32(dev, prec):
    """BASELINE config 5's conv path (fp16 MFMA inputs, fp32 accumulation; "f16": fp32 storage, "f16s": the covered 3x3
    convs of the ResNet FPN through the blocked-fp16 kernels of graphecho_amd/half.py) on the whole FPN.  A random-init
    FPN in train-mode BN on noise frames is ill-conditioned (an input perturbation of fp16-rounding size, 2^-11
    relative, moves the fp32 logits by ~14 %), so the yardstick is that conditioning: the fp16 path must deviate no
    more than such a perturbation does on the fp32 path.  (Per-layer exactness on fp16-rounded operands is checked in
    test_ops_gpu.py::test_conv2d_f16_mfma_path.)"""
    from graphecho_amd import functional as GF
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    x, m = synthetic_batch(4, 3, 4, 128, dev, 5)
    ref = GraphEchoTrainer(dev, workload="fpn", image_size=128, seed=3)
    low = GraphEchoTrainer(dev, workload="fpn", image_size=128, seed=3, conv_precision=prec)
    sd = {k: v.clone() for k, v in ref.network.state_dict().items()}
    gen = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():
        l32, p32 = ref.network(x)
        lp, pp = ref.network(x * (1 + 2.0 ** -11 * torch.randn(x.shape, device=dev, generator=gen)))
        GF.CONV_PRECISION = "f16"
        GF.ACT_STORAGE = "f16" if prec == "f16s" else "f32"
        try:
            l16, p16 = low.network(x)
        finally:
            GF.CONV_PRECISION = "f32"
            GF.ACT_STORAGE = "f32"
    rms = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert 1e-6 < rms(l16, l32) <= 1.2 * rms(lp, l32), (rms(l16, l32), rms(lp, l32))
    for a, b, c in zip(p16, p32, pp):
        assert rms(a, b) <= 1.2 * rms(c, b), (rms(a, b), rms(c, b))
    ref.network.load_state_dict(sd)
    low.network.load_state_dict(sd)
    loss32, loss16 = ref.step(x, m), low.step(x, m)
    assert GF.CONV_PRECISION == "f32" and GF.ACT_STORAGE == "f32"
    assert torch.isfinite(loss16) and abs(loss16.item() - loss32.item()) < 0.1 * abs(loss32.item())
    w32, w16 = ref.optimizers["Net"].fp.flat, low.optimizers["Net"].fp.flat
    assert torch.isfinite(w16).all() and (w16 - w32).abs().max().item() <= 2.1e-4   # Adam's first step moves <= lr


# ----------------------------------------------------------------------------------------------------------
# BASELINE.json's configurations at their own sizes, against fixtures the reference produced
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,cin,nc", [("resnet_c3_n4_256", 3, 4), ("resnet_c1_n3_256", 1, 3)])
def test_fpn_256_vs_reference_fixture(dev, tag, cin, nc):
    """Config 1 exactly (2 x 3 x 256 x 256, 4 classes; and the 1-channel / 3-class variant the trainers use): HIP FPN
    forward + backward against what the reference computed -- logits / pyramid / loss within 1e-3 (north_star), the
    stored gradient probes within 5e-3 of their scale (train-mode BN chain; test_fpn_eval_mode_gradients holds the
    well-conditioned case to 1e-3)."""
    from graphecho_amd import functional as GF
    from graphecho_amd.models.fpnseg import FPN
    from oracle.weights import det_tensor, fill_state_dict

    g = _gold("fpn_" + tag)
    net = FPN([2, 4, 23, 3], nc, cin, back_bone="resnet")
    assert list(net.state_dict().keys()) == list(g["keys"])
    net.load_state_dict(fill_state_dict(net.state_dict(), seed=1))
    net = net.to(dev).train()
    x = det_tensor(f"{tag}.x", (2, cin, 256, 256), "uniform").to(dev).requires_grad_(True)
    t = (det_tensor(f"{tag}.t", (2, nc, 256, 256), "uniform") > 0.6).float().to(dev)
    logits, pyr = net(x)
    loss = GF.dice_loss(logits, t) + GF.bce_with_logits(logits, t)
    loss.backward()
    _close(logits[:, :, ::8, ::8], g["logits"], 1e-3, "logits")
    _close(pyr[3], g["p5"], 1e-3, "p5")
    _close(pyr[0].mean((0, 2, 3)), g["p2_mean"], 1e-3, "p2 mean")
    _close(pyr[1].mean((0, 2, 3)), g["p3_mean"], 1e-3, "p3 mean")
    _close(pyr[2].std((0, 2, 3)), g["p4_std"], 1e-3, "p4 std")
    _close(loss, g["loss"], 1e-4, "loss")
    _close(net.conv3.weight.grad, g["g_conv3"], 5e-3, "d conv3")
    _close(net.smooth3.weight.grad[:8, :8], g["g_smooth3"], 5e-3, "d smooth3")
    # the input gradient runs back through all 50 train-mode BN layers: two correct fp32 implementations differ by a few
    # per cent there (test_fpn_forward_backward_vs_oracle measures the CPU oracle against fp64); L2-relative bound
    gx, rx = x.grad[:, :, ::16, ::16].detach().cpu().double(), torch.as_tensor(g["g_x"]).double()
    rel = ((gx - rx).norm() / rx.norm()).item()
    assert rel < GX_TOL, f"input gradient: L2-relative error {rel:.3e}"
    _close(net.state_dict()["back_bone.bn1.running_mean"], g["running_mean0"], 1e-3, "running mean")


def test_fpn_eval_mode_gradients(dev):
    """BatchNorm in eval mode (running statistics: an affine map per channel) makes the FPN a well-conditioned function,
    so EVERY weight gradient (and the input gradient) is held to 1e-3 against the fp32 oracle -- no error-budget
    argument.  The norm is L2-relative per tensor.  (What remains ill-conditioned even in eval mode are ReLU kinks: a
    GroupNorm output within 1e-6 of zero whose incoming gradient is large flips its mask between two correct fp32
    implementations.  On the VGG16 variant at 64 x 64 one such element moves the p2-level gradient by 9e-3 -- traced:
    the HIP GroupNorm backward equals torch's to 1e-7 on identical inputs -- which is why this test uses the ResNet
    variant, where no kink happens to be hit.)"""
    from graphecho_amd import functional as GF
    from graphecho_amd.models.fpnseg import FPN
    from oracle.fpn import fpn_forward
    from oracle.misc import seg_loss_cardiac
    from oracle.weights import det_tensor, fill_state_dict

    bb, cin, nc, hw = "resnet", 3, 4, 128
    net = FPN([2, 4, 23, 3], nc, cin, back_bone=bb)
    sd = fill_state_dict(net.state_dict(), seed=1)
    net.load_state_dict(sd)
    x = det_tensor("evalgrad.x", (2, cin, hw, hw), "uniform")
    t = (det_tensor("evalgrad.t", (2, nc, hw, hw), "uniform") > 0.6).float()
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
              for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref_logits, _ = fpn_forward(params, xr, False)
    seg_loss_cardiac(ref_logits, t).backward()
    net = net.to(dev).eval()
    xg = x.to(dev).requires_grad_(True)
    logits, _ = net(xg)
    (GF.dice_loss(logits, t.to(dev)) + GF.bce_with_logits(logits, t.to(dev))).backward()
    l2 = lambda a, b: ((a.detach().cpu().double() - b.detach().double()).norm() / b.detach().double().norm()).item()
    assert _relerr(logits, ref_logits) < 1e-3
    assert l2(xg.grad, xr.grad) < 1e-3
    wscale = max(params[n].grad.abs().max().item() for n, _ in net.named_parameters())
    for n, p in net.named_parameters():
        r = params[n].grad
        if r.abs().max().item() < 1e-6 * wscale:      # structurally zero (a bias in front of a per-channel GroupNorm)
            assert p.grad.abs().max().item() < 1e-5 * wscale, n
            continue
        assert l2(p.grad, r) < 1e-3, (n, l2(p.grad, r))


def _full_step_setup(tag, nb, hw):
    from helpers.step_setup import full_step_setup
    return full_step_setup(tag, nb, hw)


@pytest.mark.parametrize("tag,nb,hw", [("128", 2, 128), ("256", 8, 256)])
def test_full_step_c3_vs_reference_fixture(dev, tag, nb, hw):
    """Config 3 ("256": exactly -- source 8 + target 8 frames of 3 x 256 x 256): two optimisation steps of the HIP
    trainer (FPN on both domains as one merged pass, seg loss, score maps, GModule with its hallucination branch fed the
    fixture's noise stream and the scikit-learn seed update, four Discriminators, backward, Adam / SGD) -- every loss
    term of both steps within 1e-3 of what the reference's own modules computed (survey F8), gradients probes, seed
    banks, running statistics and the weights after the two steps."""
    from graphecho_amd.trainer import GraphEchoTrainer

    g = _gold("step_c3_" + tag)
    fpn_sd, gm_sd, dis_sd, xs, xt, masks, noise_fn, draws = _full_step_setup(tag, nb, hw)
    tr = GraphEchoTrainer(dev, workload="full", image_size=hw, seg_loss="cardiac", seed=0)
    _no_dropout(tr.graph_model)
    tr.load_states({"Net": fpn_sd, "Graph": gm_sd, **{"Dis_P" + k[1]: v for k, v in dis_sd.items()}})
    tr.graph_model.noise_fn = noise_fn
    xs, xt, masks = xs.to(dev), xt.to(dev), masks.to(dev)
    # step 0 (the fixture's weights): 1e-3, north_star's tolerance.  Step 1 runs on weights one Adam / SGD step later:
    # Adam's first update is lr * g / (|g| + 1e-8), i.e. lr * sign(g) -- every weight whose gradient is at rounding-noise
    # level moves by +-lr in a direction two fp32 implementations need not agree on, so the second step's losses are
    # held to 5e-3 (measured 3.4e-3 at 128 x 128, 1.1e-3 at 256 x 256).
    for step in range(2):
        total = tr.step(xs, masks, xt)
        tol = 1e-3 if step == 0 else 5e-3
        for k in g["loss_keys"]:
            _close(tr.losses[str(k)], g[f"s{step}.{k}"], tol, f"step {step} {k}")
        _close(total, g[f"s{step}.total"], tol, f"step {step} total")
        # seed banks: momentum update from class means after scikit-learn spectral clustering -- a discrete filter
        # (which nodes count towards the mean), so only the step on the fixture's exact weights is comparable
        if step == 0:
            _close(tr.graph_model.sr_seed, g["s0.sr_seed"], 1e-3, "step 0 sr_seed")
            _close(tr.graph_model.tg_seed, g["s0.tg_seed"], 1e-3, "step 0 tg_seed")
    assert torch.isfinite(tr.graph_model.sr_seed).all() and torch.isfinite(tr.graph_model.tg_seed).all()
    assert len(draws) == int(g["noise_draws"])
    sd = tr.network.state_dict()
    _close(sd["back_bone.bn1.running_mean"], g["running_mean0"], 1e-3, "running mean after 4 FPN passes")
    d = (sd["conv3.weight"].cpu() - torch.as_tensor(g["conv3_after"])).abs()
    assert d.max().item() <= 4.2e-4 and d.mean().item() < 2e-5, (d.max().item(), d.mean().item())



@pytest.mark.parametrize("tag,nb,hw", [("128", 2, 128), ("256", 8, 256)])
def test_reference_loop_body_with_stock_optimizers_after_install(dev, tag, nb, hw):
    """The drop-in scenario itself (INTEGRATION.md recipe A, north_star: "train_camus_echo.py can drop them in
    unchanged"): after ``install_as_reference_modules()`` the modules are imported under the REFERENCE'S names and driven
    the way the reference's loop drives them (train_camus_echo.py:205-303 as tools/gen_golden.py:step_case composes it) --
    two FPN calls, ``utils.losses.DiceLoss`` + ``nn.BCEWithLogitsLoss``, score maps, GModule, four Discriminators x 0.1,
    ``zero_grad`` / one ``backward`` / ``step`` of STOCK ``torch.optim.Adam`` and ``torch.optim.SGD`` over
    ``module.parameters()``.  No trainer of this package, no flat buffers, no fused optimizers, no merged passes: every
    loss term of two steps against what the reference's own modules computed (config 3's fixture)."""
    import sys
    import graphecho_amd
    from helpers.step_setup import full_step_setup

    g = _gold("step_c3_" + tag)
    fpn_sd, gm_sd, dis_sd, xs, xt, masks, noise_fn, draws = full_step_setup(tag, nb, hw)
    saved = {k: sys.modules.pop(k) for k in [k for k in sys.modules if k.partition(".")[0] in ("models", "utils")]}
    graphecho_amd.install_as_reference_modules()
    try:
        from models.fpnseg import FPN, Discriminator           # the reference's import lines (train_camus_echo.py:33-35)
        from models.graph_matching import GModule
        from utils.losses import DiceLoss

        assert FPN.__module__.startswith("graphecho_amd.")
        net = FPN([2, 4, 23, 3], 4, 3, back_bone="resnet").to(dev)
        net.load_state_dict(fpn_sd)
        gm = GModule(256, 4, dev).to(dev)
        gm.load_state_dict(gm_sd)
        _no_dropout(gm)
        gm.noise_fn = noise_fn
        dis = {}
        for name in ("p2", "p3", "p4", "p5"):
            dis[name] = Discriminator(grad_reverse_lambda=0.02).to(dev)
            dis[name].load_state_dict(dis_sd[name])
        for m in [net, gm] + list(dis.values()):
            m.train()
        opts = [torch.optim.Adam(net.parameters(), lr=3e-4 / 3, weight_decay=1e-4)]
        opts += [torch.optim.SGD(m.parameters(), lr=0.0025 / 3, momentum=0.9, weight_decay=1e-4)
                 for m in [gm] + list(dis.values())]
        dice, bce = DiceLoss(), torch.nn.BCEWithLogitsLoss(reduction="mean")
        xs, xt, masks = xs.to(dev), xt.to(dev), masks.to(dev)
        losses = {}
        for step in range(2):
            pred_s, feat_s = net(xs)
            losses["seg_loss"] = dice(pred_s, masks) + bce(pred_s, masks)
            pred_t, feat_t = net(xt)
            score = torch.where(torch.sigmoid(pred_t) > 0.5, 1, 0)
            (f_s, f_t), _nodes, mh = gm((xs, xt), (feat_s, feat_t), targets=masks, score_maps=score)
            losses.update(mh)
            for lvl, name in enumerate(("p2", "p3", "p4", "p5")):
                losses["loss_adv_" + name] = 0.1 * dis[name]((f_s[lvl], f_t[lvl]))
            for o in opts:
                o.zero_grad()
            total = sum(losses.values())
            total.backward()
            for o in opts:
                o.step()
            tol = 1e-3 if step == 0 else 5e-3        # (second step: as in test_full_step_c3_vs_reference_fixture)
            for k in g["loss_keys"]:
                _close(losses[str(k)], g[f"s{step}.{k}"], tol, f"step {step} {k}")
            _close(total, g[f"s{step}.total"], tol, f"step {step} total")
            if step == 0:
                _close(gm.sr_seed, g["s0.sr_seed"], 1e-3, "step 0 sr_seed")
        assert len(draws) == int(g["noise_draws"])
        d = (net.state_dict()["conv3.weight"].cpu() - torch.as_tensor(g["conv3_after"])).abs()
        assert d.max().item() <= 4.2e-4 and d.mean().item() < 2e-5, (d.max().item(), d.mean().item())
    finally:
        graphecho_amd.uninstall_reference_modules()
        for k in [k for k in sys.modules if k.partition(".")[0] in ("models", "utils")]:
            del sys.modules[k]
        sys.modules.update(saved)


def _temporal_c5_trainer(dev, precision):
    from helpers.step_setup import temporal_step_setup
    from graphecho_amd.trainer import GraphEchoTrainer

    sds, xs, xt, masks, clips, noise_fn, draws = temporal_step_setup()
    tr = GraphEchoTrainer(dev, workload="temporal", back_bone="VGG16", in_channel=1, image_size=256, seg_loss="cardiac",
                          clip_len=16, transport_method="sinkhorn_distance", seed=0, conv_precision=precision)
    _no_dropout(tr.graph_model)
    _no_dropout(tr.tgcn)
    tr.load_states(sds)
    tr.graph_model.noise_fn = noise_fn
    data = (xs.to(dev), masks.to(dev), xt.to(dev), {k: v.to(dev) for k, v in clips.items()})
    return tr, data, draws


def test_temporal_step_c5_vs_reference_fixture(dev):
    """BASELINE config 5 as the reference runs it (train_cardiac_uda.py:73,222-320): FPN(in_channel=1, back_bone="VGG16"),
    Dice + BCE over all channels, GModule + four Discriminators on 2 + 2 frames, then the temporal branch -- one source and
    one target clip of 16 frames @256 x 256 through the FPN, GModule on the clip features (sparsely labelled clip: every
    fourth frame hands its prediction on as the target), TGCN with the fp32 SinkhornDistance transport loss -- one
    backward, Adam / SGD.  Every loss term of the step, the TGCN's and the second GModule call's own terms, the Sinkhorn
    cost / plan / cost matrix, logits, gradient probes and the weights after the step against what the reference's own
    modules computed (tools/gen_golden.py:temporal_case), 1e-3 (north_star's tolerance)."""
    g = _gold("temporal_c5")
    tr, (xs, masks, xt, clips), draws = _temporal_c5_trainer(dev, "f32")
    sk_calls, real_sk = [], tr.sinkhorn

    def recording_sinkhorn(x, y):
        out = real_sk(x, y)
        sk_calls.append([o.detach() for o in out])
        return out

    tr.sinkhorn = recording_sinkhorn
    total = tr.step(xs, masks, xt, clips)
    for k in g["loss_keys"]:
        _close(tr.losses[str(k)], g[str(k)], 1e-3, str(k))
    _close(total, g["total"], 1e-3, "total")
    for k in g["tgcn_keys"]:
        _close(tr.last_temporal["tgcn"][str(k)], g["tgcn." + str(k)], 1e-3, "TGCN " + str(k))
    for k in g["clipgm_keys"]:
        _close(tr.last_temporal["graph"][str(k)], g["clipgm." + str(k)], 1e-3, "clip GModule " + str(k))
    assert len(sk_calls) == int(g["sk_calls"]) and len(draws) == int(g["noise_draws"])
    cost, pi, C = sk_calls[-1]
    _close(cost, g["sk_cost"], 1e-3, "Sinkhorn cost")
    _close(pi[..., ::4, ::4], g["sk_pi"], 1e-3, "Sinkhorn plan")
    _close(C[..., ::4, ::4], g["sk_C"], 1e-3, "Sinkhorn cost matrix")
    _close(pi.sum(), g["sk_pi_sum"], 1e-3, "plan mass")
    net = tr.network
    _close(net.conv3.weight.grad, g["g_conv3"], 5e-3, "d conv3")
    l2 = lambda a, b: ((a.detach().cpu().double() - torch.as_tensor(b).double()).norm() / torch.as_tensor(b).double().norm()).item()
    e_top = l2(net.toplayer.weight.grad[:8, :8, 0, 0], g["g_top"])
    e_mlp = l2(tr.tgcn.grapher.MLP[0].weight.grad[:8, :8, 0, 0], g["g_tgcn_mlp"])
    e_gm = l2(tr.graph_model.node_affinity.fc_M[0].weight.grad[:8, :8], g["g_gm"])
    print(f"gradient probes, L2-relative: toplayer {e_top:.2e}, TGCN MLP {e_mlp:.2e}, affinity {e_gm:.2e}")
    # GModule's gradient (no recurrence) is held tight.  The two probes downstream of the TGCN differ by ~1e-1 between two
    # correct fp32 implementations although every loss agrees to 1e-3: 16 recurrent time steps each rebuild a k-NN graph
    # from the previous step's output (a flipped 9th neighbour re-routes gradient, the loss barely notices), and the
    # Sinkhorn cost of 28 at eps = 0.1 sits on exp(+-280)-scaled potentials.  (3 time steps: 1e-2, test_tgcn_vs_reference_
    # fixture.)  Measured 1.0e-1 / 8.0e-2.  These two are SIGN-AND-TERM GUARDS, not accuracy checks: bounded at 1.5e-1 so that a
    # wrong sign or a missing term (error >= 1) fails; the accuracy of the recurrence's backward is held per time step, on
    # oracle-exact inputs, by test_tgcn_backward_per_time_step_on_oracle_inputs (2e-2, measured 5e-7).
    CHAOTIC_PROBE_GUARD = 0.15
    assert e_gm <= 5e-3 and e_top <= CHAOTIC_PROBE_GUARD and e_mlp <= CHAOTIC_PROBE_GUARD, (e_gm, e_top, e_mlp)
    sd = net.state_dict()
    d = (sd["conv3.weight"].cpu() - torch.as_tensor(g["conv3_after"])).abs()
    assert d.max().item() <= 2.1e-4 and d.mean().item() < 2e-5, (d.max().item(), d.mean().item())


@pytest.mark.parametrize("low", ["f16", "f16s"])
def test_temporal_step_c5_f16_convs_dice_vs_fp32(dev, low):
    """The same configuration-5 step with the fp16-MFMA conv path (fp32 Sinkhorn / GModule / statistics): BASELINE's "Dice
    vs ref" for the f16 path = Dice of its thresholded prediction (logits > 0) against the fp32 path's on the fixture's
    source frames and clip frames, from the fixture's weights; the step's losses stay finite and close.
    low = "f16": fp16 MFMA operands, fp32 storage; "f16s": additionally fp16 ACTIVATION STORAGE in the VGG16 conv stacks and
    every other covered 3x3 conv on the blocked-fp16 kernels (graphecho_amd/half.py)."""
    from graphecho_amd import functional as GF

    outs = {}
    for prec in ("f32", low):
        tr, (xs, masks, xt, clips), _ = _temporal_c5_trainer(dev, prec)
        frames = torch.cat([xs, clips["source"].permute(0, 4, 1, 2, 3).reshape(-1, 1, 256, 256)])
        GF.CONV_PRECISION = "f16" if prec == "f16s" else prec
        GF.ACT_STORAGE = "f16" if prec == "f16s" else "f32"
        try:
            with torch.no_grad():
                logits = tr.network(frames)[0]
        finally:
            GF.CONV_PRECISION = "f32"
            GF.ACT_STORAGE = "f32"
        tr.load_states({"Net": {k: v for k, v in _temporal_sd_cache().items()}})      # undo the running-statistics update
        outs[prec] = (logits, tr.step(xs, masks, xt, clips).item(), {k: v.item() for k, v in tr.losses.items()})
        # (a non-finite BatchNorm weight would go unnoticed in the losses: ReLU turns a NaN channel into zeros)
        bad = [n for n, p in tr.network.named_parameters() if not torch.isfinite(p).all()]
        assert not bad, f"{prec}: non-finite parameters after the step: {bad[:4]}"
    a, b = outs[low][0] > 0, outs["f32"][0] > 0
    tp, fp, fn = (a & b).sum().double(), (a & ~b).sum().double(), (~a & b).sum().double()
    dice = ((2 * tp + 1e-5) / (2 * tp + fp + fn + 1e-5)).item()
    print(f"{low} vs fp32 Dice {dice:.5f}, pixels differing {int((a != b).sum())} of {a.numel()}; "
          f"step loss {outs[low][1]:.5f} vs {outs['f32'][1]:.5f}")
    print({k: (round(outs[low][2][k], 4), round(outs["f32"][2][k], 4)) for k in outs["f32"][2]})
    assert dice >= 0.99, dice
    # every loss term but the temporal one agrees with the fp32 step to 1e-2 (measured 1e-4 .. 6e-4); the temporal term -- the
    # Sinkhorn transport cost at the end of TGCN's 16-step recurrence, each step rebuilding a k-NN graph from the previous
    # step's output -- is the chaotic one (two fp32 implementations already differ on its gradients by 1e-1, see the fixture
    # test above): 30.39 in fp32, 32.3 .. 35.1 under the fp16 modes; bounded at 25 %
    for k, v32 in outs["f32"][2].items():
        v = outs[low][2][k]
        assert np.isfinite(v), k
        tol = 0.25 if k == "temporal_graph_loss" else 1e-2
        assert abs(v - v32) <= tol * abs(v32) + 1e-4, (k, v, v32)


def test_temporal_step_c5_f16s_vs_reference_fixture(dev):
    """Config 5 in its STATED dtype (fp16 MFMA conv path + fp16 activation storage, fp32 Sinkhorn / statistics) held directly
    to what the reference's own fp32 modules computed (tests/golden/temporal_c5.npz, tools/gen_golden.py:temporal_case) -- not
    to this build's fp32 path (that was a self-comparison one hop from the fixture).

    fp16 tolerances, stated: segmentation / adversarial losses 5e-3, GModule's terms of both calls (node, matching, node
    discriminator: built on ~500 sampled pyramid rows) 1e-2; the Sinkhorn call is fp32 on both sides: cost, plan and cost matrix
    on the nodes the step itself handed it, 1e-3 against oracle/misc.py.  The temporal transport term is the chaotic one (16
    recurrent steps, each rebuilding a k-NN graph from the previous step's output): 6 - 15 % away from the fixture under fp16.
    Its cause is ATTRIBUTED here: the oracle's TGCN (fp32, CPU) is run on the very inputs the HIP TGCN received in this step --
    the f16s FPN's clip pyramid and GModule's nodes -- and must agree with the HIP value to 2e-2; what is left of the distance
    to the fixture is therefore the FPN's fp16 features perturbing a chaotic recurrence, not an error of the TGCN path."""
    from oracle.misc import sinkhorn_distance
    from oracle.tgcn import tgcn_forward

    g = _gold("temporal_c5")
    tr, (xs, masks, xt, clips), draws = _temporal_c5_trainer(dev, "f16s")
    sk_calls, real_sk = [], tr.sinkhorn

    def recording_sinkhorn(x, y):
        out = real_sk(x, y)
        sk_calls.append((x.detach().cpu(), y.detach().cpu(), [o.detach().cpu() for o in out]))
        return out

    tr.sinkhorn = recording_sinkhorn
    seen = {}

    def grab(_m, args, kwargs):
        seen["feats"] = [f.detach().cpu() for f in args[0]]
        seen["nodes"] = tuple(n.detach().cpu() for n in args[1])

    hook = tr.tgcn.register_forward_pre_hook(grab, with_kwargs=True)
    tg_sd = {k: v.detach().cpu().clone() for k, v in tr.tgcn.state_dict().items()}
    total = tr.step(xs, masks, xt, clips)
    hook.remove()
    assert np.isfinite(total.item())
    measured = {}
    for k in g["loss_keys"]:
        k = str(k)
        if k == "temporal_graph_loss":
            continue
        tol = 5e-3 if (k == "seg_loss" or k.startswith("loss_adv")) else 1e-2
        measured[k] = abs(tr.losses[k].item() - float(g[k])) / max(abs(float(g[k])), 1e-8)
        _close(tr.losses[k], g[k], tol, "f16s " + k)
    for k in g["clipgm_keys"]:
        measured["clip." + str(k)] = abs(tr.last_temporal["graph"][str(k)].item() - float(g["clipgm." + str(k)])) / \
            max(abs(float(g["clipgm." + str(k)])), 1e-8)
        _close(tr.last_temporal["graph"][str(k)], g["clipgm." + str(k)], 1e-2, "f16s clip GModule " + str(k))
    assert len(sk_calls) == int(g["sk_calls"]) and len(draws) == int(g["noise_draws"])
    # fp32 Sinkhorn on the step's own nodes
    x, y, (cost, pi, C) = sk_calls[-1]
    rc, rpi, rC, _ = sinkhorn_distance(x, y, 0.1, 5, "mean")
    _close(cost, rc, 1e-3, "Sinkhorn cost on the step's nodes")
    _close(pi, rpi, 1e-3, "Sinkhorn plan on the step's nodes")
    _close(C, rC, 1e-3, "Sinkhorn cost matrix on the step's nodes")
    # the chaotic term, attributed: same inputs -> the fp32 oracle's TGCN agrees with the HIP TGCN
    hip = tr.last_temporal["tgcn"]["sinkhorn_loss"].item()
    with torch.no_grad():
        ref_tl, _ = tgcn_forward(tg_sd, seen["feats"], seen["nodes"], [8, 4, 2, 1], "sinkhorn_distance", True)
    same_inputs = ref_tl["sinkhorn_loss"].item()
    fixture = float(g["tgcn.sinkhorn_loss"])
    print("f16s vs fixture, relative:", {k: float(f"{v:.1e}") for k, v in measured.items()})
    print(f"TGCN transport loss: HIP {hip:.4f}, fp32 oracle on the SAME (f16s) inputs {same_inputs:.4f}, reference fixture "
          f"(fp32 inputs) {fixture:.4f}")
    assert abs(hip - same_inputs) <= 2e-2 * abs(same_inputs), (hip, same_inputs)
    assert abs(hip - fixture) <= 0.25 * abs(fixture), (hip, fixture)          # (chaos, bounded so that a wrong sign / missing term fails)


def _temporal_sd_cache(_c={}):
    if not _c:
        from helpers.step_setup import temporal_step_setup
        _c.update(temporal_step_setup()[0]["Net"])
    return _c


def test_config5_shaped_step_f16_convs_fp32_sinkhorn(dev):
    """BASELINE config 5's shape in one step: 16-frame clips of 256 x 256 (one source + one target clip, 32 clip frames)
    next to a source/target frame pair, through FPN (fp16-MFMA conv path), GModule, TGCN over 16 time steps and the
    fp32 SinkhornDistance transport loss.  The whole chain is chaotic under fp16 rounding (test_fpn_f16_conv_path_
    tracks_fp32), so each piece is held to its own oracle on the inputs it actually received inside the step:
      * conv layers (first frames of the batch): oracle conv2d on fp16-ROUNDED operands with fp32 accumulation, 2e-4;
      * SinkhornDistance: oracle/misc.py:sinkhorn_distance on the step's own node features -- cost, plan, cost matrix
        within 1e-3 (north_star) and the same stopping iteration;
      * every loss of the step finite, every model updated."""
    import torch.nn.functional as F

    from graphecho_amd import nn as gnn
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
    from oracle.misc import sinkhorn_distance

    hw, t = 256, 16
    tr = GraphEchoTrainer(dev, workload="temporal", image_size=hw, clip_len=t, seed=1, conv_precision="f16",
                          transport_method="sinkhorn_distance")
    xs, ms = synthetic_batch(2, 3, 4, hw, dev, 21)
    xt, _ = synthetic_batch(2, 3, 4, hw, dev, 22)

    def clip(seed):
        f, mk = synthetic_batch(t, 3, 4, hw, dev, seed)
        return (f.reshape(1, t, 3, hw, hw).permute(0, 2, 3, 4, 1).contiguous(),
                mk.reshape(1, t, 4, hw, hw).permute(0, 2, 3, 4, 1).contiguous())

    cs, cm = clip(23)
    ct, _ = clip(24)
    # record what a sample of conv layers and the Sinkhorn call saw
    net = tr.network
    # (layers entered through __call__; the fan-out convs -- conv1 of a Bottleneck, lateral and smoothing convs -- go
    # through forward_with_skip and run the same kernels)
    watch = {"back_bone.layer1.0.conv2": net.back_bone.layer1[0].conv2, "back_bone.layer3.2.conv3": net.back_bone.layer3[2].conv3,
             "toplayer": net.toplayer, "conv2": net.conv2, "back_bone.layer4.0.conv2": net.back_bone.layer4[0].conv2}
    import functools

    seen, hooks = {}, []

    def record(mod, inp, out, name=None):      # must return None: a hook's return value replaces the module output
        if name not in seen:
            seen[name] = (inp[0][:2].detach().cpu(), (out[0] if isinstance(out, tuple) else out)[:2].detach().cpu())

    for name, m in watch.items():
        assert isinstance(m, gnn.Conv2d)
        hooks.append(m.register_forward_hook(functools.partial(record, name=name)))
    sk_calls = []
    real_sk = tr.sinkhorn

    def recording_sinkhorn(x, y):
        out = real_sk(x, y)
        sk_calls.append((x.detach().cpu(), y.detach().cpu(), [o.detach().cpu() for o in out], real_sk.actual_nits))
        return out

    tr.sinkhorn = recording_sinkhorn
    before = {k: o.fp.flat.clone() for k, o in tr.optimizers.items()}
    total = tr.step(xs, ms, xt, {"source": cs, "target": ct, "masks": cm})
    for h in hooks:
        h.remove()
    assert torch.isfinite(total) and all(torch.isfinite(v).all() for v in tr.losses.values())
    assert len(seen) == len(watch)
    for name, (xin, yout) in seen.items():
        m = watch[name]
        w = m.weight.detach().cpu()      # weights AFTER the step moved by <= lr: compare with the pre-step copy below
        flat = tr.optimizers["Net"].fp
        idx = next(i for i, p in enumerate(flat.params) if p is m.weight)
        w0 = before["Net"][flat.offsets[idx]:flat.offsets[idx] + w.numel()].reshape(w.shape).cpu()
        ref = F.conv2d(xin.half().float(), w0.half().float(), None, m.stride, m.padding, 1, m.groups)
        if m.bias is not None:
            ib = next(i for i, p in enumerate(flat.params) if p is m.bias)
            ref = ref + before["Net"][flat.offsets[ib]:flat.offsets[ib] + m.bias.numel()].cpu().view(1, -1, 1, 1)
        err = (yout - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-4, (name, err)
    assert len(sk_calls) == 1
    x, y, (cost, pi, C), nits = sk_calls[0]
    assert x.shape == (1, 64, 256) and y.shape == (1, 64, 256)          # one clip per domain, 8 x 8 nodes, 256 features
    rc, rpi, rC, rn = sinkhorn_distance(x, y, 0.1, 5, "mean")
    assert int(nits.item()) == rn
    _close(C, rC.reshape(C.shape), 1e-3, "Sinkhorn cost matrix")
    _close(pi, rpi.reshape(pi.shape), 1e-3, "Sinkhorn transport plan")
    _close(cost, rc, 1e-3, "Sinkhorn cost")
    assert "temporal_graph_loss" in tr.losses
    for name, opt in tr.optimizers.items():
        assert (opt.fp.flat != before[name]).any(), f"{name}: nothing was updated"



# ----------------------------------------------------------------------------------------------------------
# Reference branches off the trainers' default path (fixture: tools/gen_golden.py:edge_case)
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ver", ["v1", "v2"])
def test_mha_four_heads_vs_reference_fixture(dev, ver):
    """MultiHeadAttention with 4 heads, both head layouts (v2: column blocks; v1: a reshape of the (N, 256) buffer into
    4 x N x 64 without moving data, transformer.py:62-64,92-94): outputs, attention, gradients."""
    from graphecho_amd.models.transformer import MultiHeadAttention
    from oracle.weights import det_tensor, fill_state_dict

    g = _gold("edge_branches")
    mha = MultiHeadAttention(256, 4, dropout=0.0, version=ver)
    mha.load_state_dict(fill_state_dict(mha.state_dict(), seed=11))
    mha = mha.to(dev)
    kv = det_tensor("edge.mha.kv", (70, 256)).to(dev).requires_grad_(True)
    q = det_tensor("edge.mha.q", (50, 256)).to(dev).requires_grad_(True)
    o, a = mha(kv, kv, kv if ver == "v1" else q)
    (o * det_tensor(f"edge.mha.g.{ver}", tuple(o.shape)).to(dev)).sum().backward()
    _close(o, g[f"mha_{ver}_out"], 1e-3, "output")
    _close(a[:, ::5, ::5], g[f"mha_{ver}_att"], 1e-3, "attention")
    _close(kv.grad[::5, ::16], g[f"mha_{ver}_g_kv"], 5e-3, "d key/value")
    _close(mha.linear_q.weight.grad[:8, :8], g[f"mha_{ver}_g_wq"], 5e-3, "d linear_q")


def test_cross_graph_vs_reference_fixture(dev):
    from graphecho_amd.models.transformer import CrossGraph
    from oracle.weights import det_tensor, fill_state_dict

    g = _gold("edge_branches")
    cg = CrossGraph(256, 0.0)
    assert list(cg.state_dict().keys()) == list(g["cg_keys"])
    cg.load_state_dict(fill_state_dict(cg.state_dict(), seed=12))
    cg = cg.to(dev)
    o1, o2 = cg(det_tensor("edge.cg.n1", (37, 256)).to(dev), det_tensor("edge.cg.n2", (45, 256)).to(dev))
    _close(o1, g["cg_o1"], 1e-3, "node_1")
    _close(o2, g["cg_o2"], 1e-3, "node_2")


def test_stochastic_dilation_vs_reference_fixture(dev):
    """DenseDilated(stochastic=True, epsilon=1): in train mode the k neighbours are a random subset of the k*d nearest
    (torch.rand / torch.randperm on the CPU generator: same seed, same picks as the reference); in eval mode every d-th."""
    from graphecho_amd.models.vig import DenseDilatedKnnGraph
    from oracle.knn import knn_graph
    from oracle.weights import det_tensor

    g = _gold("edge_branches")
    x = det_tensor("edge.sto.x", (2, 64, 196, 1))
    mod = DenseDilatedKnnGraph(9, 2, stochastic=True, epsilon=1.0).train()
    torch.manual_seed(777)
    idx = mod(x.to(dev)).cpu().numpy()
    torch.manual_seed(777)
    torch.rand(1)
    pick = torch.randperm(18)[:9].numpy()
    full = knn_graph(x.numpy(), None, 18, 1)
    assert np.array_equal(idx, full[:, :, :, pick])                 # bit-exact against the C oracle
    assert (idx == g["sto_idx"]).mean() > 0.995                    # the reference (tie order of torch.topk aside)
    ev = mod.eval()(x.to(dev)).cpu().numpy()
    assert np.array_equal(ev, full[:, :, :, ::2]) and (ev == g["sto_idx_eval"]).mean() > 0.995


def test_knn_over_10000_points_vs_reference_fixture(dev):
    """dense_knn_matrix switches to 10 000-row chunks above 10 000 points (vig.py:291-303); the HIP kernel tiles the
    rows anyway and must return the same graph: exact on every stable row, bit-exact against the C oracle everywhere."""
    from graphecho_amd.models.vig import DenseDilatedKnnGraph
    from oracle.knn import knn_graph
    from oracle.weights import det_tensor

    g = _gold("edge_branches")
    x = det_tensor("edge.big.x", (1, 16, 10050, 1))
    idx = DenseDilatedKnnGraph(9, 1)(x.to(dev)).cpu().numpy()
    assert idx.shape == (2, 1, 10050, 9)
    ref, stable = g["big_idx"].astype(np.int64), g["big_stable"]
    assert stable.mean() > 0.9
    assert np.array_equal(idx[0, 0][stable], ref[stable])
    assert np.array_equal(idx[1, 0], np.arange(10050)[:, None].repeat(9, 1))
    assert np.array_equal(idx, knn_graph(x.numpy(), None, 9, 1))


def test_gmodule_fewer_than_six_source_nodes(dev):
    """graph_matching.py:258-260: with fewer than 6 source nodes GModule returns the untouched features, the raw node
    sets and NO losses; seed banks stay as they were.  A trainer step then leaves every GModule parameter untouched
    (no gradient: torch's `grad is None` skip) while the other models train."""
    from graphecho_amd.models.graph_matching import GModule
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
    from oracle.weights import det_tensor, fill_state_dict, rect_masks

    g = _gold("edge_branches")
    gm = GModule(256, 4, dev)
    sd = fill_state_dict(gm.state_dict(), seed=6)
    gm.load_state_dict(sd)
    gm = gm.to(dev).train()
    sizes = (64, 32, 16, 8)
    fs = [det_tensor(f"gm.fs{l}", (2, 256, s, s)).to(dev) for l, s in enumerate(sizes)]
    ft = [det_tensor(f"gm.ft{l}", (2, 256, s, s)).to(dev) for l, s in enumerate(sizes)]
    tgt = torch.zeros(2, 4, 256, 256)
    for c in range(4):
        tgt[:, c, 10 + c * 20:12 + c * 20, 10:12] = 1          # 2 x 2 pixel masks contain no sampling location
    feats, (n1, n2), losses = gm(None, (fs, ft), targets=tgt.to(dev), score_maps=rect_masks(2, 4, 256, 256, seed=2).to(dev))
    assert losses == {} and feats[0][0] is fs[0]
    assert [n1.shape[0], n2.shape[0]] == list(g["few_n"]) and n1.shape[1] == 256
    _close(n2[::7, ::16], g["few_n2"], 1e-5, "raw target nodes")
    _close(gm.sr_seed, g["few_sr_seed"], 0, "seed bank untouched")
    # the same inside a training step
    tr = GraphEchoTrainer(dev, workload="full", image_size=128, seed=4)
    x, _ = synthetic_batch(2, 3, 4, 128, dev, 31)
    xt, _ = synthetic_batch(2, 3, 4, 128, dev, 32)
    tiny = torch.zeros(2, 4, 128, 128, device=dev)
    tiny[:, :, 5:7, 5:7] = 1
    before = {k: o.fp.flat.clone() for k, o in tr.optimizers.items()}
    loss = tr.step(x, tiny, xt)
    assert torch.isfinite(loss) and "node_loss" not in tr.losses
    assert torch.equal(tr.optimizers["Graph"].fp.flat, before["Graph"]) and not any(tr.optimizers["Graph"].fp.used)
    assert (tr.optimizers["Net"].fp.flat != before["Net"]).any() and (tr.optimizers["Dis_P3"].fp.flat != before["Dis_P3"]).any()
    # alternating: a regular step leaves GModule's losses in the trainer's persistent dict (train_camus_echo.py:185); the
    # early-return step behind it must not sum those tensors of a freed graph again
    _, m = synthetic_batch(2, 3, 4, 128, dev, 31)
    for phased in (True, False):
        tr.split_backward = phased
        tr.step(x, m, xt)
        assert "node_loss" in tr.losses
        before = tr.optimizers["Graph"].fp.flat.clone()
        loss = tr.step(x, tiny, xt)
        assert torch.isfinite(loss) and not any(k in tr.losses for k in GModule.LOSS_KEYS)
        assert torch.equal(tr.optimizers["Graph"].fp.flat, before)


@pytest.mark.parametrize("workload", ["full", "temporal"])
def test_phased_backward_equals_single_backward(dev, workload):
    """trainer._step_phased cuts the backward pass at the pyramid into three autograd calls (head + discriminators first,
    GModule / TGCN next, the FPN last, from the summed pyramid gradients): same losses and -- up to the association of
    the pyramid-gradient sum -- the same parameters as the single backward call."""
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    res = []
    for phased in (True, False):
        tr = GraphEchoTrainer(dev, workload=workload, image_size=128, seed=5, clip_len=4,
                              transport_method="sinkhorn_distance")
        tr.split_backward = phased
        tr.graph_model.async_seed_update = False
        x, m = synthetic_batch(2, 3, 4, 128, dev, 41)
        xt, _ = synthetic_batch(2, 3, 4, 128, dev, 42)
        clips = None
        if workload == "temporal":
            f, mk = synthetic_batch(2 * 4, 3, 4, 128, dev, 43)
            f = f.reshape(2, 4, 3, 128, 128).permute(0, 2, 3, 4, 1).contiguous()
            mk = mk.reshape(2, 4, 4, 128, 128).permute(0, 2, 3, 4, 1).contiguous()
            clips = {"source": f[:1].repeat(2, 1, 1, 1, 1)[:2], "target": f[1:].repeat(2, 1, 1, 1, 1)[:2],
                     "masks": mk[:1].repeat(2, 1, 1, 1, 1)[:2]}
        for mod in tr.modules.values():       # dropout draws would differ between the two runs
            for sub in mod.modules():
                if isinstance(getattr(sub, "p", None), float):     # nn.Dropout and dot_attention's own rate
                    sub.p = 0.0
        torch.manual_seed(0)
        l0 = tr.step(x, m, xt, clips)
        grads = {k: o.fp.grad.clone() for k, o in tr.optimizers.items()}
        used = {k: list(o.fp.used) for k, o in tr.optimizers.items()}
        res.append((l0.item(), {k: v.item() for k, v in tr.losses.items()}, grads, used))
    (la, da, ga, ua), (lb, db, gb, ub) = res
    assert abs(la - lb) <= 1e-5 * max(1.0, abs(lb)) and da.keys() == db.keys()
    for k in da:
        assert abs(da[k] - db[k]) <= 1e-5 * max(1.0, abs(db[k])), k
    assert ua == ub
    for k in ga:
        err = (ga[k] - gb[k]).norm() / gb[k].norm().clamp_min(1e-12)
        assert err < 2e-4, f"{k}: gradients differ by {err:.2e}"



@pytest.mark.parametrize("workload,models,variant", [
    ("full", {"Net", "Graph", "Dis_P2", "Dis_P3", "Dis_P4", "Dis_P5"}, ""),
    ("temporal", {"Net", "Graph", "Dis_P2", "Dis_P3", "Dis_P4", "Dis_P5", "tgcn_p5"}, ""),
    ("full", {"Net", "Graph", "Dis_P2", "Dis_P3", "Dis_P4", "Dis_P5"}, "_phased"),
    ("temporal", {"Net", "Graph", "Dis_P2", "Dis_P3", "Dis_P4", "Dis_P5", "tgcn_p5"}, "_phased"),
    ("full", {"Net", "Graph", "Dis_P2", "Dis_P3", "Dis_P4", "Dis_P5"}, "pgraphs"),
    ("temporal", {"Net", "Graph", "Dis_P2", "Dis_P3", "Dis_P4", "Dis_P5", "tgcn_p5"}, "pgraphs"),
    # config 5's own backbone and dtype under data parallelism: VGG16 / 1 channel, fp16 MFMA + fp16 activation storage
    # (SyncBN inside the fp16 stacks, every rank its own device-resident loss scale), SinkhornDistance transport
    ("temporal", {"Net", "Graph", "Dis_P2", "Dis_P3", "Dis_P4", "Dis_P5", "tgcn_p5"}, "vgg_f16s")])
def test_ddp_world2_full_workload(dev, tmp_path, workload, models, variant):
    """Config 3/4 (and the temporal config-5 shape: + TGCN, SinkhornDistance, a second GModule call, unused
    TGCN.prediction parameters) under data parallelism (two gloo ranks on this GPU): FPN + GModule + 4 discriminators, SyncBN,
    every model's flat gradients all-reduced (GModule included), unused parameters zero-filled, seed banks rank-local.
    All replicas of every model must stay bit-identical; the run must not deadlock."""
    import subprocess
    import sys

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "ddp_gpu_worker.py")
    port = str(29900 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path), workload, variant])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    a, b = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert set(a["all"]) == models
    for name in a["all"]:
        assert torch.equal(a["all"][name], b["all"][name]), f"replicas of {name} diverged"
        assert torch.isfinite(a["all"][name]).all()
    assert all(np.isfinite(a["losses"])) and all(np.isfinite(b["losses"]))
    if variant == "pgraphs":     # graphs="auto" under data parallelism: head + discriminators replayed, backbone eager
        assert a["graphs"] == b["graphs"] == "head+discriminators" and a["captured"] >= 5, (a["graphs"], a["captured"])
        if workload == "temporal":      # ... and TGCN's recurrence from its own graph (no collective inside)
            assert a["tgcn_graphs"] == b["tgcn_graphs"] == (1, 1), a["tgcn_graphs"]


def _run_ddp_workers(tmp_path, workload, variant=""):
    import subprocess
    import sys

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "ddp_gpu_worker.py")
    port = str(29900 + os.getpid() % 90)
    os.makedirs(tmp_path, exist_ok=True)
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path), workload, variant])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2)]


def test_ddp_world2_sharded_exchange_equals_allreduce(dev, tmp_path):
    """mode "rs_ag" on the real trainer (config 3/4 workload, two gloo ranks on this GPU): reduce-scatter of the gradient
    buckets, fused Adam / SGD kernels on the owned shards only, all-gather of the updated parameters, conv operands
    re-packed after the gather.  Replicas identical, and identical to the all-reduce mode (a two-rank sum)."""
    a, b = _run_ddp_workers(tmp_path / "rs", "full", "rs_ag")
    c, _ = _run_ddp_workers(tmp_path / "ar", "full", "")
    assert a["mode"] == "rs_ag" and c["mode"] == "allreduce"
    for name in a["all"]:
        assert torch.equal(a["all"][name], b["all"][name]), f"rs_ag: replicas of {name} diverged"
        assert torch.equal(a["all"][name], c["all"][name]), f"{name}: sharded exchange changed the result"
    assert a["losses"] == c["losses"]


@pytest.mark.parametrize("variant", ["few1", "few1_phased"])
def test_ddp_world2_gmodule_early_return_on_one_rank(dev, tmp_path, variant):
    """GModule's `< 6 source nodes` early return on rank 1 only (graph_matching.py:258-260): that rank contributes zero
    gradients for GModule, the ranks agree (every step) that its parameters were used, both step them with the
    averaged gradient and stay bit-identical; nothing deadlocks although the ranks' autograd graphs differ -- with the
    single backward call (GModule's buckets exchanged last) and with the phased one (declared complete between the
    autograd calls, exchanged before the FPN's)."""
    a, b = _run_ddp_workers(tmp_path, "full", variant)
    assert "node_loss" in a["loss_keys"] and "node_loss" not in b["loss_keys"]
    assert a["used"] == b["used"] and any(a["used"]["Graph"])
    for name in a["all"]:
        assert torch.equal(a["all"][name], b["all"][name]), f"replicas of {name} diverged"
        assert torch.isfinite(a["all"][name]).all()


def _bench_two_ranks(extra, **env_extra):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GE_DIST_BACKEND="gloo", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200), os.path.join(root, "bench.py"), "--gpus", "2",
           "--steps", "2", "--warmup", "1", "--size", "128"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    return json.loads(lines[0])


def test_bench_two_ranks_rehearsal(dev):
    """bench.py's N > 1 path (torch.distributed.run launch, barrier + max-over-ranks timing, SyncBN, bucketed
    all-reduce, one JSON line from rank 0) rehearsed with two ranks sharing this GPU over gloo.  Default at N > 1 =
    BASELINE config 4: full GraphEcho, strong scaling of a fixed global batch, with the `comm` report."""
    out = _bench_two_ranks(["--global-batch", "8", "--weak-batch", "4"])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["global_batch"] == 8 and out["config"]["per_gpu_batch"] == 4
    assert out["config"]["workload"].startswith("C4") and "syncbn" in out["config"]["parallelism"]
    assert "cpu_baseline" not in out and "scaling_base" not in out      # N = 1 only
    comm = out["comm"]
    assert comm["mode"] == "allreduce" and comm["allreduce_busbw_GBps"] > 0 and comm["grad_collectives_per_step"] >= 6
    assert comm["syncbn"]["allgathers_per_step"] == 50 and comm["syncbn"]["allreduces_per_step"] == 50   # one per BN layer
    # one-shot readiness for the 8-GPU node: what the group reports, every rank's step time, the exchange's exposed share
    assert comm["world_size"] == 2 and len(comm["per_rank_ms_per_step"]["all"]) == 2
    assert comm["compute_only_ms_per_step"] > 0 and "exposed_ms_per_step" in comm
    # SyncBN's exposed time is MEASURED (same exchange, local statistics), and the same run carries a weak-scaling point
    assert comm["syncbn"]["local_bn_ms_per_step"] > 0 and "exposed_ms" in comm["syncbn"]
    wp = out["weak_point"]
    assert wp["scaling"] == "weak" and wp["per_gpu_batch"] == 4 and wp["global_batch"] == 8 and wp["n_gpus"] == 2
    assert wp["value"] > 0 and wp["ms_per_step"] > 0 and wp["steps"] == 5
    # round 6: both curves' points of this N side by side, and one latency figure per SyncBN exchange
    assert out["strong"]["scaling"] == "strong" and out["strong"]["value"] == out["value"] and out["weak"] == wp
    assert comm["syncbn_us_per_collective"] > 0 and comm["syncbn"]["allreduce_us"] > 0
    # at N > 1 the default replays only the collective-free pieces (head, discriminators); the SyncBN backbone is eager
    assert out["config"]["hip_graphs"] in (False, "head+discriminators")


def test_bench_two_ranks_dp_probe_verdicts(dev):
    """bench.py's guard for the one form of the N > 1 step no multi-GPU box has run (SyncBN exchanges captured inside the HIP
    graphs): a sacrificial child per rank steps the distributed trainer first, on a rendezvous of its own; any failure there --
    here a child that never answers -- switches every rank to GE_GRAPHS_DP=partial and is reported in the line, which is still
    printed.  Then the child for real (two gloo ranks on this GPU): six steps, verdict ok."""
    out = _bench_two_ranks(["--global-batch", "8", "--no-comm-report"], GE_DP_PROBE="force", GE_DP_PROBE_FAKE="hang",
                           GE_DP_PROBE_TIMEOUT_S="5")
    probe = out["config"]["dp_probe"]
    assert probe["ok"] is False and "no verdict" in probe["note"] and 4 < probe["seconds"] < 60
    assert out["value"] > 0 and out["config"]["hip_graphs"] in (False, "head+discriminators")
    out = _bench_two_ranks(["--global-batch", "8", "--no-comm-report"], GE_DP_PROBE="force", GE_DP_PROBE_FAKE="fail")
    assert out["config"]["dp_probe"]["ok"] is False and "exit code 3" in out["config"]["dp_probe"]["note"]
    out = _bench_two_ranks(["--global-batch", "8", "--no-comm-report"], GE_DP_PROBE="force")
    assert out["config"]["dp_probe"]["ok"] is True and "note" not in out["config"]["dp_probe"] and out["value"] > 0


def test_bench_two_ranks_aux_watchdog(dev):
    """A rank that stalls in the auxiliary legs behind the timed region (here: rank 1 never enters them) must not cost the
    line: after GE_AUX_TIMEOUT_S rank 0 prints it with what it has, every rank exits 0."""
    out = _bench_two_ranks(["--global-batch", "8"], GE_AUX_FAKE="hang", GE_AUX_TIMEOUT_S="20")
    assert out["value"] > 0 and out["n_gpus"] == 2
    assert "aux_note" in out or "aux_error" in out


def test_dp_probe_child_on_one_rank_rccl(dev):
    """The probe's child itself over RCCL (one rank: all this box has): distributed trainer, default graph mode = everything
    replayed, SyncBN exchanges captured on their own communicator; six steps, exit code 0."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29950 + os.getpid() % 40))
    env.pop("GE_DIST_BACKEND", None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dp-probe", "--batch", "8", "--size", "128",
                          "--ring", "2"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0, (res.stdout + res.stderr)[-2000:]
    assert "DP_PROBE rank 0 ok=1 hip_graphs=all" in res.stdout


def test_bench_two_ranks_weak_scaling_and_sharded_exchange(dev):
    """The weak-scaling option (config 2 at a fixed per-GPU batch) and the reduce-scatter / sharded-optimizer /
    all-gather exchange, same rehearsal."""
    out = _bench_two_ranks(["--workload", "fpn_grapher", "--batch", "4", "--ddp-mode", "rs_ag"])
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 8 and out["value"] > 0
    assert out["comm"]["mode"] == "rs_ag"


def test_bench_two_ranks_config5_f16s_rehearsal(dev):
    """Config 5's 8-GPU leg as far as one GPU allows: bench.py launched by torch.distributed.run with two ranks
    (gloo on this GPU) on the temporal workload with the VGG16 backbone, one input channel, the cardiac loss and the stated dtype
    (--precision f16s): SyncBN over the blocked-fp16 stacks, GModule's parameters receiving gradient in two autograd calls of the
    step with their buckets held until the branch is complete (ddp.GradSynchronizer.defer_fps), the Winograd forward / data /
    weight-gradient kernels on the fp32 3x3 layers left outside the fp16 domain.  One JSON line, finite throughput, per-rank times,
    the SyncBN exchange counted."""
    out = _bench_two_ranks(["--workload", "temporal", "--backbone", "VGG16", "--in-channel", "1", "--seg-loss", "cardiac",
                            "--precision", "f16s", "--batch", "4", "--clip-len", "4"])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["ms_per_step"] > 0
    assert out["dtype"] != "f32" and "f16" in out["dtype"]
    assert out["config"]["hip_graphs"] in (False, "head+discriminators")      # N > 1: nothing with a collective inside is replayed
    comm = out["comm"]
    assert comm["world_size"] == 2 and len(comm["per_rank_ms_per_step"]["all"]) == 2
    assert comm["syncbn"]["allgathers_per_step"] > 0 and comm["grad_collectives_per_step"] >= 6


def test_full_workload_updates_every_model(dev):
    """One config-3 step moves the parameters of every model (FPN incl. its semantic head, GModule, the four
    discriminators) and leaves exactly the parameters without a gradient untouched: the 'gradient complete' hooks fire
    for modules applied twice (FPN on source and target) and for weights shared across pyramid levels."""
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    tr = GraphEchoTrainer(dev, workload="full", image_size=128, seed=2)
    x, m = synthetic_batch(4, 3, 4, 128, dev, 11)
    xt, _ = synthetic_batch(4, 3, 4, 128, dev, 12)
    before = {k: o.fp.flat.clone() for k, o in tr.optimizers.items()}
    loss = tr.step(x, m, xt)
    assert torch.isfinite(loss)
    for name, opt in tr.optimizers.items():
        fp = opt.fp
        moved = (fp.flat != before[name])
        assert moved.any(), f"{name}: nothing was updated"
        for p, o, u in zip(fp.params, fp.offsets, fp.used):
            seg = moved[o:o + p.numel()]
            if u:
                # (a structurally zero gradient -- the affinity MLP's output bias in front of an InstanceNorm -- is
                # rounding noise that may come out as exactly 0: such a parameter legitimately stays put)
                if fp.grad[o:o + p.numel()].abs().max().item() == 0.0 and p.numel() == 1:
                    continue
                assert seg.any(), f"{name}: a parameter marked used did not move"
            else:
                assert not seg.any(), f"{name}: a parameter without gradient moved"
    net = dict(tr.network.named_parameters())
    fpn_fp = tr.optimizers["Net"].fp
    idx = {id(p): i for i, p in enumerate(fpn_fp.params)}
    for key in ("back_bone.conv1.weight", "semantic_branch.weight", "conv2.weight", "conv3.weight", "gn1.weight",
                "smooth3.weight", "toplayer.weight"):
        assert fpn_fp.used[idx[id(net[key])]], f"{key} received no gradient"
