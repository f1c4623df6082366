"""Pins the CPU oracle (oracle/) to the reference: every fixture under tests/golden/ was produced by importing
/root/reference (tools/gen_golden.py); inputs and weights are regenerated here from the same seeds.
Also checks that the HIP modules expose exactly the reference's state_dict keys/shapes (constructed on CPU)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.weights import det_tensor, fill_state_dict, rect_masks

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def close(a, b, rtol=1e-4, what=""):
    a = torch.as_tensor(np.asarray(a.detach() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, f"{what}: {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = max(b.abs().max().item(), 1e-8)
    err = (a - b).abs().max().item()
    assert err <= rtol * scale + 1e-9, f"{what}: err {err:.3e} scale {scale:.3e}"


FPN_CASES = [("resnet_c3_n4_128", "resnet", 3, 4, 128), ("vgg_c1_n1_128", "VGG16", 1, 1, 128),
             ("resnet_c1_n3_256", "resnet", 1, 3, 256),
             ("resnet_c3_n4_256", "resnet", 3, 4, 256)]      # BASELINE config 1 exactly


@pytest.mark.parametrize("tag,bb,cin,nc,hw", FPN_CASES)
def test_fpn_oracle_matches_reference(tag, bb, cin, nc, hw):
    from graphecho_amd.models.fpnseg import FPN
    from oracle.fpn import fpn_forward
    from oracle.misc import seg_loss_cardiac

    g = gold("fpn_" + tag)
    net = FPN([2, 4, 23, 3], nc, cin, back_bone=bb)          # CPU construction only (no forward)
    sd0 = net.state_dict()
    assert list(sd0.keys()) == list(g["keys"]), "state_dict keys differ from the reference"
    assert [str(tuple(v.shape)) for v in sd0.values()] == list(g["shapes"])
    sd = fill_state_dict(sd0, seed=1)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
              for k, v in sd.items()}
    x = det_tensor(f"{tag}.x", (2, cin, hw, hw), "uniform").requires_grad_(True)
    t = (det_tensor(f"{tag}.t", (2, nc, hw, hw), "uniform") > 0.6).float()
    logits, pyr = fpn_forward(params, x, True)
    loss = seg_loss_cardiac(logits, t)
    loss.backward()
    close(logits[:, :, ::8, ::8], g["logits"], 1e-5, "logits")
    close(pyr[3], g["p5"], 1e-5, "p5")
    close(pyr[0].mean((0, 2, 3)), g["p2_mean"], 1e-5, "p2 mean")
    close(pyr[2].std((0, 2, 3)), g["p4_std"], 1e-5, "p4 std")
    close(loss, g["loss"], 1e-6, "loss")
    close(params["smooth3.weight"].grad[:8, :8], g["g_smooth3"], 1e-4, "d smooth3")
    close(params["conv3.weight"].grad, g["g_conv3"], 1e-4, "d conv3")
    close(x.grad[:, :, ::16, ::16], g["g_x"], 1e-4, "d input")


def test_discriminator_oracle_matches_reference():
    from graphecho_amd.models.fpnseg import Discriminator
    from oracle.fpn import discriminator_forward

    g = gold("discriminator")
    dis = Discriminator(grad_reverse_lambda=0.02)
    assert list(dis.state_dict().keys()) == list(g["keys"])
    sd = fill_state_dict(dis.state_dict(), seed=2)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fs = det_tensor("dis.fs", (2, 256, 16, 16)).requires_grad_(True)
    ft = det_tensor("dis.ft", (2, 256, 16, 16)).requires_grad_(True)
    loss = discriminator_forward(params, (fs, ft), 0.02)
    loss.backward()
    close(loss, g["loss"], 1e-6, "loss")
    close(fs.grad[:, ::32, ::4, ::4], g["g_fs"], 1e-4, "d fs (through GRL)")
    close(ft.grad[:, ::32, ::4, ::4], g["g_ft"], 1e-4, "d ft")
    close(params["cls_logits.weight"].grad[0, :16], g["g_cls"], 1e-4, "d cls")


KNN = {"n64_m64": (2, 256, 64, 64, 1), "self256": (2, 64, 256, None, 1), "n1024_m256_d2": (1, 128, 1024, 256, 2),
       "n4096_m256": (1, 256, 4096, 256, 1)}      # config 2's p2 graph: 4096 queries against 256 pooled candidates


@pytest.mark.parametrize("tag", list(KNN))
def test_knn_c_oracle_matches_reference(tag):
    """Bit-exact on every stable row (all top-(k+1) gaps > 1e-5); tie/near-tie rows may differ in order only."""
    from oracle.knn import knn_graph

    g = gold("knn")
    B, C, N, M, d = KNN[tag]
    x = det_tensor(f"knn.{tag}.x", (B, C, N, 1))
    y = None if M is None else det_tensor(f"knn.{tag}.y", (B, C, M, 1))
    idx = knn_graph(x.numpy(), None if y is None else y.numpy(), 9, d)
    ref, stable = g[tag + "_idx"].astype(np.int64), g[tag + "_stable"]
    assert idx.shape == ref.shape
    assert stable.mean() > 0.9
    assert np.array_equal(idx[1], ref[1])                       # centre ids
    assert np.array_equal(idx[0][stable], ref[0][stable])       # neighbour ids, stable rows: exact
    assert (idx[0] == ref[0]).mean() > 0.999


@pytest.mark.parametrize("tag,C,hw,r", [("c64_r2", 64, 16, 2), ("c256_r1", 256, 8, 1), ("c256_r4_64", 256, 64, 4)])
def test_grapher_oracle_matches_reference(tag, C, hw, r):
    from graphecho_amd.models.vig import Grapher
    from oracle.vig import grapher_forward

    g = gold("grapher_" + tag)
    mod = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.0, r, n=hw * hw)
    assert list(mod.state_dict().keys()) == list(g["keys"])
    sd = fill_state_dict(mod.state_dict(), seed=3)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
              for k, v in sd.items()}
    x = det_tensor(f"grapher.{tag}.x", (2, C, hw, hw)).requires_grad_(True)
    y = grapher_forward(params, "", x, 9, 1, r, "gelu", True, True)
    (y * det_tensor(f"grapher.{tag}.g", tuple(y.shape))).sum().backward()
    sp = 4 if hw >= 64 else 1
    close(y[:, ::8, ::sp, ::sp], g["y"], 1e-4, "grapher out")
    close(x.grad[:, ::8, ::sp, ::sp], g["g_x"], 1e-3, "grapher d x")
    close(params["fc1.0.weight"].grad[:8, :8, 0, 0], g["g_fc1"], 1e-3, "d fc1")
    close(params["graph_conv.gconv.nn.0.weight"].grad[:8, :8, 0, 0], g["g_gconv"], 1e-3, "d gconv")


GRAPHCONV_CASES = {  # tag: (conv, act, norm, C_in, C_out, N, M or None) -- tools/gen_golden.py:GRAPHCONV_CASES
    "edge_relu_batch": ("edge", "relu", "batch", 32, 64, 49, None),
    "edge_leaky_batch_xy": ("edge", "leakyrelu", "batch", 32, 48, 50, 16),
    "sage_prelu_batch_xy": ("sage", "prelu", "batch", 32, 64, 50, 16),
    "gin_hswish_none": ("gin", "hswish", None, 32, 64, 49, None),
    "mr_relu_none_xy": ("mr", "relu", None, 32, 64, 50, 16),
}


@pytest.mark.parametrize("tag", list(GRAPHCONV_CASES))
def test_graphconv_oracle_matches_reference(tag):
    """Every aggregator (edge / sage / gin / mr) x activation x norm of GraphConv2d, and the oracle's k-NN edges."""
    from graphecho_amd.models.vig import GraphConv2d
    from oracle.vig import edge_index, graph_conv

    conv, act, norm, ci, co, N, M = GRAPHCONV_CASES[tag]
    g = gold("graphconv")
    mod = GraphConv2d(ci, co, conv, act, norm, True)
    assert list(mod.state_dict().keys()) == list(g[tag + ".keys"])
    sd = fill_state_dict(mod.state_dict(), seed=7)
    params = {"m." + k: (v.clone().requires_grad_(True) if "running" not in k and v.is_floating_point() else v.clone())
              for k, v in sd.items()}
    x = det_tensor(f"gconv.{tag}.x", (2, ci, N, 1)).requires_grad_(True)
    y = det_tensor(f"gconv.{tag}.y", (2, ci, M, 1)).requires_grad_(True) if M else None
    edge = edge_index(x, y, 9, 1)
    assert (edge.numpy() == g[tag + ".edge"]).mean() > 0.995      # ties aside, the C k-NN reproduces torch.topk
    edge = torch.from_numpy(g[tag + ".edge"])
    out = graph_conv(params, "m", conv, x, edge, y, act, norm)
    (out * det_tensor(f"gconv.{tag}.g", tuple(out.shape))).sum().backward()
    close(out, g[tag + ".out"], 1e-5, "out")
    close(x.grad, g[tag + ".g_x"], 1e-4, "d x")
    if M:
        close(y.grad, g[tag + ".g_y"], 1e-4, "d y")
    wkey = "m.gconv.nn1.0.weight" if conv == "sage" else "m.gconv.nn.0.weight"
    close(params[wkey].grad, g[tag + ".g_w"], 1e-4, "d w")
    if conv == "gin":
        close(params["m.gconv.eps"].grad, g[tag + ".g_eps"], 1e-4, "d eps")


def _pvig_state(mod):
    sd = mod.state_dict()
    filled = fill_state_dict(sd, seed=5)
    for k in sd:
        if "relative_pos" in k:     # derived sin-cos table, kept as constructed (tools/gen_golden.py:pvig_case)
            filled[k] = sd[k].clone()
    return filled


def test_pvig_oracle_matches_reference():
    """Pyramid ViG tiny: keys/shapes, the frozen relative-position tables the build constructs,
    and the oracle's logits + gradient probes against the reference's."""
    from graphecho_amd.models.vig import pvig_ti_224_gelu
    from oracle.vig import deepgcn_forward

    g = gold("pvig_ti")
    mod = pvig_ti_224_gelu(num_classes=10)
    sd0 = mod.state_dict()
    assert list(sd0.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in sd0.values()] == list(g["shapes"])
    first = next(k for k in sd0 if k.endswith("relative_pos"))
    last = [k for k in sd0 if k.endswith("relative_pos")][-1]
    close(sd0[first][0, ::64, ::16], g["rel_first"], 1e-6, "relative_pos (first block)")
    close(sd0[last][0, ::7, ::7], g["rel_last"], 1e-6, "relative_pos (last block)")
    sd = _pvig_state(mod)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
                  and "relative_pos" not in k else v.clone()) for k, v in sd.items()}
    x = det_tensor("pvig.x", (2, 3, 224, 224), "uniform").requires_grad_(True)
    y = deepgcn_forward(params, x, [2, 2, 6, 2])
    (y * det_tensor("pvig.g", tuple(y.shape))).sum().backward()
    close(y, g["y"], 1e-3, "pvig logits")
    close(x.grad[:, :, ::16, ::16], g["g_x"], 5e-3, "pvig d x")
    close(params["stem.convs.0.weight"].grad[:8], g["g_stem"], 5e-3, "d stem")
    close(params["pos_embed"].grad[0, :8, ::8, ::8], g["g_pos"], 5e-3, "d pos_embed")
    close(params["backbone.0.0.fc1.0.weight"].grad[:8, :8, 0, 0], g["g_fc1_first"], 5e-3, "d first fc1")
    lastg = [k for k in params if k.endswith("graph_conv.gconv.nn.0.weight")][-1]
    close(params[lastg].grad[:8, :8, 0, 0], g["g_gconv_last"], 5e-3, "d last gconv")


def test_small_ops_oracle_matches_reference():
    from graphecho_amd.models.transformer import MultiHeadAttention
    from graphecho_amd.models.affinity_layer import Affinity
    from oracle import misc

    g = gold("small_ops")
    mha = MultiHeadAttention(256, 1, dropout=0.0, version="v2")
    assert list(mha.state_dict().keys()) == list(g["mha_keys"])
    sd = fill_state_dict(mha.state_dict(), seed=4)
    pre = {"m." + k: v for k, v in sd.items()}
    kv, q = det_tensor("mha.kv", (70, 256)), det_tensor("mha.q", (50, 256))
    o, a = misc.mha_v2(pre, "m", kv, kv, q)
    close(o, g["mha_out"], 1e-5, "mha out")
    close(a[::5, ::5], g["mha_att"], 1e-5, "mha attention")
    aff = Affinity(256)
    assert list(aff.state_dict().keys()) == list(g["aff_keys"])
    sd = {"a." + k: v for k, v in fill_state_dict(aff.state_dict(), seed=5).items()}
    M = misc.affinity(sd, "a", det_tensor("aff.x", (37, 256)), det_tensor("aff.y", (45, 256)))
    close(M, g["aff_M"], 1e-5, "affinity")
    rpm = misc.sinkhorn_rpm(det_tensor("rpm.a", (1, 60, 75)), 20).exp()
    close(rpm[0, ::3, ::3], g["rpm"], 1e-5, "sinkhorn_rpm")
    c3, p3, C3, _ = misc.sinkhorn_distance(det_tensor("sd.x", (2, 64, 256), "uniform"),
                                           det_tensor("sd.y", (2, 64, 256), "uniform"), 0.1, 5, "mean")
    close(c3, g["sd3_cost"], 1e-5, "sinkhorn cost")
    close(p3[:, ::4, ::4], g["sd3_pi"], 1e-5, "sinkhorn plan")
    close(C3[:, ::4, ::4], g["sd3_C"], 1e-6, "sinkhorn C")
    c2, p2, _, _ = misc.sinkhorn_distance(det_tensor("sd2.x", (64, 32), "uniform"),
                                          det_tensor("sd2.y", (50, 32), "uniform"), 0.1, 5, "mean")
    close(c2, g["sd2_cost"], 1e-5, "sinkhorn 2-D cost")
    close(p2[::4, ::4], g["sd2_pi"], 1e-5, "sinkhorn 2-D plan")


@pytest.mark.parametrize("cluster", [0, 1])
def test_gmodule_oracle_matches_reference(cluster):
    from graphecho_amd.models.graph_matching import GModule
    from oracle.gmodule import gmodule_forward

    g = gold(f"gmodule_cluster{cluster}")
    gm = GModule(256, 4, "cpu")
    assert list(gm.state_dict().keys()) == list(g["keys"])
    sd = fill_state_dict(gm.state_dict(), seed=6)
    params = {k: (v.clone().requires_grad_(True) if "seed" not in k or "project" in k else v.clone())
              for k, v in sd.items()}
    sizes = (64, 32, 16, 8)
    fs = [det_tensor(f"gm.fs{l}", (2, 256, s, s)).requires_grad_(True) for l, s in enumerate(sizes)]
    ft = [det_tensor(f"gm.ft{l}", (2, 256, s, s)).requires_grad_(True) for l, s in enumerate(sizes)]
    tgt, sm = rect_masks(2, 4, 256, 256, seed=1), rect_masks(2, 4, 256, 256, seed=2)
    n1, n2, losses, seeds, counts = gmodule_forward(params, (fs, ft), tgt, sm, 4, bool(cluster))
    sum(losses.values()).backward()
    assert [len(n1), len(n2)] == list(g["n_nodes"])
    close(n1[::7, ::16], g["n1"], 1e-4, "nodes_1")
    close(n2[::7, ::16], g["n2"], 1e-4, "nodes_2")
    for k in ("dis_loss", "node_loss", "mat_loss_aff", "mat_loss_qu"):
        close(losses[k], g[k], 1e-4, k)
    close(seeds[0], g["sr_seed"], 1e-4, "sr_seed")
    close(seeds[1], g["tg_seed"], 1e-4, "tg_seed")
    close(fs[0].grad[:, ::32, ::8, ::8], g["g_fs0"], 1e-3, "d p2")
    close(params["node_affinity.fc_M.0.weight"].grad[:8, :8], g["g_aff"], 1e-3, "d fc_M.0")


from helpers.step_setup import full_step_setup  # noqa: E402


STEP_KEYS = ("seg_loss", "dis_loss", "node_loss", "mat_loss_aff", "mat_loss_qu", "loss_adv_p2", "loss_adv_p3",
             "loss_adv_p4", "loss_adv_p5")


@pytest.mark.parametrize("tag,nb,hw", [("128", 2, 128), ("256", 8, 256)])
def test_full_step_oracle_matches_reference(tag, nb, hw):
    """Survey F8: two optimisation steps of the full GraphEcho loop (FPN src + tgt, seg loss, score maps, GModule incl.
    its hallucination branch on a shared noise stream and the scikit-learn seed update, 4 Discriminators, Adam / SGD)
    -- oracle/steps.py:FullCpuTrainer against what the reference's own modules computed.  "256" is BASELINE config 3
    exactly (source 8 + target 8 frames of 3 x 256 x 256)."""
    from oracle.steps import FullCpuTrainer

    g = gold("step_c3_" + tag)
    assert tuple(g["loss_keys"]) == STEP_KEYS
    fpn_sd, gm_sd, dis_sd, xs, xt, masks, noise_fn, draws = full_step_setup(tag, nb, hw)
    tr = FullCpuTrainer(fpn_sd, gm_sd, dis_sd, "cardiac", 4, True, noise_fn)
    for step in range(2):
        total, losses, pred_t, counts = tr.step(xs, masks, xt)
        if step == 0:
            close(pred_t[:, :, ::16, ::16], g["logits_t"], 1e-5, "target logits")
            close(tr.fpn["conv3.weight"].grad, g["g_conv3"], 1e-3, "d conv3")
            close(tr.fpn["toplayer.weight"].grad[:8, :8, 0, 0], g["g_top"], 1e-3, "d toplayer")
            close(tr.dis["p3"]["cls_logits.weight"].grad[0, :16], g["g_dis_p3"], 1e-3, "d dis_p3.cls_logits")
            close(tr.gm["node_affinity.fc_M.0.weight"].grad[:8, :8], g["g_gm"], 1e-3, "d fc_M.0")
        for k in STEP_KEYS:
            close(losses[k], g[f"s{step}.{k}"], 1e-4, f"step {step} {k}")
        close(total, g[f"s{step}.total"], 1e-4, f"step {step} total")
        close(tr.gm["sr_seed"], g[f"s{step}.sr_seed"], 1e-4, f"step {step} sr_seed")
        close(tr.gm["tg_seed"], g[f"s{step}.tg_seed"], 1e-4, f"step {step} tg_seed")
    assert len(draws) == int(g["noise_draws"])
    close(tr.gm["sr_seed"], g["sr_seed"], 1e-4, "sr_seed")
    close(tr.gm["tg_seed"], g["tg_seed"], 1e-4, "tg_seed")
    close(tr.fpn["back_bone.bn1.running_mean"], g["running_mean0"], 1e-5, "running mean after 4 FPN passes")
    # Adam's first steps move each weight by ~lr * sign(g): elements whose gradient is rounding noise may differ by 2 lr
    d = (tr.fpn["conv3.weight"].detach() - torch.as_tensor(g["conv3_after"])).abs()
    assert d.max().item() <= 4.2e-4 and d.mean().item() < 1e-5, (d.max().item(), d.mean().item())


@pytest.mark.parametrize("method", ["node_discriminate", "sinkhorn_distance"])
def test_tgcn_oracle_matches_reference(method):
    from graphecho_amd.models.TGCN import TGCN
    from oracle.tgcn import tgcn_forward

    g = gold("tgcn_" + method)
    m = TGCN(256, 256, (3, 8, 8), 10, 10, transport_method=method)
    assert list(m.state_dict().keys()) == list(g["keys"])
    sd = fill_state_dict(m.state_dict(), seed=7)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
              for k, v in sd.items()}
    feats = [det_tensor(f"tgcn.f{l}", (2, 3, 256, s, s)) for l, s in enumerate((64, 32, 16, 8))]
    nodes = (det_tensor("tgcn.ns", (33, 256)), det_tensor("tgcn.nt", (34, 256)))
    losses, graph = tgcn_forward(params, feats, nodes, [8, 4, 2, 1], method, True)
    sum(losses.values()).backward()
    close(graph[:, ::16, ::4], g["graph"], 1e-4, "current_graph")
    for k, v in losses.items():
        close(v, g[k], 1e-4, k)
    close(params["pos_embed"].grad[:, 0, ::32], g["g_pos"], 2e-3, "d pos_embed")
    close(params["grapher.MLP.0.weight"].grad[:8, :8, 0, 0], g["g_mlp"], 2e-3, "d MLP.0")


def test_input_formatting_oracle_matches_torch_nearest():
    """oracle/data.py against the arithmetic MONAI's Resized(mode='nearest') calls (F.interpolate) + crop + /255 +
    the reference's np.where one-hot and clip fold, at the reference's sizes (328 -> 256 crop, 124 -> 112 crop)."""
    import torch
    import torch.nn.functional as F
    from oracle import data as od

    rng = np.random.default_rng(3)
    for (H, W, S, crop, T, To) in [(300, 420, 328, 256, 1, 1), (150, 131, 124, 112, 1, 1), (96, 80, 72, 64, 5, 8)]:
        img = rng.integers(0, 256, (2, 1, H, W, T), dtype=np.uint8)
        lab = rng.integers(0, 4, (2, H, W, T), dtype=np.uint8)
        offs = [(3, 5), (S - crop, 0)]
        t = torch.from_numpy(img.astype(np.float32))
        size = (S, S, To) if T > 1 else (S, S)
        res = F.interpolate(t if T > 1 else t[..., 0], size=size, mode="nearest")
        if T == 1:
            res = res[..., None]
        ref = torch.stack([res[n, :, oy:oy + crop, ox:ox + crop] for n, (oy, ox) in enumerate(offs)]) / 255.0
        ref = ref.permute(0, 4, 1, 2, 3).reshape(-1, 1, crop, crop).numpy()
        got = od.prepare_frames(img if T > 1 else img[..., 0], S, crop, offsets=offs, clip_length=To if T > 1 else None)
        assert got.shape == ref.shape and np.array_equal(got, ref.astype(np.float32))
        onehot = np.stack([np.where(lab == v, 1, 0) for v in (0, 1, 2)], axis=1).astype(np.float32)   # N,3,H,W,T
        r2 = F.interpolate(torch.from_numpy(onehot if T > 1 else onehot[..., 0]), size=size, mode="nearest")
        if T == 1:
            r2 = r2[..., None]
        c = S // 2 - crop // 2
        r2 = r2[:, :, c:c + crop, c:c + crop].permute(0, 4, 1, 2, 3).reshape(-1, 3, crop, crop).numpy()
        g2 = od.onehot_labels(lab if T > 1 else lab[..., 0], (0, 1, 2), S, crop, center=True, clip_length=To if T > 1 else None)
        assert np.array_equal(g2, r2)


def test_edge_branches_oracle_matches_reference():
    """The two off-default branches the oracle itself restates: the > 10 000-point k-NN (chunked in the reference,
    vig.py:291-303) on the C oracle, and GModule's `< 6 source nodes` early return (graph_matching.py:258-260)."""
    from oracle.gmodule import gmodule_forward
    from oracle.knn import knn_graph

    g = gold("edge_branches")
    x = det_tensor("edge.big.x", (1, 16, 10050, 1))
    idx = knn_graph(x.numpy(), None, 9, 1)
    ref, stable = g["big_idx"].astype(np.int64), g["big_stable"]
    assert stable.mean() > 0.9 and np.array_equal(idx[0, 0][stable], ref[stable])
    assert (idx[0, 0] == ref).mean() > 0.999
    from graphecho_amd.models.graph_matching import GModule
    sd = fill_state_dict(GModule(256, 4, "cpu").state_dict(), seed=6)
    sizes = (64, 32, 16, 8)
    fs = [det_tensor(f"gm.fs{l}", (2, 256, s, s)) for l, s in enumerate(sizes)]
    ft = [det_tensor(f"gm.ft{l}", (2, 256, s, s)) for l, s in enumerate(sizes)]
    tgt = torch.zeros(2, 4, 256, 256)
    for c in range(4):
        tgt[:, c, 10 + c * 20:12 + c * 20, 10:12] = 1
    n1, n2, losses, seeds, counts = gmodule_forward(sd, (fs, ft), tgt, rect_masks(2, 4, 256, 256, seed=2), 4)
    assert losses == {} and list(counts) == list(g["few_n"])
    close(n2[::7, ::16], g["few_n2"], 1e-6, "raw target nodes")
    close(seeds[0], g["few_sr_seed"], 0, "seed bank untouched")


def test_temporal_step_oracle_matches_reference():
    """BASELINE config 5 as the reference runs it (FPN(in_channel=1, VGG16), Dice + BCE, 2 + 2 frames, one source + one target
    clip of 16 frames @256 x 256, GModule twice, TGCN + SinkhornDistance): oracle/steps.py:TemporalCpuTrainer against what
    the reference's own modules computed (tools/gen_golden.py temporal_case) -- every loss term of the step, the TGCN's
    and the clip GModule call's own terms, gradient probes, weights after the step."""
    from helpers.step_setup import temporal_step_setup
    from oracle.steps import TemporalCpuTrainer

    g = gold("temporal_c5")
    sds, xs, xt, masks, clips, noise_fn, draws = temporal_step_setup()
    dis = {"p" + k[-1]: v for k, v in sds.items() if k.startswith("Dis_P")}
    tr = TemporalCpuTrainer(sds["Net"], sds["Graph"], dis, sds["tgcn_p5"], "cardiac", 4, True, noise_fn)
    total, losses, tl, tgl = tr.step(xs, masks, xt, clips)
    assert list(losses) == [str(k) for k in g["loss_keys"]]
    for k in g["loss_keys"]:
        close(losses[str(k)], g[str(k)], 2e-4, str(k))
    close(total, g["total"], 2e-4, "total")
    for k in g["tgcn_keys"]:
        close(tl[str(k)], g["tgcn." + str(k)], 2e-4, "TGCN " + str(k))
    for k in g["clipgm_keys"]:
        close(tgl[str(k)], g["clipgm." + str(k)], 2e-4, "clip GModule " + str(k))
    assert len(draws) == int(g["noise_draws"])
    close(tr.fpn["conv3.weight"].grad, g["g_conv3"], 1e-3, "d conv3")
    close(tr.gm["node_affinity.fc_M.0.weight"].grad[:8, :8], g["g_gm"], 1e-3, "d fc_M.0")
    close(tr.gm["sr_seed"], g["sr_seed"], 1e-4, "sr_seed")
    d = (tr.fpn["conv3.weight"].detach() - torch.as_tensor(g["conv3_after"])).abs()
    assert d.max().item() <= 2.1e-4 and d.mean().item() < 1e-5, (d.max().item(), d.mean().item())


def test_fpn_oracle_rounded_storage_mode_properties():
    """oracle/fpn.py HALF_PLAN (round 6, config 5's stated dtype restated at the arithmetic level): without a plan nothing changes
    (the fixtures above pin that path); with one, the VGG16 features are exactly fp16-representable (they are the stacks' stored
    outputs), the result sits within a few 1e-2 of the fp32 evaluation, and the plan is consulted once per convolution call."""
    import torch
    from graphecho_amd.models.fpnseg import FPN
    from oracle import fpn as O

    torch.manual_seed(1)
    sd = {k: v.detach().clone() for k, v in FPN([2, 4, 23, 3], 4, 1, back_bone="VGG16").state_dict().items()}
    x = torch.rand(2, 1, 64, 64)
    seen = []

    def plan(name, xs, ws, stride, padding, groups):
        seen.append(name)
        if "block_" in name:
            return "stem" if ws[1] < 32 else "f16s"
        return "f16" if ws[0] % 32 == 0 and ws[1] % 32 == 0 else "f32"

    with torch.no_grad():
        a, _ = O.fpn_forward({k: v.clone() for k, v in sd.items()}, x, True)
        O.HALF_PLAN = plan
        try:
            feats = O.vgg_forward({k: v.clone() for k, v in sd.items()}, "back_bone", x, True)
            n_vgg = len(seen)
            b, _ = O.fpn_forward({k: v.clone() for k, v in sd.items()}, x, True)
        finally:
            O.HALF_PLAN = None
        c, _ = O.fpn_forward({k: v.clone() for k, v in sd.items()}, x, True)
    assert torch.equal(a, c)
    assert all(torch.equal(f, f.half().float()) for f in feats)
    assert n_vgg == 2 * 13          # mode query + convolution per layer
    rel = ((a - b).abs().max() / a.abs().max()).item()
    assert 1e-4 < rel < 5e-2, rel
