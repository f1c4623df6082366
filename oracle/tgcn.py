"""Oracle: TGCN.forward as a pure function of a state_dict (reference models/TGCN.py:62-78, 224-285).

Dropout is taken at p=0 (parity runs use eval-mode dropout; BatchNorm stays in train mode when training=True).
TEST INFRASTRUCTURE ONLY.
"""
import torch
import torch.nn.functional as F

from .misc import grad_reverse, layer_norm, mha_v2, sinkhorn_distance
from .vig import _bn, edge_index, mr_conv


def grapher_step(sd, feats, rs, hidden, pos, training):
    """One time step of TGCN.DyGraphConv2d.forward (TGCN.py:62-78) -> (B, 256, h*w)."""
    pooled = [F.avg_pool2d(f, r, r) if r > 1 else f for f, r in zip(feats, rs)]
    x = torch.cat(pooled, dim=1)
    x = F.conv2d(x, sd["grapher.MLP.0.weight"], sd["grapher.MLP.0.bias"])
    x = F.gelu(_bn(sd, "grapher.MLP.1", x, training))
    x = F.conv2d(x, sd["grapher.MLP.4.weight"], sd["grapher.MLP.4.bias"])
    x = x + pos
    B, C, H, W = x.shape
    x = x.reshape(B, C, -1, 1)
    e = edge_index(x, hidden, 9, 1)
    out = mr_conv(sd, "grapher.gconv", x, e, hidden.unsqueeze(-1) if hidden.dim() == 3 else hidden, "gelu", False,
                  training)
    return out.reshape(B, -1, H * W), H, W


def tgcn_forward(sd, input_features, nodes, r, transport_method="node_discriminate", training=True,
                 sinkhorn_cfg=(0.1, 5, "mean")):
    """Returns (losses dict, current_graph)."""
    f1, f2, f3, f4 = input_features
    src_nodes, tgt_nodes = nodes
    b, t = f1.shape[0], f1.shape[1]
    hw = sd["pos_embed"].shape[-2] * sd["pos_embed"].shape[-1]
    hidden = torch.zeros(b, f1.shape[2], hw)
    for i in range(t):
        hidden, H, W = grapher_step(sd, [f1[:, i], f2[:, i], f3[:, i], f4[:, i]], r, hidden, sd["pos_embed"][i], training)
    graph = hidden
    out_g = graph.transpose(1, 2)
    bg, dg, ng = out_g.shape
    out_g = out_g.reshape(bg * dg, ng)
    allnodes = torch.cat([out_g, src_nodes, tgt_nodes])
    att_nodes, _ = mha_v2(sd, "graph_attention", allnodes, allnodes, allnodes)
    nodes_g = att_nodes[:bg * dg].reshape(bg, dg, ng)
    losses = {}
    if transport_method == "node_discriminate":
        ns, nt = nodes_g[:bg // 2].reshape(-1, ng), nodes_g[bg // 2:].reshape(-1, ng)
        z = grad_reverse(torch.cat([ns, nt]), 0.02)
        for i in (0, 3, 6):
            z = F.relu(layer_norm(F.linear(z, sd[f"node_dis_2.{i}.weight"], sd[f"node_dis_2.{i}.bias"])))
        z = F.linear(z, sd["node_dis_2.9.weight"], sd["node_dis_2.9.bias"]).view(-1)
        tgt = torch.cat([torch.ones(ns.shape[0]), torch.zeros(nt.shape[0])])
        losses["node_dis_loss"] = 0.1 * F.binary_cross_entropy_with_logits(z, tgt)
    else:
        eps, it, red = sinkhorn_cfg
        losses["sinkhorn_loss"] = sinkhorn_distance(nodes_g[:b // 2], nodes_g[b // 2:], eps, it, red)[0]
    return losses, graph
