"""Temporal graph module (reference models/TGCN.py:41-78, 168-285) on the gfx950 kernels.

``TGCN(input_dim, hidden_dim, clip_shape, soucre_class, target_class, cluster_method, transport_method)`` and
``forward(input_features, input_feature_nodes, loss_trans, loss_cluster, update_index, r)`` keep the reference
contract (including the "soucre" spelling) and the state_dict layout: ``pos_embed``, ``grapher.gconv.nn.0``,
``grapher.MLP.{0,1,4}``, ``graph_attention.*``, ``prediction.{0,1}``, ``node_dis_2.*``.

Per time step: average-pool the four pyramid levels to the clip grid, concat (B,1024,h,w), 1x1-conv MLP,
+ learnable position, k-NN(k=9) of the current nodes against the previous hidden state, max-relative
aggregation + grouped 1x1 conv.  The time loop is a true recurrence and stays sequential on one device.
The reference's unused helpers (TGCNGraphConvolution, TGCNCell, laplacian utilities, concat_all_gather) are dead
code there and are not reproduced.
"""
import torch
import torch.nn as nn

from .. import functional as GF
from .. import nn as gnn
from .gradient_reversal import GradientReversal
from .transformer import MultiHeadAttention
from .vig import DenseDilatedKnnGraph, GraphConv2d


class DyGraphConv2d(GraphConv2d):
    """Dynamic graph conv over time: current nodes attend to the previous step's graph (TGCN.py:41-78)."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv="mr", act="gelu", norm=None,
                 bias=True, stochastic=False, epsilon=0.2):
        super().__init__(in_channels, out_channels, conv, act, norm, bias)
        self.k = kernel_size
        self.d = dilation
        self.MLP = nn.Sequential(
            gnn.Conv2d(in_channels * 4, out_channels, 1, stride=1, bias=True),
            gnn.BatchNorm2d(out_channels),
            gnn.GELU(),
            nn.Dropout(0.1),
            gnn.Conv2d(out_channels, out_channels, 1, stride=1, bias=True),
        )
        self.dilated_knn_graph = DenseDilatedKnnGraph(kernel_size, dilation, stochastic, epsilon)

    def forward(self, input, rs, y, learnable_pos, relative_pos=None):
        pooled = [GF.avg_pool2d(f, r) if r > 1 else f for f, r in zip(input, rs)]
        x = self.MLP(torch.cat(pooled, dim=1))
        x = x + learnable_pos
        B, C, H, W = x.shape
        x = x.reshape(B, C, -1, 1)
        edge_index = self.dilated_knn_graph(x, y, relative_pos)
        x = super().forward(x, edge_index, y)
        return x.reshape(B, -1, H * W), H, W


class TGCN(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int, clip_shape: tuple, soucre_class: int, target_class: int,
                 cluster_method=None, transport_method="node_discriminate"):
        super().__init__()
        self._input_dim = input_dim
        self._hidden_dim = hidden_dim
        self.grapher = DyGraphConv2d(input_dim, hidden_dim)
        self.graph_attention = MultiHeadAttention(256, 1, dropout=0.1, version="v2")
        self.clip_l, self.clip_h, self.clip_w = clip_shape
        self.cluster_method = cluster_method
        self.transport_method = transport_method
        self.pos_embed = nn.Parameter(torch.zeros(self.clip_l, 1, input_dim, self.clip_h, self.clip_w))
        self.prediction = nn.Sequential(
            gnn.Conv2d(self._hidden_dim, self._hidden_dim, 3, stride=2, bias=True),
            gnn.BatchNorm2d(self._hidden_dim),
            gnn.GELU(),
            nn.Dropout(0.1),
            gnn.AdaptiveAvgPool2d1(),
        )
        if self.cluster_method == "momentum_queue":
            self.m = 0.99
            self.K = 150
            self.register_buffer("queue_source", torch.randn(self._hidden_dim, self.K))
            self.register_buffer("queue_target", torch.randn(self._hidden_dim, self.K))
            self.queue_source = nn.functional.normalize(self.queue_source, dim=0)
            self.queue_target = nn.functional.normalize(self.queue_target, dim=0)
        elif self.cluster_method == "linear_clustering":
            self.classifer_source = gnn.Linear(self._hidden_dim, soucre_class)
            self.classifer_target = gnn.Linear(self._hidden_dim, target_class)
        if self.transport_method == "node_discriminate":
            self.grad_reverse = GradientReversal(0.02)
            self.node_dis_2 = nn.Sequential(
                gnn.Linear(256, 256), gnn.LayerNorm(256, elementwise_affine=False), gnn.ReLU(),
                gnn.Linear(256, 256), gnn.LayerNorm(256, elementwise_affine=False), gnn.ReLU(),
                gnn.Linear(256, 256), gnn.LayerNorm(256, elementwise_affine=False), gnn.ReLU(),
                gnn.Linear(256, 1),
            )
            for layer in self.node_dis_2:
                if isinstance(layer, nn.Linear):
                    nn.init.normal_(layer.weight, std=0.01)
                    nn.init.constant_(layer.bias, 0)

    def loss_bce(self, logits, target):
        return GF.bce_with_logits(logits, target)

    def forward(self, input_features, input_feature_nodes, loss_trans, loss_cluster, update_index, r=1.0):
        losses = dict()
        x_f1, x_f2, x_f3, x_f4 = input_features
        source_nodes, target_nodes = input_feature_nodes
        batch_size, seq_len = x_f1.shape[0], x_f1.shape[1]

        hidden_state = torch.zeros(batch_size, self._input_dim, self.clip_h * self.clip_w, dtype=x_f1.dtype,
                                   device=x_f1.device)
        for i in range(seq_len):
            step = [x_f1[:, i], x_f2[:, i], x_f3[:, i], x_f4[:, i]]
            current_graph, H, W = self.grapher(step, r, hidden_state, self.pos_embed[i])
            hidden_state = current_graph
        batch_size, features, num_nodes = current_graph.shape
        output_f = self.prediction(current_graph.reshape(batch_size, features, H, W)).view(batch_size, -1)

        update_index_source, update_index_target = update_index
        half = batch_size // 2
        if self.cluster_method == "momentum_queue":
            q = nn.functional.normalize(output_f, dim=1)
            bank = torch.cat([self.queue_source, self.queue_target], dim=-1).clone().detach()
            l_pos = GF.matmul(q, bank)
            self._dequeue_and_enqueue(q[:half], self.queue_source, update_index_source)
            self._dequeue_and_enqueue(q[half:], self.queue_target, update_index_target)
            losses["clustering_loss"] = loss_cluster(
                l_pos, torch.cat([update_index_source, torch.add(update_index_target, 150)]))
        elif self.cluster_method == "linear_clustering":
            losses["clustering_loss"] = \
                loss_cluster(self.classifer_source(output_f[:half]), update_index_source) + \
                loss_cluster(self.classifer_target(output_f[half:]), update_index_target)

        output_g = current_graph.transpose(1, 2)            # (b, nodes, 256)
        b_g, d_g, n_g = output_g.shape
        output_g = output_g.reshape(b_g * d_g, n_g)
        n_out = output_g.shape[0]
        nodes_ = torch.cat([output_g, source_nodes, target_nodes])
        nodes_ = self.graph_attention(nodes_, nodes_, nodes_)[0]
        nodes_g = nodes_[:n_out].reshape(b_g, d_g, n_g)
        nodes_source = nodes_g[:b_g // 2].reshape(-1, n_g)
        nodes_target = nodes_g[b_g // 2:].reshape(-1, n_g)

        if self.transport_method == "node_discriminate":
            nodes_rev = self.grad_reverse(torch.cat([nodes_source, nodes_target], dim=0))
            tg_rev = torch.cat([torch.ones(nodes_source.size(0), device=nodes_g.device),
                                torch.zeros(nodes_target.size(0), device=nodes_g.device)])
            nodes_rev = self.node_dis_2(nodes_rev)
            losses["node_dis_loss"] = 0.1 * self.loss_bce(nodes_rev.view(-1), tg_rev)
        elif self.transport_method == "sinkhorn_distance":
            losses["sinkhorn_loss"] = loss_trans(nodes_g[:half], nodes_g[half:])[0]
        return losses

    @torch.no_grad()
    def _momentum_update_key_encoder(self, encoder_q, encoder_k):
        for param_q, param_k in zip(encoder_q.parameters(), encoder_k.parameters()):
            param_k.data = param_k.data * self.m + param_q.data * (1.0 - self.m)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, features, queue, labels):
        for idx, l_idx in enumerate(labels):
            queue[:, l_idx] = queue[:, l_idx] * self.m + features[idx] * (1.0 - self.m)

    @property
    def hyperparameters(self):
        return {"input_dim": self._input_dim, "hidden_dim": self._hidden_dim}
