"""Single/multi-head attention over node sets (reference models/transformer.py:5-110) on the HIP GEMM,
softmax and LayerNorm kernels.

Reference quirks kept: ``forward(key, value, query)`` argument order; ``scale = (dim_per_head // num_heads) ** -0.5``;
the returned attention matrix is the post-dropout one; both outputs are ``.squeeze()``-d.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from .. import functional as GF
from .. import nn as gnn

# GE_FUSED_MHA=0: the attention block as its dozen separate ops (and two dozen in backward) instead of ge_mha1_*
FUSED_MHA = os.environ.get("GE_FUSED_MHA", "1") != "0"


class dot_attention(nn.Module):
    """softmax(q k^T * scale) v  (transformer.py:5-23)."""

    def __init__(self, attention_dropout=0.0):
        super().__init__()
        self.p = attention_dropout

    def forward(self, q, k, v, scale=None, attn_mask=None):
        if attn_mask is not None:
            raise NotImplementedError("attn_mask is never used by the reference training path")
        ctxs, atts = [], []
        for h in range(q.shape[0]):
            att = GF.softmax_lastdim(GF.matmul(q[h], k[h], False, True), scale if scale else 1.0)
            att = F.dropout(att, self.p, self.training)
            ctxs.append(GF.matmul(att, v[h]))
            atts.append(att)
        if len(ctxs) == 1:       # one head (every attention of GModule and TGCN): a view instead of two copy kernels each way
            return ctxs[0].unsqueeze(0), atts[0].unsqueeze(0)
        return torch.stack(ctxs), torch.stack(atts)


class MultiHeadAttention(nn.Module):
    def __init__(self, model_dim=256, num_heads=4, dropout=0.0, version="v2"):
        super().__init__()
        self.dim_per_head = model_dim // num_heads
        self.num_heads = num_heads
        self.linear_k = gnn.Linear(model_dim, self.dim_per_head * num_heads)
        self.linear_v = gnn.Linear(model_dim, self.dim_per_head * num_heads)
        self.linear_q = gnn.Linear(model_dim, self.dim_per_head * num_heads)
        self.dot_product_attention = dot_attention(dropout)
        self.linear_final = gnn.Linear(model_dim, model_dim)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = gnn.LayerNorm(model_dim)
        self.version = version

    def _fused_params_ok(self):
        """ge_mha1_* reads all four biases and the LayerNorm affine pair: any of them missing -> the composed path."""
        lins = (self.linear_k, self.linear_v, self.linear_q, self.linear_final)
        return all(m.bias is not None for m in lins) and self.layer_norm.weight is not None and self.layer_norm.bias is not None

    def forward(self, key, value, query, attn_mask=None):
        if self.version not in ("v1", "v2"):
            raise ValueError(self.version)
        H, dph = self.num_heads, self.dim_per_head
        if (FUSED_MHA and H == 1 and self.version == "v2" and attn_mask is None and query.is_cuda and key.dim() == 2 and
                value.dim() == 2 and query.dim() == 2 and key.shape == value.shape and self._fused_params_ok()):
            # every attention block of GModule and TGCN: the same kernels from ONE call per direction (csrc/ge_attention.hip);
            # the two dropout masks are drawn here (Bernoulli keep masks instead of F.dropout's fused draw: same law)
            p_att, p_out = (self.dot_product_attention.p, self.dropout.p) if self.training else (0.0, 0.0)
            m_att = torch.empty((query.size(0), key.size(0)), device=query.device).bernoulli_(1.0 - p_att) if p_att > 0 else None
            m_out = torch.empty((query.size(0), H * dph), device=query.device).bernoulli_(1.0 - p_out) if p_out > 0 else None
            ln = self.layer_norm
            out, attention = GF.mha1(key, value, query, self.linear_k.weight, self.linear_k.bias, self.linear_v.weight,
                                     self.linear_v.bias, self.linear_q.weight, self.linear_q.bias, self.linear_final.weight,
                                     self.linear_final.bias, ln.weight, ln.bias, m_att, m_out, (dph // H) ** -0.5,
                                     1.0 / (1.0 - p_att) if p_att < 1 else 0.0, 1.0 / (1.0 - p_out) if p_out < 1 else 0.0, ln.eps)
            return out.squeeze(), attention.squeeze()
        residual = query
        k = self.linear_k(key)
        v = self.linear_v(value)
        q = self.linear_q(query)
        if self.version == "v2":
            # (N, H*dph) -> (H, N, dph): head h takes columns [h*dph, (h+1)*dph)   (transformer.py:62-64)
            split = lambda t: t.view(t.size(0), H, dph).transpose(0, 1)
        else:
            # v1 views the (1, N, H*dph) buffer as (H, N, dph) without moving data   (transformer.py:92-94)
            split = lambda t: t.reshape(H, -1, dph)
        k, v, q = split(k), split(v), split(q)
        scale = (dph // H) ** -0.5
        context, attention = self.dot_product_attention(q, k, v, scale, attn_mask)
        if self.version == "v2":
            context = context.transpose(0, 1).reshape(query.size(0), H * dph)
        else:
            context = context.reshape(-1, H * dph)
        out = self.dropout(self.linear_final(context))
        out = self.layer_norm(residual + out)
        return out.squeeze(), attention.squeeze()


class CrossGraph(nn.Module):
    """Bidirectional cross-graph attention (transformer.py:115-160; unused by the trainers)."""

    def __init__(self, model_dim=256, dropout=0.0):
        super().__init__()
        self.linear_edge = gnn.Linear(model_dim, model_dim)
        self.linear_node1 = gnn.Linear(model_dim, model_dim)
        self.linear_node2 = gnn.Linear(model_dim, model_dim)
        self.dot_product_attention = dot_attention(dropout)
        self.linear_final = gnn.Linear(model_dim, model_dim)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = gnn.LayerNorm(model_dim)

    def forward(self, node_1, node_2, attn_mask=None):
        e1, e2 = self.linear_edge(node_1), self.linear_edge(node_2)
        n1, n2 = self.linear_node1(node_1), self.linear_node1(node_2)
        att = GF.matmul(e1, e2, False, True)
        o1 = GF.matmul(GF.softmax_lastdim(att), n2)
        o2 = GF.matmul(GF.softmax_lastdim(att.t().contiguous()), n1)
        o1 = self.dropout(self.linear_final(o1))
        o2 = self.dropout(self.linear_final(o2))
        return self.layer_norm(node_1 + o1), self.layer_norm(node_2 + o2)
