"""Learned node affinity (reference models/affinity_layer.py:8-73), algebraically fused.

The reference materialises an (N1, N2, 512) broadcast-concat and runs Linear(512,512)-ReLU-Linear(512,1) on
every pair: O(N1*N2*512^2).  Because the first Linear acts on [X_i ; Y_j], it splits into
``P = X' W1[:, :256]^T`` (N1 x 512) and ``Q = Y' W1[:, 256:]^T`` (N2 x 512), and
``M[i,j] = b2 + w2 . relu(P_i + Q_j + b1)`` is evaluated by one kernel in O(N1*N2*512) without the concat.
Parameter names / shapes are unchanged (fc_M.0, fc_M.2, project_sr, project_tg).
"""
import torch.nn as nn

from .. import functional as GF
from .. import nn as gnn


class Affinity(nn.Module):
    def __init__(self, d=256):
        super().__init__()
        self.d = d
        self.fc_M = nn.Sequential(gnn.Linear(512, 512), gnn.ReLU(), gnn.Linear(512, 1))
        self.project_sr = gnn.Linear(256, 256, bias=False)
        self.project_tg = gnn.Linear(256, 256, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        for layer in self.fc_M:
            if isinstance(layer, nn.Linear):
                nn.init.normal_(layer.weight, std=0.01)
                nn.init.constant_(layer.bias, 0)
        nn.init.normal_(self.project_sr.weight, std=0.01)
        nn.init.normal_(self.project_tg.weight, std=0.01)

    def forward(self, X, Y):
        X = self.project_sr(X)
        Y = self.project_tg(Y)
        c = X.shape[1]
        w1, b1 = self.fc_M[0].weight, self.fc_M[0].bias
        P = GF.matmul(X, w1[:, :c], False, True)
        Q = GF.matmul(Y, w1[:, c:], False, True)
        return GF.affinity_mlp(P, Q, b1, self.fc_M[2].weight, self.fc_M[2].bias).squeeze()
