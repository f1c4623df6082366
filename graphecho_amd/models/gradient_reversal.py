"""Gradient reversal layer (reference models/gradient_reversal.py:6-33): identity forward, -lambda * grad backward."""
import torch
from torch.autograd import Function


class GradientReversalFunction(Function):
    @staticmethod
    def forward(ctx, x, lambda_):
        ctx.lambda_ = lambda_
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grads):
        return grads * (-float(ctx.lambda_)), None


class GradientReversal(torch.nn.Module):
    def __init__(self, lambda_=1):
        super().__init__()
        self.lambda_ = lambda_

    def forward(self, x):
        return GradientReversalFunction.apply(x, self.lambda_)
