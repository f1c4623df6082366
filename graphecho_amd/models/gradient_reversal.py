"""Gradient reversal (Ganin & Lempitsky 2015), the adversarial hinge of the discriminators.

Same surface as the reference's models/gradient_reversal.py (``GradientReversal(lambda_)``,
``GradientReversalFunction.apply(x, lambda_)``): the forward result equals the input, the backward pass hands
``-lambda_`` times the incoming gradient upstream.  Differences in mechanics only: the forward returns an alias of
the input instead of a copy (the pyramid levels it is applied to are 134 MB each at batch 32), and the factor is
folded into a Python float once instead of being materialised as a tensor per call.
"""
import torch


class _ScaleGradient(torch.autograd.Function):
    """y = x;  dL/dx = factor * dL/dy."""

    @staticmethod
    def forward(ctx, x, factor):
        ctx.factor = float(factor)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out.mul(ctx.factor), None


class GradientReversalFunction:
    """Kept for callers that use the reference's function object directly."""

    @staticmethod
    def apply(x, lambda_):
        return _ScaleGradient.apply(x, -float(lambda_))


class GradientReversal(torch.nn.Module):
    def __init__(self, lambda_=1):
        super().__init__()
        self.lambda_ = lambda_

    def forward(self, x):
        return _ScaleGradient.apply(x, -float(self.lambda_))

    def extra_repr(self):
        return f"lambda_={self.lambda_}"
