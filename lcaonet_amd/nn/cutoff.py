"""Radial cutoff envelopes (reference: lcaonet/nn/cutoff.py:10-67), evaluated by the HIP kernel.

The reference's expanded polynomials cancel catastrophically towards rc; both have a triple root at x = 1 and are
evaluated in factored form (csrc/lcao_common.h):  1 + a x^p + b x^(p+1) + c x^(p+2) = (1 - x)^3 sum_{k<p} C(k+2,2) x^k
and 1 - 6x^5 + 15x^4 - 10x^3 = (1 - x)^3 (1 + 3x + 6x^2).  `differentiable(r)` is the same function from tensor ops
under autograd (any order): the twice-differentiable force-training path (model/double.py) uses it."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops


class BaseCutoff(nn.Module):
    kind: str | None = None  # kernel enum name; None = evaluated by the subclass' own forward()

    def __init__(self, cutoff: float):
        super().__init__()
        self.cutoff = cutoff

    def extra_repr(self) -> str:
        return f"cutoff={self.cutoff}"

    def forward(self, r: Tensor) -> Tensor:
        if self.kind is None:
            raise NotImplementedError
        return ops.cutoff(r.reshape(-1), self.kind, self.cutoff).reshape(r.shape)

    def differentiable(self, r: Tensor) -> Tensor:
        """The envelope from tensor ops (autograd-visible); subclasses without one fall back to forward()."""
        return self.forward(r)


def _factored(x: Tensor, n_terms: int) -> Tensor:
    """(1 - x)^3 * sum_{k < n_terms} C(k+2, 2) x^k  (Horner)."""
    q = torch.full_like(x, float(math.comb(n_terms + 1, 2)))
    for k in range(n_terms - 2, -1, -1):
        q = q * x + float(math.comb(k + 2, 2))
    om = 1.0 - x
    return om * om * om * q


class PolynomialCutoff(BaseCutoff):
    """1 - 6x^5 + 15x^4 - 10x^3, x = r/rc; 0 beyond rc."""
    kind = "polynomial"

    def differentiable(self, r: Tensor) -> Tensor:
        return torch.where(r <= self.cutoff, _factored(r / self.cutoff, 3), torch.zeros_like(r))


class CosineCutoff(BaseCutoff):
    """0.5 (cos(pi r / rc) + 1) = cos^2(pi r / 2 rc); 0 beyond rc."""
    kind = "cosine"

    def differentiable(self, r: Tensor) -> Tensor:
        c = torch.cos(r * (0.5 * math.pi / self.cutoff))
        return torch.where(r <= self.cutoff, c * c, torch.zeros_like(r))


class EnvelopeCutoff(BaseCutoff):
    """DimeNet envelope 1 + a x^p + b x^(p+1) + c x^(p+2).  The kernel implements the reference default p = 5; other
    exponents are evaluated with tensor ops (and multiplied onto the envelope-free radial basis)."""
    kind = "envelope"

    def __init__(self, cutoff: float, p: int = 5):
        super().__init__(cutoff)
        assert p > 0
        self.p = p
        self.a = -(p + 1) * (p + 2) / 2
        self.b = p * (p + 2)
        self.c = -p * (p + 1) / 2
        if p != 5:
            self.kind = None   # not a kernel mode: forward() below, applied by the radial basis as a user envelope

    def forward(self, r: Tensor) -> Tensor:
        if self.kind is not None:
            return super().forward(r)
        return self.differentiable(r)

    def differentiable(self, r: Tensor) -> Tensor:
        return torch.where(r <= self.cutoff, _factored(r / self.cutoff, self.p), torch.zeros_like(r))
