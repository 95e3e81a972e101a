"""Radial cutoff envelopes (reference: lcaonet/nn/cutoff.py:10-67), evaluated by the HIP kernel."""
from __future__ import annotations

import torch.nn as nn
from torch import Tensor

from .. import ops


class BaseCutoff(nn.Module):
    kind: str | None = None  # kernel enum name; None = user subclass evaluated by its own forward()

    def __init__(self, cutoff: float):
        super().__init__()
        self.cutoff = cutoff

    def extra_repr(self) -> str:
        return f"cutoff={self.cutoff}"

    def forward(self, r: Tensor) -> Tensor:
        if self.kind is None:
            raise NotImplementedError
        return ops.cutoff(r.reshape(-1), self.kind, self.cutoff).reshape(r.shape)


class PolynomialCutoff(BaseCutoff):
    """1 - 6x^5 + 15x^4 - 10x^3, x = r/rc; 0 beyond rc."""
    kind = "polynomial"


class CosineCutoff(BaseCutoff):
    """0.5 (cos(pi r / rc) + 1); 0 beyond rc."""
    kind = "cosine"


class EnvelopeCutoff(BaseCutoff):
    """DimeNet envelope 1 + a x^p + b x^(p+1) + c x^(p+2); the kernel implements the reference default p = 5."""
    kind = "envelope"

    def __init__(self, cutoff: float, p: int = 5):
        super().__init__(cutoff)
        assert p > 0
        if p != 5:
            raise NotImplementedError("EnvelopeCutoff: only p = 5 (the reference default) is built into the kernel")
        self.p = p
        self.a = -(p + 1) * (p + 2) / 2
        self.b = p * (p + 2)
        self.c = -p * (p + 1) / 2
