"""The two non-torch activations the reference accepts by name (lcaonet/nn/activation.py): `Swish` (x * sigmoid(beta x),
beta optionally learnable) and `ShiftedSoftplus` (softplus(x) - log 2).  They are off the kernel hot path: the fused HIP
kernels implement the model default SiLU; these run as element-wise ops between the unfused HIP dense kernels."""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn


class Swish(nn.Module):
    def __init__(self, beta: float = 1.0, train_beta: bool = True):
        super().__init__()
        self.train_beta = bool(train_beta)
        value = torch.tensor(float(beta))
        # "beta" appears in the state_dict either way: as a parameter when learnable, as a buffer when frozen
        if self.train_beta:
            self.beta = nn.Parameter(value)
        else:
            self.register_buffer("beta", value)

    def forward(self, x: Tensor) -> Tensor:
        return torch.sigmoid(x * self.beta) * x

    def extra_repr(self) -> str:
        return "beta={:.2f}, trainable_beta:{}".format(float(self.beta), self.train_beta)


class ShiftedSoftplus(nn.Module):
    """ln(1 + e^x) - ln 2, so that the activation passes through the origin (SchNet)."""

    LOG2 = math.log(2.0)

    def __init__(self, shift: float | None = None):
        super().__init__()
        self.shift = self.LOG2 if shift is None else float(shift)

    def forward(self, x: Tensor) -> Tensor:
        return nn.functional.softplus(x) - self.shift

    def extra_repr(self) -> str:
        return "shift={:.2f}".format(self.shift)
