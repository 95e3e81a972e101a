"""Swish(beta) and ShiftedSoftplus (reference: lcaonet/nn/activation.py:7-60).  Off the hot path: the fused
kernels implement SiLU (the model default, lcaonet.py:345); other activations run as elementwise ops
between the unfused HIP dense kernels."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch import Tensor


class Swish(nn.Module):
    def __init__(self, beta: float = 1.0, train_beta: bool = True):
        super().__init__()
        self.train_beta = train_beta
        if train_beta:
            self.beta = nn.Parameter(torch.tensor(float(beta)))
        else:
            self.register_buffer("beta", torch.tensor(float(beta)))

    def extra_repr(self) -> str:
        return f"beta={self.beta.item():.2f}, trainable_beta:{self.train_beta}"

    def forward(self, x: Tensor) -> Tensor:
        return x * torch.sigmoid(self.beta * x)


class ShiftedSoftplus(nn.Module):
    def __init__(self, shift: float = math.log(2.0)):
        super().__init__()
        self.shift = shift

    def extra_repr(self) -> str:
        return f"shift={self.shift:.2f}"

    def forward(self, x: Tensor) -> Tensor:
        return torch.nn.functional.softplus(x) - self.shift
