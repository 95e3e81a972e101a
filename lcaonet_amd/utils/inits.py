"""Weight initialisers the reference pulls from torch_geometric.nn.inits (glorot, glorot_orthogonal;
used via lcaonet/utils/resolve.py:108-123).  Restated from their published definitions."""
from __future__ import annotations

import math

import torch


def glorot(value):
    if isinstance(value, torch.Tensor):
        stdv = math.sqrt(6.0 / (value.size(-2) + value.size(-1)))
        value.data.uniform_(-stdv, stdv)
    return value


def glorot_orthogonal(tensor, scale):
    """Orthogonal init rescaled so that Var(W) = scale / (fan_in + fan_out)  (DimeNet's initialiser)."""
    if tensor is not None:
        torch.nn.init.orthogonal_(tensor.data)
        s = scale / ((tensor.size(-2) + tensor.size(-1)) * tensor.var())
        tensor.data *= s.sqrt()
    return tensor
