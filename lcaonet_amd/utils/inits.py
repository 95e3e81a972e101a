"""Weight initialisers the reference pulls from torch_geometric.nn.inits (glorot, glorot_orthogonal;
used via lcaonet/utils/resolve.py:108-123).  Restated from their published definitions."""
from __future__ import annotations

import math

import torch


def glorot(value):
    if isinstance(value, torch.Tensor):
        stdv = math.sqrt(6.0 / (value.size(-2) + value.size(-1)))
        # in place on the tensor itself (not through .data): the write bumps its version counter, which is what the
        # weight-derived caches of the model (species tables, folded node weights) are keyed on
        with torch.no_grad():
            value.uniform_(-stdv, stdv)
    return value


def glorot_orthogonal(tensor, scale):
    """Orthogonal init rescaled so that Var(W) = scale / (fan_in + fan_out)  (DimeNet's initialiser)."""
    if tensor is not None:
        with torch.no_grad():     # on the tensor itself, see glorot
            torch.nn.init.orthogonal_(tensor)
            s = scale / ((tensor.size(-2) + tensor.size(-1)) * tensor.var())
            tensor.mul_(s.sqrt())
    return tensor
