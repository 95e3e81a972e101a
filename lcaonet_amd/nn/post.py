"""PostProcess: add per-atom reference values and the mean offset (reference: lcaonet/nn/post.py:8-89)."""
from __future__ import annotations

import torch.nn as nn
from torch import Tensor

from .scatter import scatter


class PostProcess(nn.Module):
    def __init__(self, out_dim: int, is_extensive: bool = True, atomref: Tensor | None = None,
                 mean: Tensor | None = None):
        super().__init__()
        self.out_dim = out_dim
        self.is_extensive = is_extensive
        self.register_buffer("atomref", atomref)
        self.register_buffer("mean", mean)

    def extra_repr(self) -> str:
        return f"is_extensive={self.is_extensive}, atomref={self.atomref}, mean={self.mean}"

    def forward(self, out, z: Tensor, batch_idx: Tensor | None, n_graphs: int | None = None, batch_plan=None):
        out, force = out if isinstance(out, tuple) else (out, None)
        red = "sum" if self.is_extensive else "mean"
        if self.atomref is not None:
            aref = self.atomref[z]
            if batch_idx is None:
                aref = aref.sum(0, keepdim=True) if self.is_extensive else aref.mean(0, keepdim=True)
            else:
                aref = scatter(aref, batch_idx, dim=0, dim_size=n_graphs, reduce=red,
                               plan=batch_plan if n_graphs is not None else None)
            out = out + aref
        if self.mean is not None:
            mean = self.mean.unsqueeze(0)
            if self.is_extensive:
                mean = mean.expand(z.size(0), -1)
                mean = mean.sum(0, keepdim=True) if batch_idx is None else \
                    scatter(mean.contiguous(), batch_idx, dim=0, dim_size=n_graphs, reduce="sum",
                            plan=batch_plan if n_graphs is not None else None)
            out = out + mean
        if force is not None:
            return out, force
        return out
