"""Angular basis on triplet cosines -> (T, n_orb): real Y_l^0 only (reference: lcaonet/nn/shbf.py:14-87).
Y_l^0(c) = sqrt((2l+1)/4pi) P_l(c); the kernel holds the l <= 3 closed forms."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..atomistic.info import ElecInfo


class SphericalHarmonicsBasis(nn.Module):
    def __init__(self, elec_info: ElecInfo):
        super().__init__()
        self.elec_info = elec_info
        self.register_buffer("l_of_d", elec_info.nl_list[:, 1].to(torch.int32).contiguous(), persistent=False)

    def extra_repr(self) -> str:
        return "elec_info={}(max_z={}, n_orb={}, n_per_orb={})".format(
            self.elec_info.__class__.__name__, self.elec_info.max_z, self.elec_info.n_orb, self.elec_info.n_per_orb)

    def forward(self, costheta: Tensor) -> Tensor:
        return ops.sph_basis(costheta, self.l_of_d)
