"""Radial bases on edge distances -> (E, n_orb)  (reference: lcaonet/nn/rbf.py:17-182).

The closed forms are re-derived here from the associated-Laguerre series (no sympy at run time):
    R_nl(r) = cutoff(r) * C_nl * P_nl(zeta) * zeta^l * exp(-zeta/2),   zeta = 2 r / (n r0)
    P_nl(x) = -(n+l)! * L^{2l+1}_{n-l-1}(x) = sum_i  -(-1)^i C(n+l, n-l-1-i) (n+l)!/i!  x^i     (integers)
    C_nl    = -sqrt( (2/(n r0))^3 (n-l-1)! / (2n (n+l)!^3) )
and handed to the HIP kernel as one 12-float row per orbital."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..atomistic.info import ElecInfo
from .cutoff import BaseCutoff


def laguerre_coeffs(n: int, l: int) -> list[int]:
    k, a = n - l - 1, 2 * l + 1
    return [-((-1) ** i) * math.comb(k + a, k - i) * (math.factorial(n + l) // math.factorial(i)) for i in range(k + 1)]


def orbital_table(nl_list, r0: float = 0.529) -> Tensor:
    rows = []
    for n, l in nl_list:
        n, l = int(n), int(l)
        c = laguerre_coeffs(n, l)
        if len(c) > 7 or l > 3:
            raise ValueError(f"orbital (n={n}, l={l}) outside the kernel's table (n-l-1 <= 6, l <= 3)")
        norm = -math.sqrt((2.0 / n / r0) ** 3 * math.factorial(n - l - 1) / 2.0 / n / math.factorial(n + l) ** 3)
        rows.append([2.0 / n / r0, norm, float(l), float(len(c) - 1)] + [float(v) for v in c] + [0.0] * (7 - len(c))
                    + [math.pi * n])
    return torch.tensor(rows, dtype=torch.float64).to(torch.float32)


class BaseRadialBasis(nn.Module):
    rbf_kind: str | None = None

    def __init__(self, cutoff: float, elec_info: ElecInfo, cutoff_net: BaseCutoff):
        super().__init__()
        self.cutoff = cutoff
        self.elec_info = elec_info
        self.cutoff_net = cutoff_net
        self.n_orb = elec_info.n_orb

    def extra_repr(self) -> str:
        return "cutoff={}, elec_info={}(max_z={}, n_orb={}, n_per_orb={})".format(
            self.cutoff, self.elec_info.__class__.__name__, self.elec_info.max_z, self.elec_info.n_orb,
            self.elec_info.n_per_orb)

    def _eval(self, r: Tensor) -> Tensor:
        kind = getattr(self.cutoff_net, "kind", None)
        if kind is not None:
            return ops.radial_basis(r, self.orb_table, self.rbf_kind, kind, self.cutoff)
        # user-defined BaseCutoff subclass: basis without envelope from the kernel, times the user's envelope
        rb = ops.radial_basis(r, self.orb_table, self.rbf_kind, "none", self.cutoff)
        return rb * self.cutoff_net(r).unsqueeze(1)

    def forward(self, r: Tensor) -> Tensor:
        return self._eval(r)


class HydrogenRadialBasis(BaseRadialBasis):
    rbf_kind = "hydrogen"

    def __init__(self, cutoff: float, elec_info: ElecInfo, cutoff_net: BaseCutoff, bohr_radius: float = 0.529,
                 integral_norm: bool = False):
        super().__init__(cutoff, elec_info, cutoff_net)
        if integral_norm:
            raise NotImplementedError("integral_norm=True (unreachable through LCAONet) is not built yet")
        self.bohr_radius = bohr_radius
        self.integral_norm = integral_norm
        self.register_buffer("orb_table", orbital_table(elec_info.nl_list.tolist(), bohr_radius), persistent=False)


class SphericalBesselRadialBasis(BaseRadialBasis):
    """cutoff(r) * sin(n pi r / rc) / r   (depends on n only; rbf.py:165-182)."""
    rbf_kind = "bessel"

    def __init__(self, cutoff: float, elec_info: ElecInfo, cutoff_net: BaseCutoff):
        super().__init__(cutoff, elec_info, cutoff_net)
        self.register_buffer("orb_table", orbital_table(elec_info.nl_list.tolist()), persistent=False)
