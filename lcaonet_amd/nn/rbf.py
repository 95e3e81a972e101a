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


def orbital_table(nl_list, r0: float = 0.529, norms: list[float] | None = None) -> Tensor:
    """One 12-float row per orbital (include/lcaonet_hip.h: LCAO_ORB_STRIDE).  norms: per-orbital prefactors replacing
    the all-space normalisation C_nl (integral_norm=True)."""
    rows = []
    for i, (n, l) in enumerate(nl_list):
        n, l = int(n), int(l)
        c = laguerre_coeffs(n, l)
        if len(c) > 7 or l > 3:
            raise ValueError(f"orbital (n={n}, l={l}) outside the kernel's table (n-l-1 <= 6, l <= 3)")
        norm = -math.sqrt((2.0 / n / r0) ** 3 * math.factorial(n - l - 1) / 2.0 / n / math.factorial(n + l) ** 3)
        if norms is not None:
            norm = norms[i]
        rows.append([2.0 / n / r0, norm, float(l), float(len(c) - 1)] + [float(v) for v in c] + [0.0] * (7 - len(c))
                    + [math.pi * n])
    return torch.tensor(rows, dtype=torch.float64).to(torch.float32)


def _sphere_norms(nl_list, r0: float, rc: float, cutoff_net: BaseCutoff) -> list[float]:
    """integral_norm=True (rbf.py:88-89,107-127): prefactor -2/(n r0) times 1 / (sqrt(int_0^rc (r f(r))^2 dr) + 1e-12), the
    integral of the cutoff-weighted orbital over the cutoff sphere by adaptive quadrature (fp64)."""
    from scipy.integrate import quad

    out = []
    for n, l in nl_list:
        n, l = int(n), int(l)
        c = laguerre_coeffs(n, l)
        pre = -2.0 / n / r0

        def f(r, n=n, l=l, c=c, pre=pre):
            zeta = 2.0 / n / r0 * r
            poly = sum(ci * zeta**i for i, ci in enumerate(c))
            cw = float(cutoff_net.differentiable(torch.tensor([r], dtype=torch.float64)))
            return cw * pre * poly * zeta**l * math.exp(-zeta / 2.0)

        integral = quad(lambda r: (r * f(r)) ** 2, 0.0, rc)[0]
        out.append(pre / (math.sqrt(integral) + 1e-12))
    return out


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

    def differentiable(self, r: Tensor) -> Tensor:
        """The same basis as a twice-differentiable function of r (the exact double backward of force training,
        model/double.py, needs R, R' and R''): one kernel per derivative level (csrc/lcao_double.hip::k_radial_levels,
        second-order jets of the closed forms).  A user-defined envelope multiplies the envelope-free basis through its
        own `differentiable`."""
        kind = getattr(self.cutoff_net, "kind", None)
        rb = _RadialFn.apply(r.contiguous(), self.orb_table, self.rbf_kind, kind if kind is not None else "none", self.cutoff)
        return rb if kind is not None else rb * self.cutoff_net.differentiable(r).unsqueeze(1)


class _RadialFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, tab, rbf_kind, cutoff_kind, rc):
        ctx.args = (tab, rbf_kind, cutoff_kind, rc)
        ctx.save_for_backward(r)
        return ops.radial_levels(0, r, *ctx.args)

    @staticmethod
    def backward(ctx, g):
        (r,) = ctx.saved_tensors
        return _RadialBwdFn.apply(g.contiguous(), r, *ctx.args), None, None, None, None


class _RadialBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, r, tab, rbf_kind, cutoff_kind, rc):
        ctx.args = (tab, rbf_kind, cutoff_kind, rc)
        ctx.save_for_backward(g, r)
        return ops.radial_levels(1, r, *ctx.args, g=g)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, h):
        g, r = ctx.saved_tensors
        gg, gr2 = ops.radial_levels(2, r, *ctx.args, g=g, h=h.contiguous())
        return gg, gr2, None, None, None, None


class HydrogenRadialBasis(BaseRadialBasis):
    rbf_kind = "hydrogen"

    def __init__(self, cutoff: float, elec_info: ElecInfo, cutoff_net: BaseCutoff, bohr_radius: float = 0.529,
                 integral_norm: bool = False):
        super().__init__(cutoff, elec_info, cutoff_net)
        self.bohr_radius = bohr_radius
        self.integral_norm = integral_norm
        nl = elec_info.nl_list.tolist()
        norms = _sphere_norms(nl, bohr_radius, cutoff, cutoff_net) if integral_norm else None
        self.register_buffer("orb_table", orbital_table(nl, bohr_radius, norms), persistent=False)


class SphericalBesselRadialBasis(BaseRadialBasis):
    """cutoff(r) * sin(n pi r / rc) / r   (depends on n only; rbf.py:165-182)."""
    rbf_kind = "bessel"

    def __init__(self, cutoff: float, elec_info: ElecInfo, cutoff_net: BaseCutoff):
        super().__init__(cutoff, elec_info, cutoff_net)
        self.register_buffer("orb_table", orbital_table(elec_info.nl_list.tolist()), persistent=False)
