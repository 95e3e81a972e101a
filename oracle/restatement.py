"""CPU restatement of the reference's per-edge interaction hot path  —  TEST INFRASTRUCTURE ONLY.

This file restates, with plain PyTorch CPU ops in the reference's own *literal* dataflow
(materialised triplet tensors, gather + bmm + scatter), the algorithm of
nmdl-mizo/lcaonet v0.0.3 for the path named in BASELINE.json.  It is the checker for the
HIP kernels and the `cpu_baseline` leg of bench.py.  Nothing under `lcaonet_amd/` may import
it; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg do.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here against
  (a) golden vectors dumped from the reference itself (imported from /root/reference in the
      build container by `oracle/make_golden.py`; fixtures in tests/golden/*.npz), and
  (b) the reference's own known-answer tests, restated: closed-form hydrogen R_nl
      (reference tests/nn/test_rbf.py:37-48) and scipy Y_l^0 (tests/nn/test_shbf.py:46).
The three third-party packages the reference needs but the image lacks were replaced by the
stand-ins in oracle/standins/ *only while generating (a)*; what that means for each call
site is recorded in DESIGN.md ("Oracle").

Each function cites the reference file:line it follows (paths relative to /root/reference).
The state_dict keys are the reference's own.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F
from torch import Tensor

# ----------------------------------------------------------------------------------------
# atomistic tables  (lcaonet/atomistic/elec.py:6-27, info.py:39-127)
# ----------------------------------------------------------------------------------------
# (n, l) per orbital in aufbau order 1s 2s 2p 3s 3p 4s 3d 4p 5s 4d 5p 6s 4f 5d 6p 7s 5f 6d
ORB_NAMES = ["1s", "2s", "2p", "3s", "3p", "4s", "3d", "4p", "5s", "4d", "5p", "6s", "4f", "5d", "6p", "7s", "5f", "6d"]
_L_OF = {"s": 0, "p": 1, "d": 2, "f": 3}
NL = [(int(o[0]), _L_OF[o[1]]) for o in ORB_NAMES]
# last orbital index reached by max_z (info.py:39-75)
_Z_BREAKS = [(2, 0), (4, 1), (10, 2), (12, 3), (18, 4), (20, 5), (30, 6), (36, 7), (38, 8), (48, 9), (54, 10),
             (56, 11), (80, 13), (86, 14), (88, 15), (96, 17)]


def max_orb_idx(max_z: int, max_orb: str | None = None) -> int:
    """info.py:28-33,39-75,77-101."""
    if max_z <= 0:
        raise ValueError(f"max_z={max_z} is too small.")
    for zmax, idx in _Z_BREAKS:
        if max_z <= zmax:
            break
    else:
        raise ValueError(f"max_z={max_z} is too large.")
    if max_orb is not None:
        if max_orb not in ORB_NAMES:
            raise ValueError(f"max_orb={max_orb} is not supported.")
        idx = max(idx, ORB_NAMES.index(max_orb))
    return idx


def nl_list(max_z: int, max_orb: str | None = None, n_per_orb: int = 1) -> list[tuple[int, int]]:
    """info.py:126-127: NL_LIST[:idx+1].repeat_interleave(n_per_orb)."""
    out = []
    for nl in NL[: max_orb_idx(max_z, max_orb) + 1]:
        out += [nl] * n_per_orb
    return out


# ----------------------------------------------------------------------------------------
# cutoffs  (lcaonet/nn/cutoff.py)
# ----------------------------------------------------------------------------------------
def cutoff_fn(r: Tensor, rc: float, kind: str = "envelope", p: int = 5) -> Tensor:
    if kind == "envelope":  # cutoff.py:46-67 (p = 5 is the default LCAONet uses)
        a = -(p + 1) * (p + 2) / 2
        b = p * (p + 2)
        c = -p * (p + 1) / 2
        x = r / rc
        val = 1 + a * x**p + b * x ** (p + 1) + c * x ** (p + 2)
        return torch.where(r <= rc, val, torch.zeros_like(x))
    if kind == "polynomial":  # cutoff.py:32-34
        x = r / rc
        return torch.where(r <= rc, 1 - 6 * x**5 + 15 * x**4 - 10 * x**3, 0.0)
    if kind == "cosine":  # cutoff.py:41-43
        val = 0.5 * (torch.cos(r * math.pi / rc) + 1.0)
        return torch.where(r <= rc, val, 0.0)
    raise ValueError(kind)


# ----------------------------------------------------------------------------------------
# radial bases  (lcaonet/nn/rbf.py)
# ----------------------------------------------------------------------------------------
def laguerre_poly_coeffs(n: int, l: int) -> list[int]:
    """Integer coefficients c_i (ascending powers) of
        P_nl(x) = L^{2l+1}_{n-l-1}(x) * (n+l)! * (-1)^{2l+1}          (rbf.py:82-87)
    from the closed form  L_k^a(x) = sum_i (-1)^i C(k+a, k-i) x^i / i!  (Abramowitz-Stegun 22.3.9).
    (n+l)!/i! * C(n+l, n-l-1-i) is an integer, so the table is exact.
    """
    k, a = n - l - 1, 2 * l + 1
    coeffs = []
    for i in range(k + 1):
        c = math.comb(k + a, k - i) * (math.factorial(n + l) // math.factorial(i))
        coeffs.append(-((-1) ** i) * c)
    return coeffs


def hydrogen_norm(n: int, l: int, r0: float = 0.529) -> float:
    """rbf.py:92-94 (integral_norm=False branch)."""
    return -math.sqrt((2.0 / n / r0) ** 3 * math.factorial(n - l - 1) / 2.0 / n / math.factorial(n + l) ** 3)


def sphere_norm(n: int, l: int, rc: float, cutoff: str = "envelope", r0: float = 0.529, p: int = 5) -> float:
    """rbf.py:88-89,107-127 (integral_norm=True): the prefactor is -2/(n r0) and the orbital is rescaled by
    1 / (sqrt(int_0^rc (r f(r))^2 dr) + 1e-12), the integral by scipy.integrate.quad over the cutoff-weighted orbital."""
    from scipy.integrate import quad

    c = laguerre_poly_coeffs(n, l)
    pre = -2.0 / n / r0

    def f(x: float) -> float:
        zeta = 2.0 / n / r0 * x
        cw = float(cutoff_fn(torch.tensor([x], dtype=torch.float64), rc, cutoff, p))
        return cw * pre * sum(ci * zeta**i for i, ci in enumerate(c)) * zeta**l * math.exp(-zeta / 2.0)

    integral = quad(lambda x: (x * f(x)) ** 2, 0.0, rc)[0]
    sc = 1 / (torch.sqrt(torch.tensor([integral])) + 1e-12)       # a float32 tensor in the reference (rbf.py:127)
    return pre * float(sc)


def hydrogen_radial_basis(r: Tensor, nl: list[tuple[int, int]], rc: float, cutoff: str = "envelope",
                          r0: float = 0.529, integral_norm: bool = False, p: int = 5) -> Tensor:
    """rbf.py:96-103,129-142.  The lambdified sympy polynomial is evaluated in expanded
    monomial form, highest power first, with x**k — kept that way here."""
    cols = []
    cw = cutoff_fn(r, rc, cutoff, p)
    for n, l in nl:
        zeta = 2.0 / n / r0 * r
        c = laguerre_poly_coeffs(n, l)
        poly = None
        for i in range(len(c) - 1, -1, -1):
            if i == 0:
                term = torch.full_like(zeta, float(c[0])) if poly is None else float(c[0])
            elif i == 1:
                term = c[1] * zeta
            else:
                term = c[i] * zeta**i
            poly = term if poly is None else poly + term
        norm = sphere_norm(n, l, rc, cutoff, r0, p) if integral_norm else hydrogen_norm(n, l, r0)
        cols.append(cw * norm * poly * torch.pow(zeta, l) * torch.exp(-zeta / 2.0))
    return torch.stack(cols, dim=1)


def bessel_radial_basis(r: Tensor, nl: list[tuple[int, int]], rc: float, cutoff: str = "envelope") -> Tensor:
    """rbf.py:165-182: cutoff(r) * sin(n*pi*r/rc) / r  (depends on n only)."""
    cw = cutoff_fn(r, rc, cutoff)
    return torch.stack([cw * (math.pi * n * r / rc).sin() / r for n, _ in nl], dim=1)


# ----------------------------------------------------------------------------------------
# angular basis  (lcaonet/nn/shbf.py): real Y_l^0(cos theta) = sqrt((2l+1)/4pi) P_l(c)
# ----------------------------------------------------------------------------------------
def sph_harm_l0(c: Tensor, l: int) -> Tensor:
    """shbf.py:28-42 (l=0) and :54-63 (sympy Znm, m=0, simplified & evalf'ed)."""
    if l == 0:
        return 0.5 * torch.ones_like(c) * math.sqrt(1.0 / math.pi)
    if l == 1:
        return math.sqrt(3.0 / (4 * math.pi)) * c
    if l == 2:
        k = math.sqrt(5.0 / (4 * math.pi))
        return 1.5 * k * c**2 - 0.5 * k
    if l == 3:
        k = math.sqrt(7.0 / (4 * math.pi))
        return 0.5 * k * c * (5 * c**2 - 3)
    raise ValueError(f"l={l} not used by the reference (l <= 3)")


def spherical_harmonics_basis(costheta: Tensor, nl: list[tuple[int, int]]) -> Tensor:
    """shbf.py:75-87."""
    return torch.stack([sph_harm_l0(costheta, l) for _, l in nl], dim=1)


# ----------------------------------------------------------------------------------------
# index / geometry prep
# ----------------------------------------------------------------------------------------
def scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """torch_scatter.scatter(..., reduce='sum') along dim 0 (call sites lcaonet.py:189,208,293,307)."""
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add_(0, index, src)


def scatter_mean(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    cnt = torch.zeros(dim_size, dtype=src.dtype).index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype))
    return scatter_sum(src, index, dim_size) / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))


def get_triplets(edge_index: Tensor, num_nodes: int) -> tuple[Tensor, Tensor, Tensor]:
    """lcaonet.py:439-486.  For every edge e=(s->t), in edge order: all edges e'=(k->s)
    (idx_t[e'] == idx_s[e]) ordered by k (ties in original edge order), minus e' == e.
    Returns (tri_idx_k, edge_idx_ks, edge_idx_st)."""
    idx_s, idx_t = edge_index
    E = idx_s.numel()
    # in-edge lists grouped by target node, ordered by (target, source, edge id)
    order = torch.argsort(idx_t * num_nodes + idx_s, stable=True)
    cnt = torch.bincount(idx_t, minlength=num_nodes)
    ptr = torch.zeros(num_nodes + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(cnt, 0)
    n_in = cnt[idx_s]  # in-degree of the centre of each edge
    e_st = torch.repeat_interleave(torch.arange(E), n_in)
    start = torch.cumsum(n_in, 0) - n_in
    offs = torch.arange(e_st.numel()) - start[e_st]
    e_ks = order[ptr[idx_s[e_st]] + offs]
    mask = e_ks != e_st
    e_ks, e_st = e_ks[mask], e_st[mask]
    return idx_s[e_ks], e_ks, e_st


def atomic_distances(pos: Tensor, edge_index: Tensor, edge_shift: Tensor, lattice: Tensor, batch: Tensor | None):
    """model/base.py:15-43."""
    src, dst = edge_index
    if batch is None:
        batch = torch.zeros(pos.shape[0], dtype=torch.long)
    vec = pos[dst] - pos[src] + torch.einsum("ni,nij->nj", edge_shift, lattice[batch[src]])
    dist = torch.norm(vec, dim=1)
    return dist, vec / dist.unsqueeze(-1)


# ----------------------------------------------------------------------------------------
# model pieces, functional over a reference state_dict
# ----------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """Constructor arguments of lcaonet.model.LCAONet (lcaonet.py:327-351), code names."""
    emb_size: int = 128
    emb_size_coeff: int = 128
    emb_size_conv: int = 128
    out_size: int = 1
    n_interaction: int = 3
    n_per_orb: int = 1
    cutoff: float = 6.0
    rbf_type: str = "hydrogen"
    cutoff_net: str = "envelope"
    max_z: int = 36
    min_orb: str | None = None
    max_orb: str | None = None
    elec_to_node: bool = True
    add_valence: bool = False
    extend_orb: bool = False
    is_extensive: bool = True
    regress_forces: bool = False
    direct_forces: bool = True
    training: bool = False  # BatchNorm mode
    extra: dict = field(default_factory=dict)


def _act(cfg):
    """The activation module LCAONet builds once and shares between all its MLPs (lcaonet.py:372 via
    utils/resolve.py:66-76): torch.nn activations by name plus lcaonet/nn/activation.py:36-60 (ShiftedSoftplus =
    softplus(x) - ln 2).  Constructor argument `activation`, default "SiLU"; carried in cfg.extra."""
    name = str(cfg.extra.get("activation", "SiLU")).lower()
    table = {"silu": F.silu, "shiftedsoftplus": lambda x: F.softplus(x) - math.log(2.0), "tanh": torch.tanh,
             "elu": F.elu, "relu": F.relu, "sigmoid": torch.sigmoid, "softplus": F.softplus}
    if name not in table:
        raise ValueError(f"activation {name} not restated")
    return table[name]


def _lin(x: Tensor, sd: dict, key: str) -> Tensor:
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


def _bn(x: Tensor, sd: dict, key: str, training: bool) -> Tensor:
    """nn.BatchNorm1d, eps=1e-5 (embed.py:175,232)."""
    if training:
        return F.batch_norm(x, None, None, sd[key + ".weight"], sd[key + ".bias"], True, 0.0, 1e-5)
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"],
                        sd[key + ".bias"], False, 0.0, 1e-5)


def embedding(sd: dict, cfg: OracleConfig, z: Tensor, idx_s: Tensor, idx_t: Tensor) -> tuple[Tensor, Tensor]:
    """LCAOEmbedding.forward lcaonet.py:59-74 with embed.py:41 (EmbedZ), :83-99 (EmbedElec),
    :178-194 (EmbedNode), :235-249 (EmbedCoeffs)."""
    p = "emb_layer."
    H, C = cfg.emb_size, cfg.emb_size_coeff
    z_embed = sd[p + "z_embed.z_embed.weight"][z - 1]
    node_z, coeff_z = torch.split(z_embed, [H, C], dim=-1)
    elec = sd[p + "e_embed.elec"][z]  # (N, D)
    D = elec.shape[1]
    e_embed = torch.stack([sd[p + f"e_embed.e_embeds.{d}.weight"][elec[:, d]] for d in range(D)], dim=1)  # (N,D,*)
    if cfg.elec_to_node:
        node_e, coeff_e = torch.split(e_embed, [H, C], dim=-1)
        node_in = torch.cat([node_z, node_e.sum(1) / math.sqrt(D)], dim=-1)
    else:
        coeff_e = e_embed
        node_in = node_z
    act = _act(cfg)
    h = act(_lin(node_in, sd, p + "node_embed.f_enc.0"))
    h = act(_lin(h, sd, p + "node_embed.f_enc.2"))
    x = _bn(h, sd, p + "node_embed.bn", cfg.training)
    # coefficients
    zz = _lin(torch.cat([coeff_z[idx_s], coeff_z[idx_t]], dim=-1), sd, p + "coeff_embed.f_z.0")
    fe = act(_lin(coeff_e, sd, p + "coeff_embed.f_e.0"))
    fe = act(_lin(fe, sd, p + "coeff_embed.f_e.2"))[idx_t]
    coeff = fe + fe * zz.unsqueeze(1)
    cst = _bn(coeff.reshape(coeff.size(0), -1), sd, p + "coeff_embed.bn", cfg.training).reshape(coeff.size())
    return x, cst


def interaction(sd: dict, prefix: str, cfg: OracleConfig, x: Tensor, cst: Tensor, valence_mask: Tensor | None,
                rb: Tensor, shb: Tensor, idx_s: Tensor, idx_t: Tensor, tri_idx_k: Tensor, edge_idx_ks: Tensor,
                edge_idx_st: Tensor, trace: dict | None = None) -> Tensor:
    """LCAOInteraction.forward lcaonet.py:130-216, literal dataflow."""
    if cfg.add_valence and valence_mask is None:
        raise ValueError("valence_mask must be provided when add_valence=True")
    x_before = x
    N = x.size(0)
    x, xk = torch.chunk(_lin(x, sd, prefix + "node_weight"), 2, dim=-1)  # :167
    act = _act(cfg)
    cst = act(_lin(cst, sd, prefix + "f_coeffs.0"))  # :170
    cst = act(_lin(cst, sd, prefix + "f_coeffs.2"))
    cks = cst[edge_idx_ks]  # :173
    if cfg.add_valence:
        cks, cks_valence = torch.chunk(cks, 2, dim=-1)
        cks_valence = cks_valence * valence_mask[edge_idx_ks]
    three_body_orbs = rb[edge_idx_ks] * shb  # :179
    three_body_w = torch.einsum("ed,edh->eh", three_body_orbs, cks).contiguous()
    if cfg.add_valence:
        three_body_w = three_body_w + torch.einsum("ed,edh->eh", three_body_orbs, cks_valence).contiguous()
    three_body_w = F.normalize(three_body_w, dim=-1)  # :184
    three_body_w = three_body_w * torch.sigmoid(xk[tri_idx_k])  # :187-188
    three_body_w = scatter_sum(three_body_w, edge_idx_st, rb.size(0))  # :189
    if trace is not None:
        trace["three_body_w"] = three_body_w
    cst = cst + cst * _lin(three_body_w, sd, prefix + "f_three.0").unsqueeze(1)  # :192
    if cfg.add_valence:
        cst, cst_valence = torch.chunk(cst, 2, dim=-1)
        cst_valence = cst_valence * valence_mask
    lcao_w = torch.einsum("ed,edh->eh", rb, cst).contiguous()  # :200
    if cfg.add_valence:
        lcao_w = lcao_w + torch.einsum("ed,edh->eh", rb, cst_valence).contiguous()
    lcao_w = F.normalize(lcao_w, dim=-1)  # :204
    if trace is not None:
        trace["lcao_w"] = lcao_w
    h = act(_lin(torch.cat([x[idx_s], x[idx_t]], dim=-1), sd, prefix + "f_node.0"))  # :209
    h = act(_lin(h, sd, prefix + "f_node.2"))
    msg = _lin(lcao_w, sd, prefix + "basis_weight") * h
    return x_before + _lin(scatter_sum(msg, idx_s, N), sd, prefix + "out_weight")  # :207-214


def out_layer(sd: dict, cfg: OracleConfig, x: Tensor, batch: Tensor | None, idx_s: Tensor, idx_t: Tensor,
              edge_vec: Tensor, n_graphs: int | None = None):
    """LCAOOut.forward lcaonet.py:271-319 (autograd forces: handled by the caller)."""
    p = "out_layer.out_lin."
    act = _act(cfg)
    prop = act(_lin(x, sd, p + "0"))
    prop = act(_lin(prop, sd, p + "2"))
    prop = _lin(prop, sd, p + "4")
    if batch is not None:
        B = int(batch.max().item()) + 1 if n_graphs is None else n_graphs
        prop = scatter_sum(prop, batch, B) if cfg.is_extensive else scatter_mean(prop, batch, B)
    else:
        prop = prop.sum(0, keepdim=True) if cfg.is_extensive else prop.mean(0, keepdim=True)
    if not (cfg.regress_forces and cfg.direct_forces):
        return prop
    p = "out_layer.out_lin_force."
    f = act(_lin(torch.cat([x[idx_s], x[idx_t]], dim=-1), sd, p + "0"))
    f = act(_lin(f, sd, p + "2"))
    f = _lin(f, sd, p + "4") * edge_vec
    return prop, scatter_sum(f, idx_s, x.size(0))


def post_process(sd: dict, cfg: OracleConfig, out: Tensor, z: Tensor, batch: Tensor | None) -> Tensor:
    """PostProcess.forward post.py:44-89."""
    atomref, mean = sd.get("pp_layer.atomref"), sd.get("pp_layer.mean")
    B = out.size(0)
    if atomref is not None:
        aref = atomref[z]
        if batch is None:
            aref = aref.sum(0, keepdim=True) if cfg.is_extensive else aref.mean(0, keepdim=True)
        else:
            aref = scatter_sum(aref, batch, B) if cfg.is_extensive else scatter_mean(aref, batch, B)
        out = out + aref
    if mean is not None:
        m = mean.unsqueeze(0)
        if cfg.is_extensive:
            m = m.expand(z.size(0), -1)
            m = m.sum(0, keepdim=True) if batch is None else scatter_sum(m, batch, B)
        out = out + m
    return out


def forward(sd: dict, cfg: OracleConfig, graph: dict, dtype=torch.float32, trace: dict | None = None):
    """LCAONet.forward lcaonet.py:488-540.  `graph` is a plain dict with the GraphKeys names
    (data/keys.py:4-27).  All float tensors are cast to `dtype` (fp64 gives the conditioning reference)."""
    sd = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}
    z = graph["z"]
    pos = graph["pos"].to(dtype)
    edge_index = graph["edge_index"]
    idx_s, idx_t = edge_index
    batch = graph.get("batch")
    autograd_forces = cfg.regress_forces and not cfg.direct_forces
    if autograd_forces:
        pos = pos.detach().requires_grad_(True)
    tri_idx_k, e_ks, e_st = get_triplets(edge_index, z.size(0))
    dist, vec = atomic_distances(pos, edge_index, graph["edge_shift"].to(dtype), graph["lattice"].to(dtype), batch)
    costheta = (vec[e_st] * vec[e_ks]).sum(-1)  # lcaonet.py:432-433
    nl = nl_list(cfg.max_z, cfg.max_orb, cfg.n_per_orb)
    if cfg.rbf_type == "hydrogen":
        rb = hydrogen_radial_basis(dist, nl, cfg.cutoff, cfg.cutoff_net)
    else:
        rb = bessel_radial_basis(dist, nl, cfg.cutoff, cfg.cutoff_net)
    shb = spherical_harmonics_basis(costheta, nl)
    x, cst = embedding(sd, cfg, z, idx_s, idx_t)
    vmask = None
    if cfg.add_valence:  # embed.py:122-133
        V = cfg.emb_size_conv
        vmask = sd["valence_mask.valence"][z].to(dtype).unsqueeze(-1).expand(-1, -1, V)[idx_t]
    if trace is not None:
        trace.update(edge_dist=dist, edge_vec=vec, idx_k_3b=tri_idx_k, edge_idx_ks_3b=e_ks, edge_idx_st_3b=e_st,
                     angles_3b=costheta, rb=rb, shb=shb, x0=x, cst=cst)
    for i in range(cfg.n_interaction):
        tr = {} if trace is not None else None
        x = interaction(sd, f"int_layers.{i}.", cfg, x, cst, vmask, rb, shb, idx_s, idx_t, tri_idx_k, e_ks, e_st, tr)
        if trace is not None:
            trace[f"x{i + 1}"] = x
            for k, v in tr.items():
                trace[f"{k}{i + 1}"] = v
    out = out_layer(sd, cfg, x, batch, idx_s, idx_t, vec)  # B = batch.max()+1 as lcaonet.py:292
    if autograd_forces:  # lcaonet.py:310-317
        prop = out
        if cfg.out_size > 1:
            force = torch.stack([-torch.autograd.grad(prop[:, i].sum(), pos, create_graph=True)[0]
                                 for i in range(cfg.out_size)], dim=1)
        else:
            force = -torch.autograd.grad(prop.sum(), pos, create_graph=True)[0]
        out = (prop, force)
    if isinstance(out, tuple):
        return post_process(sd, cfg, out[0], z, batch), out[1]
    return post_process(sd, cfg, out, z, batch)
