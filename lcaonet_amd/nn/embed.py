"""Atomic-number / electron-count embeddings (reference: lcaonet/nn/embed.py:14-249).

Off the kernel hot path (runs once per forward): lookups and BatchNorm stay PyTorch-ROCm
ops, the Dense layers run the HIP dense kernel.  Parameter and buffer names are the reference's, so its
checkpoints load with strict=True.  `LCAOEmbedding.species_tables()` (model/lcaonet.py) collapses these
modules to per-species tables for the fused `lcao_embed_coeffs` kernel in eval mode."""
from __future__ import annotations

import math
from collections.abc import Callable

import torch
import torch.nn as nn
from torch import Tensor

from ..atomistic.info import ElecInfo
from .. import ops
from .base import Dense, run_mlp


class EmbedZ(nn.Module):
    """Atomic number -> vector.  One row per element 1..max_z (the lookup index is z - 1)."""

    def __init__(self, emb_size: int, max_z: int = 94):
        super().__init__()
        self.emb_size = emb_size
        self.z_embed = nn.Embedding(max_z, emb_size)
        self.reset_parameters()

    def reset_parameters(self):
        bound = math.sqrt(3.0)
        with torch.no_grad():
            self.z_embed.weight.uniform_(-bound, bound)

    def forward(self, z: Tensor) -> Tensor:
        return self.z_embed(z - 1)


class EmbedElec(nn.Module):
    """Ground-state occupation of every orbital -> vector, one lookup table per orbital.

    An orbital the element does not occupy maps to the zero vector (padding_idx = 0) unless `extend_orb` is set or
    the orbital lies at or below `min_orb` (embed.py:63-71)."""

    def __init__(self, emb_size: int, elec_info: ElecInfo, extend_orb: bool = False):
        super().__init__()
        self.emb_size, self.extend_orb, self.n_orb = emb_size, extend_orb, elec_info.n_orb
        self.register_buffer("elec", elec_info.elec_table)
        always_on = -1 if not elec_info.min_orb_idx else elec_info.min_orb_idx
        tables = []
        for orb, vocab in enumerate(elec_info.max_elec_idx.tolist()):
            pad = None if (extend_orb or orb <= always_on) else 0
            tables.append(nn.Embedding(vocab, emb_size, padding_idx=pad))
        self.e_embeds = nn.ModuleList(tables)
        self.reset_parameters()

    def reset_parameters(self):
        bound = math.sqrt(2.0)
        with torch.no_grad():
            for table in self.e_embeds:
                table.weight.uniform_(-bound, bound)
                if table.padding_idx is not None:
                    table.weight[table.padding_idx].zero_()

    def forward(self, z: Tensor) -> Tensor:
        occ = self.elec[z]                                          # (N, n_orb) electron counts
        cols = [table(occ[:, orb]) for orb, table in enumerate(self.e_embeds)]
        return torch.stack(cols, dim=1)                             # (N, n_orb, emb)


class ValenceMask(nn.Module):
    def __init__(self, emb_size: int, elec_info: ElecInfo):
        super().__init__()
        self.register_buffer("valence", elec_info.valence_table)
        self.n_orb = elec_info.n_orb
        self.emb_size = emb_size

    def forward(self, z: Tensor, idx_j: Tensor) -> Tensor:
        return self.valence[z].unsqueeze(-1).expand(-1, -1, self.emb_size)[idx_j]

    def per_edge(self, z: Tensor, idx_j: Tensor) -> Tensor:
        """(E, n_orb) float mask: what the fused kernels consume (the (E, n_orb, V) view above is constant along V)."""
        return self.valence[z][idx_j].to(torch.float32).contiguous()


class EmbedNode(nn.Module):
    """Initial node state: BatchNorm(MLP([z vector | orbital-pooled electron vector])) — reference embed.py:136-194
    (parameter names `f_enc.{0,2}`, `bn` are the checkpoint contract).  The electron vectors of the D orbitals are pooled
    as sum / sqrt(D).  At inference the whole module is evaluated on max_z rows and gathered
    (LCAOEmbedding.species_tables)."""

    def __init__(self, emb_size: int, emb_size_z: int, use_elec: bool, emb_size_e: int | None = None,
                 activation: nn.Module = nn.SiLU(), weight_init: Callable[[Tensor], Tensor] | None = None):
        super().__init__()
        if use_elec and emb_size_e is None:
            raise ValueError("emb_size_e is required when use_elec is True")
        self.emb_size, self.emb_size_z, self.use_elec = emb_size, emb_size_z, use_elec
        self.emb_size_e = emb_size_e if use_elec else 0
        width_in = emb_size_z + self.emb_size_e
        hidden = max(emb_size, width_in // 2)
        self.f_enc = nn.Sequential(Dense(width_in, hidden, True, weight_init), activation,
                                   Dense(hidden, emb_size, True, weight_init), activation)
        self.bn = nn.BatchNorm1d(emb_size)

    def encoder_input(self, z_emb: Tensor, e_emb: Tensor | None) -> Tensor:
        if not self.use_elec:
            return z_emb
        if e_emb is None:
            raise ValueError("e_emb must be set when use_elec is True.")
        pooled = torch.div(e_emb.sum(dim=1), math.sqrt(e_emb.shape[1]))
        return torch.cat([z_emb, pooled], dim=-1)

    def forward(self, z_emb: Tensor, e_emb: Tensor | None = None) -> Tensor:
        return self.bn(run_mlp(self.f_enc, self.encoder_input(z_emb, e_emb)))


class EmbedCoeffs(nn.Module):
    """Per-edge coefficient vectors cst (E, D, C) = BatchNorm(fe[t] + fe[t] * f_z([zc[s] | zc[t]])) — reference
    embed.py:197-249 (parameter names `f_z.0`, `f_e.{0,2}`, `bn`).  f_z is linear and bias-free, so it is applied to the
    N node rows as two half-width products (centre half, neighbour half) that are then added per edge, instead of to E
    concatenated rows — the same split lcao_embed_coeffs uses on the per-species tables."""

    def __init__(self, emb_size: int, emb_size_z: int, emb_size_e: int, n_orb: int,
                 activation: nn.Module = nn.SiLU(), weight_init: Callable[[Tensor], Tensor] | None = None):
        super().__init__()
        self.emb_size, self.emb_size_z, self.emb_size_e = emb_size, emb_size_z, emb_size_e
        self.f_z = nn.Sequential(Dense(2 * emb_size_z, emb_size, False, weight_init))
        self.f_e = nn.Sequential(Dense(emb_size_e, emb_size, False, weight_init), activation,
                                 Dense(emb_size, emb_size, False, weight_init), activation)
        self.bn = nn.BatchNorm1d(emb_size * n_orb)

    def node_halves(self, z_emb: Tensor) -> tuple[Tensor, Tensor]:
        """(N, C) contributions of a node to f_z as the centre / as the neighbour of an edge."""
        w = self.f_z[0].weight.detach()
        k = self.emb_size_z
        return ops.dense(z_emb, w[:, :k].contiguous()), ops.dense(z_emb, w[:, k:].contiguous())

    def forward(self, z_emb: Tensor, e_emb: Tensor, idx_s: Tensor, idx_t: Tensor) -> Tensor:
        as_centre, as_neigh = self.node_halves(z_emb)
        zz = as_centre[idx_s] + as_neigh[idx_t]                           # (E, C)
        orb = run_mlp(self.f_e, e_emb)[idx_t]                             # (E, D, C): the neighbour's orbital vectors
        raw = torch.addcmul(orb, orb, zz.unsqueeze(1))
        n_edges, n_orb, width = raw.shape
        return self.bn(raw.view(n_edges, n_orb * width)).view(n_edges, n_orb, width)
