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
    def __init__(self, emb_size: int, emb_size_z: int, use_elec: bool, emb_size_e: int | None = None,
                 activation: nn.Module = nn.SiLU(), weight_init: Callable[[Tensor], Tensor] | None = None):
        super().__init__()
        self.emb_size = emb_size
        self.emb_size_z = emb_size_z
        self.use_elec = use_elec
        if use_elec:
            assert emb_size_e is not None
            self.emb_size_e = emb_size_e
        else:
            self.emb_size_e = 0
        hid_size = max(emb_size, (emb_size_z + self.emb_size_e) // 2)
        self.f_enc = nn.Sequential(
            Dense(emb_size_z + self.emb_size_e, hid_size, True, weight_init), activation,
            Dense(hid_size, emb_size, True, weight_init), activation,
        )
        self.bn = nn.BatchNorm1d(emb_size)

    def forward(self, z_emb: Tensor, e_emb: Tensor | None = None) -> Tensor:
        if self.use_elec:
            if e_emb is None:
                raise ValueError("e_emb must be set when use_elec is True.")
            e_emb = e_emb.sum(1) / math.sqrt(e_emb.shape[1])
            z_e_emb = torch.cat([z_emb, e_emb], dim=-1)
        else:
            z_e_emb = z_emb
        return self.bn(run_mlp(self.f_enc, z_e_emb))


class EmbedCoeffs(nn.Module):
    def __init__(self, emb_size: int, emb_size_z: int, emb_size_e: int, n_orb: int,
                 activation: nn.Module = nn.SiLU(), weight_init: Callable[[Tensor], Tensor] | None = None):
        super().__init__()
        self.emb_size = emb_size
        self.emb_size_z = emb_size_z
        self.emb_size_e = emb_size_e
        self.f_z = nn.Sequential(Dense(2 * emb_size_z, emb_size, False, weight_init))
        self.f_e = nn.Sequential(
            Dense(emb_size_e, emb_size, False, weight_init), activation,
            Dense(emb_size, emb_size, False, weight_init), activation,
        )
        self.bn = nn.BatchNorm1d(emb_size * n_orb)

    def forward(self, z_emb: Tensor, e_emb: Tensor, idx_s: Tensor, idx_t: Tensor) -> Tensor:
        z_emb = run_mlp(self.f_z, torch.cat([z_emb[idx_s], z_emb[idx_t]], dim=-1))
        e_emb = run_mlp(self.f_e, e_emb)[idx_t]
        coeff_emb = e_emb + e_emb * z_emb.unsqueeze(1)
        return self.bn(coeff_emb.reshape(coeff_emb.size(0), -1)).reshape(coeff_emb.size())
