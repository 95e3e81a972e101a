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
    def __init__(self, emb_size: int, max_z: int = 94):
        super().__init__()
        self.emb_size = emb_size
        self.z_embed = nn.Embedding(max_z, emb_size)
        self.reset_parameters()

    def reset_parameters(self):
        self.z_embed.weight.data.uniform_(-math.sqrt(3), math.sqrt(3))

    def forward(self, z: Tensor) -> Tensor:
        return self.z_embed(z - 1)


class EmbedElec(nn.Module):
    """Per-orbital occupation embeddings; orbitals above `min_orb` get padding_idx=0 (a zero vector for an
    empty orbital) unless extend_orb."""

    def __init__(self, emb_size: int, elec_info: ElecInfo, extend_orb: bool = False):
        super().__init__()
        self.register_buffer("elec", elec_info.elec_table)
        self.n_orb = elec_info.n_orb
        self.emb_size = emb_size
        self.extend_orb = extend_orb
        self.e_embeds = nn.ModuleList()
        min_idx = elec_info.min_orb_idx if elec_info.min_orb_idx else -1
        for i, max_e in enumerate(elec_info.max_elec_idx):
            padding_idx = None if (i <= min_idx or extend_orb) else 0
            self.e_embeds.append(nn.Embedding(int(max_e), emb_size, padding_idx=padding_idx))
        self.reset_parameters()

    def reset_parameters(self):
        for ee in self.e_embeds:
            ee.weight.data.uniform_(-math.sqrt(2), math.sqrt(2))
            ee._fill_padding_idx_with_zero()

    def forward(self, z: Tensor) -> Tensor:
        elec = self.elec[z]  # (N, n_orb)
        return torch.stack([ce(elec[:, i]) for i, ce in enumerate(self.e_embeds)], dim=1)  # (N, n_orb, emb)


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
