"""ElecInfo: slices the static tables by (max_z, max_orb, min_orb, n_per_orb).
Same constructor, properties and errors as the reference's lcaonet/atomistic/info.py:8-127."""
from __future__ import annotations

from torch import Tensor

from .elec import ELEC_TABLE, MAX_ELEC_IDX, NL_LIST, ORB_NAMES, VALENCE_TABLE

# last orbital occupied by any element up to max_z (info.py:39-75)
_Z_TO_LAST_ORB = [(2, 0), (4, 1), (10, 2), (12, 3), (18, 4), (20, 5), (30, 6), (36, 7), (38, 8), (48, 9), (54, 10),
                  (56, 11), (80, 13), (86, 14), (88, 15), (96, 17)]


class ElecInfo:
    def __init__(self, max_z: int, max_orb: str | None, min_orb: str | None = None, n_per_orb: int = 1):
        self.max_z = max_z
        self.max_orb = max_orb
        self.min_orb = min_orb
        self.n_per_orb = n_per_orb
        self._min_orb_idx = self._get_orb_idx_byorb(min_orb) if min_orb else None
        self._max_orb_idx = self._get_orb_idx_byz(max_z)
        if max_orb is not None:
            self._max_orb_idx = max(self._max_orb_idx, self._get_orb_idx_byorb(max_orb))
        self._n_orb = (self._max_orb_idx + 1) * self.n_per_orb

    @staticmethod
    def _get_orb_idx_byz(max_z: int) -> int:
        if max_z <= 0:
            raise ValueError(f"max_z={max_z} is too small.")
        for z_hi, idx in _Z_TO_LAST_ORB:
            if max_z <= z_hi:
                return idx
        raise ValueError(f"max_z={max_z} is too large.")

    @staticmethod
    def _get_orb_idx_byorb(orb: str) -> int:
        if orb not in ORB_NAMES:
            raise ValueError(f"max_orb={orb} is not supported.")
        return ORB_NAMES.index(orb)

    @property
    def n_orb(self) -> int:
        return self._n_orb

    @property
    def min_orb_idx(self) -> int | None:
        if self._min_orb_idx is None:
            return None
        return self._min_orb_idx * self.n_per_orb + (self.n_per_orb - 1)

    def _slice(self, table: Tensor) -> Tensor:
        return table[: self.max_z + 1, : self._max_orb_idx + 1].repeat_interleave(self.n_per_orb, dim=1)

    @property
    def elec_table(self) -> Tensor:
        return self._slice(ELEC_TABLE)

    @property
    def valence_table(self) -> Tensor:
        return self._slice(VALENCE_TABLE)

    @property
    def max_elec_idx(self) -> Tensor:
        return MAX_ELEC_IDX[: self._max_orb_idx + 1].repeat_interleave(self.n_per_orb, dim=0)

    @property
    def nl_list(self) -> Tensor:
        return NL_LIST[: self._max_orb_idx + 1].repeat_interleave(self.n_per_orb, dim=0)
