"""ElecInfo: the window of the static orbital tables a model uses, chosen by (max_z, max_orb, min_orb, n_per_orb).

Public contract of the reference's lcaonet/atomistic/info.py:8-127 (constructor arguments kept as attributes, the
properties n_orb / min_orb_idx / elec_table / valence_table / max_elec_idx / nl_list, the ValueError texts).  The
internals are this package's own: one table lookup for the orbital count and one slicing helper."""
from __future__ import annotations

import bisect

from torch import Tensor

from .elec import ELEC_TABLE, MAX_ELEC_IDX, NL_LIST, ORB_NAMES, VALENCE_TABLE

# Highest atomic number whose ground state stops at orbital index i (Madelung filling order of elec.py's columns):
# He closes 1s, Be 2s, Ne 2p, ...  Orbitals 12 (4f) and 16 (5f) never close a period on their own — elements up to Hg
# (80) reach 5d, index 13, and up to Cm (96) reach 6d, index 17 — hence the two jumps.  (info.py:39-75 as data.)
_PERIOD_END_Z = (2, 4, 10, 12, 18, 20, 30, 36, 38, 48, 54, 56, 80, 86, 88, 96)
_PERIOD_END_ORB = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 17)


def _last_orbital_of(max_z: int) -> int:
    if max_z <= 0:
        raise ValueError(f"max_z={max_z} is too small.")
    slot = bisect.bisect_left(_PERIOD_END_Z, max_z)
    if slot == len(_PERIOD_END_Z):
        raise ValueError(f"max_z={max_z} is too large.")
    return _PERIOD_END_ORB[slot]


def _orbital_index(name: str) -> int:
    try:
        return ORB_NAMES.index(name)
    except ValueError:
        raise ValueError(f"max_orb={name} is not supported.") from None


class ElecInfo:
    def __init__(self, max_z: int, max_orb: str | None, min_orb: str | None = None, n_per_orb: int = 1):
        self.max_z, self.max_orb, self.min_orb, self.n_per_orb = max_z, max_orb, min_orb, n_per_orb
        top = _last_orbital_of(max_z)
        if max_orb is not None:
            top = max(top, _orbital_index(max_orb))
        self._n_base = top + 1                                            # orbitals before the n_per_orb repetition
        self._floor = _orbital_index(min_orb) if min_orb else None        # `min_orb` as an un-repeated orbital index

    def _window(self, table: Tensor, orbital_dim: int) -> Tensor:
        """The first n_base orbitals of a static table along `orbital_dim`, each repeated n_per_orb times."""
        return table.narrow(orbital_dim, 0, self._n_base).repeat_interleave(self.n_per_orb, dim=orbital_dim)

    @property
    def n_orb(self) -> int:
        return self._n_base * self.n_per_orb

    @property
    def min_orb_idx(self) -> int | None:
        """Index, after the repetition, of the LAST copy of `min_orb` (None without one)."""
        return None if self._floor is None else (self._floor + 1) * self.n_per_orb - 1

    @property
    def elec_table(self) -> Tensor:
        return self._window(ELEC_TABLE[: self.max_z + 1], 1)

    @property
    def valence_table(self) -> Tensor:
        return self._window(VALENCE_TABLE[: self.max_z + 1], 1)

    @property
    def max_elec_idx(self) -> Tensor:
        return self._window(MAX_ELEC_IDX, 0)

    @property
    def nl_list(self) -> Tensor:
        return self._window(NL_LIST, 0)
