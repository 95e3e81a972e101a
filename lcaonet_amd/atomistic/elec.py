"""Ground-state electron configurations and valence flags for Z = 1..96 in the reference's orbital order
(lcaonet/atomistic/elec.py:1-233).  The tables are GENERATED here (aufbau filling + the known anomalous
configurations; valence shells by period block) and checked element by element against the golden copy
of the reference's tables in tests/golden/atomistic_tables.npz."""
from __future__ import annotations

import torch

# 1s 2s 2p 3s 3p 4s 3d 4p 5s 4d 5p 6s 4f 5d 6p 7s 5f 6d  (Madelung order, as NL_LIST elec.py:6-27)
ORB_NAMES = ["1s", "2s", "2p", "3s", "3p", "4s", "3d", "4p", "5s", "4d", "5p", "6s", "4f", "5d", "6p", "7s", "5f", "6d"]
N_ORB = len(ORB_NAMES)
_L = {"s": 0, "p": 1, "d": 2, "f": 3}
_CAP = {"s": 2, "p": 6, "d": 10, "f": 14}

NL_LIST = torch.tensor([[int(o[0]), _L[o[1]]] for o in ORB_NAMES])
# embedding vocabulary per orbital = capacity + 1 (elec.py:29)
MAX_ELEC_IDX = torch.tensor([_CAP[o[1]] + 1 for o in ORB_NAMES], dtype=torch.long)

# deviations from strict Madelung filling: z -> {orbital: occupation}
_ANOMALIES = {
    24: {"4s": 1, "3d": 5}, 29: {"4s": 1, "3d": 10},
    41: {"5s": 1, "4d": 4}, 42: {"5s": 1, "4d": 5}, 44: {"5s": 1, "4d": 7}, 45: {"5s": 1, "4d": 8},
    46: {"5s": 0, "4d": 10}, 47: {"5s": 1, "4d": 10},
    57: {"4f": 0, "5d": 1}, 58: {"4f": 1, "5d": 1}, 64: {"4f": 7, "5d": 1},
    78: {"6s": 1, "5d": 9}, 79: {"6s": 1, "5d": 10},
    89: {"5f": 0, "6d": 1}, 90: {"5f": 0, "6d": 2}, 91: {"5f": 2, "6d": 1}, 92: {"5f": 3, "6d": 1},
    93: {"5f": 4, "6d": 1}, 96: {"5f": 7, "6d": 1},
}

# valence orbitals by range of Z (elec.py:133-233)
_VALENCE = [
    (1, 2, "1s"), (3, 4, "2s"), (5, 10, "2s 2p"), (11, 12, "3s"), (13, 18, "3s 3p"), (19, 20, "4s"),
    (21, 30, "4s 3d"), (31, 36, "3d 4p"), (37, 38, "5s"), (39, 48, "5s 4d"), (49, 54, "4d 5p"), (55, 56, "6s"),
    (57, 57, "6s 5d"), (58, 58, "6s 4f 5d"), (59, 63, "6s 4f"), (64, 64, "6s 4f 5d"), (65, 70, "6s 4f"),
    (71, 77, "4f 5d"), (78, 79, "6s 4f 5d"), (80, 80, "4f 5d"), (81, 86, "5d 6p"), (87, 88, "7s"),
    (89, 90, "7s 6d"), (91, 93, "7s 5f 6d"), (94, 95, "7s 5f"), (96, 96, "7s 5f 6d"),
]
MAX_Z = 96


def _build():
    elec = torch.zeros(MAX_Z + 1, N_ORB, dtype=torch.long)
    for z in range(1, MAX_Z + 1):
        left = z
        for i, o in enumerate(ORB_NAMES):
            n = min(left, _CAP[o[1]])
            elec[z, i] = n
            left -= n
        for o, n in _ANOMALIES.get(z, {}).items():
            elec[z, ORB_NAMES.index(o)] = n
        assert int(elec[z].sum()) == z, z
    val = torch.zeros(MAX_Z + 1, N_ORB, dtype=torch.long)
    for lo, hi, orbs in _VALENCE:
        for o in orbs.split():
            val[lo:hi + 1, ORB_NAMES.index(o)] = 1
    return elec, val


ELEC_TABLE, VALENCE_TABLE = _build()
