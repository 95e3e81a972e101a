"""Graph attribute names — the strings the reference stores its graphs under (lcaonet/data/keys.py:4-27), generated
from one table so that `GraphKeys.<Name>` and `KEYS` cannot drift apart.

Keys whose string contains "index" are offset by the running node count when graphs are batched (PyG convention)."""

_TABLE = {
    # per graph
    "Lattice": "lattice",            # (B, 3, 3) cell vectors as rows
    "PBC": "pbc",                    # (B, 3) periodic flags
    "Neighbors": "neighbors",        # (B) number of edges of each graph
    # per node
    "Batch_idx": "batch",            # (N) graph id
    "Z": "z",                        # (N) atomic numbers, 1-based
    "Pos": "pos",                    # (N, 3)
    # per edge: edge_index = [centre, neighbour]
    "Edge_idx": "edge_index",        # (2, E)
    "Edge_shift": "edge_shift",      # (E, 3) integer-valued lattice translations
    "Edge_dist": "edge_dist",        # (E)    written by forward()
    "Edge_vec_st": "edge_vec",       # (E, 3) unit vectors, written by forward()
    # per triplet (only with materialize_triplets=True)
    "Idx_s_3b": "idx_s_3b",
    "Idx_t_3b": "idx_t_3b",
    "Idx_k_3b": "idx_k_3b",
    "Edge_idx_ks_3b": "edge_idx_ks_3b",
    "Edge_idx_st_3b": "edge_idx_st_3b",
    "Angles_3b": "angles_3b",
}

GraphKeys = type("GraphKeys", (), dict(_TABLE, __doc__="Namespace of graph attribute names."))
KEYS = list(_TABLE.values())
