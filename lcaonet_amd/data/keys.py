"""Names of the graph attributes (same strings as the reference's lcaonet/data/keys.py:4-27)."""


class GraphKeys:
    Lattice = "lattice"  # (B, 3, 3)
    PBC = "pbc"  # (B, 3)
    Neighbors = "neighbors"  # (B)

    Batch_idx = "batch"  # (N)
    Z = "z"  # (N)
    Pos = "pos"  # (N, 3)

    # keys containing "index" are offset by the node count when graphs are batched
    Edge_idx = "edge_index"  # (2, E)  [centre, neighbour]
    Edge_shift = "edge_shift"  # (E, 3)
    Edge_dist = "edge_dist"  # (E)
    Edge_vec_st = "edge_vec"  # (E, 3)

    # three-body properties
    Idx_s_3b = "idx_s_3b"
    Idx_t_3b = "idx_t_3b"
    Idx_k_3b = "idx_k_3b"
    Edge_idx_ks_3b = "edge_idx_ks_3b"
    Edge_idx_st_3b = "edge_idx_st_3b"
    Angles_3b = "angles_3b"


KEYS = [v for k, v in vars(GraphKeys).items() if not k.startswith("__")]
