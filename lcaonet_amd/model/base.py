"""BaseMPNN: edge geometry under periodic boundary conditions (reference: lcaonet/model/base.py:10-47)."""
from __future__ import annotations

import torch.nn as nn

from .. import ops
from ..data.keys import GraphKeys


class BaseMPNN(nn.Module):
    def __init__(self) -> None:
        super().__init__()

    @staticmethod
    def calc_atomic_distances(graph, return_vec: bool = False):
        """Writes `edge_dist` (E) and, if asked, the unit vectors `edge_vec` (E, 3) into the graph."""
        idx_s, idx_t = graph[GraphKeys.Edge_idx]
        dist, uvec = ops.edge_geometry(graph[GraphKeys.Pos], idx_s, idx_t, graph[GraphKeys.Edge_shift],
                                       graph[GraphKeys.Lattice], graph.get(GraphKeys.Batch_idx))
        graph[GraphKeys.Edge_dist] = dist
        if return_vec:
            graph[GraphKeys.Edge_vec_st] = uvec
        return graph

    @property
    def n_param(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
