"""BaseMPNN: edge geometry under periodic boundary conditions (reference: lcaonet/model/base.py:10-47)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..data.keys import GraphKeys


def num_graphs(graph, batch_idx) -> int | None:
    """B = batch.max() + 1 (reference lcaonet.py:292) without stalling the stream on every call: PyG's `Batch.num_graphs`
    when the graph object carries one (the containers of lcaonet_amd.data do), else ONE `batch.max().item()` whose
    result is remembered in the graph for as long as it holds the same batch tensor."""
    if batch_idx is None:
        return None
    n = getattr(graph, "num_graphs", None)
    if isinstance(n, int):
        return n
    memo = graph.get("_num_graphs_memo") if hasattr(graph, "get") else None
    tag = (batch_idx.data_ptr(), batch_idx._version, batch_idx.numel())
    if memo is not None and memo[0] == tag:
        return memo[1]
    n = int(batch_idx.max().item()) + 1 if batch_idx.numel() else 0
    try:
        graph["_num_graphs_memo"] = (tag, n)
    except Exception:  # noqa: BLE001  (a read-only container: pay the sync each call)
        pass
    return n


def batch_plan(graph, batch_idx, n_graphs):
    """The sorted-segment structure of the batch index (nn/scatter.py::index_plan) for the per-graph readout, remembered in
    the graph for as long as it holds the same batch tensor — the readout, atomref and mean sums of one forward, and every
    later forward on the same graph, share one sort.  None without a batch index."""
    from ..nn.scatter import index_plan

    if batch_idx is None:
        return None
    tag = (batch_idx.data_ptr(), batch_idx._version, batch_idx.numel(), n_graphs)
    memo = graph.get("_batch_plan_memo") if hasattr(graph, "get") else None
    if memo is not None and memo[0] == tag:
        return memo[1]
    plan = index_plan(batch_idx, n_graphs)
    if not torch.cuda.is_current_stream_capturing():     # tensors made inside a capture live in the graph's pool
        try:
            graph["_batch_plan_memo"] = (tag, plan)
        except Exception:  # noqa: BLE001  (a read-only container: sort on each call)
            pass
    return plan


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
