"""GPU neighbour lists in the reference's graph format (replaces the offline ase path of
lcaonet/data/convert.py:103-172 for data that already lives on the device).

`neighbor_list(pos, lattice, cutoff, ...)` returns edge_index [centre, neighbour] and integer-valued edge_shift with
edges grouped by centre and ordered by distance, truncated to `max_neighbors` per centre (convert.py:135-148), with
the fully-linked fallback of data/utils.py:10-20 for graphs in which no atom has any neighbour."""
from __future__ import annotations

import ctypes
import math

import torch
from torch import Tensor

from .. import _lib, ops
from .graph import Batch
from .keys import GraphKeys


def _graph_table(lattice: Tensor, pbc: Tensor, cutoff: float) -> tuple[Tensor, int]:
    """(B, 32) float32/int32-punned records for the kernels + total number of bins."""
    B = lattice.shape[0]
    lat = lattice.double()
    inv = torch.linalg.inv(lat)
    # height of the cell along axis d = 1 / |column d of the inverse lattice|
    h = 1.0 / inv.norm(dim=1)
    nb = torch.clamp((h / cutoff).floor().long(), min=1)
    nb = torch.where(pbc.bool(), nb, torch.clamp(nb, min=1))
    reach = torch.ceil(cutoff / (h / nb) - 1e-9).long().clamp(min=1)
    nbins = nb.prod(dim=1)
    base = torch.cumsum(nbins, 0) - nbins
    rec_f = torch.zeros(B, 32, dtype=torch.float32, device=lattice.device)
    rec_f[:, 0:9] = lattice.reshape(B, 9).float()
    rec_f[:, 9:18] = inv.reshape(B, 9).float()
    rec_i = rec_f.view(torch.int32)
    rec_i[:, 18:21] = nb.int()
    rec_i[:, 21:24] = reach.int()
    rec_i[:, 24:27] = pbc.int()
    rec_i[:, 27] = base.int()
    return rec_f, int(nbins.sum().item())


def _link_isolated_graphs(pos: Tensor, b: Tensor, n_graphs: int, e_src, e_dst, e_shift, e_dist):
    """Fully-linked fallback of data/utils.py:10-20 for every graph of the batch that has atoms but no edge
    (pairs ordered by (centre, neighbour) as the reference's meshgrid gives them); plain torch, any device."""
    n_atoms = torch.bincount(b, minlength=n_graphs)
    n_edges = torch.bincount(b[e_src], minlength=n_graphs)
    lonely = (n_edges == 0) & (n_atoms > 1)
    if not bool(lonely.any()):
        return e_src, e_dst, e_shift, e_dist
    nodes = lonely[b].nonzero(as_tuple=True)[0]                  # atoms of the graphs to link, ascending
    bb = b[nodes]
    same = bb.unsqueeze(0) == bb.unsqueeze(1)
    same.fill_diagonal_(False)
    i, j = same.nonzero(as_tuple=True)                           # row-major: (centre, neighbour) ascending
    a_src, a_dst = nodes[i], nodes[j]
    a_dist = (pos[a_dst] - pos[a_src]).norm(dim=1)
    src = torch.cat([e_src, a_src])
    o = torch.argsort(src, stable=True)                          # keeps edges grouped by centre
    return (src[o], torch.cat([e_dst, a_dst])[o],
            torch.cat([e_shift, torch.zeros(a_src.numel(), 3, device=pos.device, dtype=e_shift.dtype)])[o],
            torch.cat([e_dist, a_dist])[o])


@torch.no_grad()
def neighbor_list(pos: Tensor, lattice: Tensor, cutoff: float, batch: Tensor | None = None, pbc: Tensor | None = None,
                  max_neighbors: int | None = 32):
    """pos (N,3) float32 on the GPU, lattice (B,3,3), batch (N) graph ids (None = one graph), pbc (B,3) bool
    (None = fully periodic).  Returns (edge_index (2,E) int64, edge_shift (E,3) float32, edge_dist (E) float32)."""
    ops._dev(pos, "pos")
    dev = pos.device
    pos = pos.float().contiguous()
    lattice = lattice.float().reshape(-1, 3, 3).contiguous()
    B, N = lattice.shape[0], pos.shape[0]
    pbc = torch.ones(B, 3, dtype=torch.bool, device=dev) if pbc is None else pbc.to(dev).bool().reshape(B, 3)
    batch = None if batch is None else batch.to(torch.int64).contiguous()
    table, n_bins = _graph_table(lattice, pbc, cutoff)
    lib = _lib.load()
    p = ops._p
    ws_bytes = int(lib.lcao_nl_workspace_bytes(N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    wpos = torch.empty(N, 3, device=dev)
    wrap = torch.empty(N, 3, dtype=torch.int32, device=dev)
    atom_of = torch.empty(N, dtype=torch.int32, device=dev)
    bin_ptr = torch.empty(n_bins + 1, dtype=torch.int32, device=dev)
    ops._call("lcao_nl_bin_sort", p(pos), p(batch), p(table), N, n_bins, p(wpos), p(wrap), p(atom_of), p(bin_ptr), p(ws),
              ws_bytes, ops._stream())
    count = torch.empty(N, dtype=torch.int32, device=dev)
    ops._call("lcao_nl_search", p(wpos), p(wrap), p(batch), p(table), p(atom_of), p(bin_ptr), N, float(cutoff), 0,
              p(count), None, None, None, None, None, ops._stream())
    cnt = count.long()
    offset = torch.cumsum(cnt, 0) - cnt
    E = int(cnt.sum().item())
    e_src = torch.empty(E, dtype=torch.int64, device=dev)
    e_dst = torch.empty(E, dtype=torch.int64, device=dev)
    e_shift = torch.empty(E, 3, device=dev)
    e_dist = torch.empty(E, device=dev)
    if E:
        ops._call("lcao_nl_search", p(wpos), p(wrap), p(batch), p(table), p(atom_of), p(bin_ptr), N, float(cutoff), 1,
                  p(count), p(offset), p(e_src), p(e_dst), p(e_shift), p(e_dist), ops._stream())
    # per-centre distance order (candidates of a centre are contiguous): stable sort by distance, then by centre
    o = torch.argsort(e_dist, stable=True)
    o = o[torch.argsort(e_src[o], stable=True)]
    e_src, e_dst, e_shift, e_dist = e_src[o], e_dst[o], e_shift[o], e_dist[o]
    if max_neighbors is not None and E:
        rank = torch.arange(E, device=dev) - offset[e_src]
        keep = rank < max_neighbors
        e_src, e_dst, e_shift, e_dist = e_src[keep], e_dst[keep], e_shift[keep], e_dist[keep]
    # data/convert.py:154-157 (per structure): a graph in which no atom has a neighbour becomes fully linked, zero shifts
    b = batch if batch is not None else torch.zeros(N, dtype=torch.long, device=dev)
    e_src, e_dst, e_shift, e_dist = _link_isolated_graphs(pos, b, B, e_src, e_dst, e_shift, e_dist)
    return torch.stack([e_src, e_dst]), e_shift, e_dist


@torch.no_grad()
def make_graph(z: Tensor, pos: Tensor, lattice: Tensor, cutoff: float, batch: Tensor | None = None,
               pbc: Tensor | None = None, max_neighbors: int | None = 32) -> Batch:
    """Batch with the keys LCAONet.forward reads, built entirely on the device."""
    ei, shift, _ = neighbor_list(pos, lattice, cutoff, batch, pbc, max_neighbors)
    lattice = lattice.float().reshape(-1, 3, 3)
    if batch is None:
        batch = torch.zeros(z.numel(), dtype=torch.long, device=pos.device)
    B = lattice.shape[0]
    pbc = torch.ones(B, 3, dtype=torch.long, device=pos.device) if pbc is None else pbc.to(pos.device).long().reshape(B, 3)
    g = Batch(z=z, pos=pos.float().contiguous(), edge_index=ei.contiguous(), edge_shift=shift.contiguous(),
              lattice=lattice.contiguous(), batch=batch)
    g["num_graphs"] = int(B)                                                  # PyG's Batch.num_graphs
    g[GraphKeys.PBC] = pbc                                                    # convert.py:168-169
    g[GraphKeys.Neighbors] = torch.bincount(batch[ei[0]], minlength=B)
    return g
