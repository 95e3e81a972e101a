"""Deterministic synthetic crystals / molecules in the reference's graph format.

Produces what `lcaonet/data/convert.py:103-172` (ase neighbour list -> PyG Data) would hand to the
model: `edge_index = [centre, neighbour]`, integer-valued `edge_shift`, edges grouped by centre
ascending and sorted by distance inside each centre, optionally truncated to `max_neighbors`
(convert.py:135-148).  Everything is plain torch so it runs on CPU or on the GPU.

SURVEY.md §7 hard part 1: the reference is ill-conditioned for edges in the last ~1 % shell
below the model cutoff (envelope cancellation x F.normalize), so the generators build neighbour
lists with `cutoff_graph` (= 0.99 * model cutoff in every parity test and in the bench).
"""
from __future__ import annotations

import math

import torch

from .graph import Batch


def _neighbor_list_cells(pos: torch.Tensor, box: torch.Tensor, cutoff: float, max_neighbors: int | None,
                         periodic: bool, chunk: int = 64):
    """Brute-force neighbour list for B equally sized orthorhombic cells.

    pos (B, n, 3), box (B, 3) edge lengths.  Returns centre, neighbour (global node ids),
    shift (E, 3) float, ordered by (centre, distance)."""
    B, n, _ = pos.shape
    dev = pos.device
    if periodic:
        n_img = int(math.ceil(cutoff / float(box.min())))
        rng = torch.arange(-n_img, n_img + 1, device=dev)
        imgs = torch.stack(torch.meshgrid(rng, rng, rng, indexing="ij"), -1).reshape(-1, 3).to(pos.dtype)
    else:
        imgs = torch.zeros(1, 3, device=dev, dtype=pos.dtype)
    M = imgs.shape[0]
    cs, ns, ss, ds = [], [], [], []
    for b0 in range(0, B, chunk):
        p = pos[b0:b0 + chunk]  # (b, n, 3)
        bx = box[b0:b0 + chunk]
        b = p.shape[0]
        off = imgs.view(1, M, 3) * bx.view(b, 1, 3)  # (b, M, 3)
        # vec[b, i, j, m] = p[b, j] - p[b, i] + off[b, m]
        vec = p.view(b, 1, n, 1, 3) - p.view(b, n, 1, 1, 3) + off.view(b, 1, 1, M, 3)
        d = vec.norm(dim=-1)  # (b, n, n, M)
        mask = (d <= cutoff) & (d > 1e-6)
        bi, i, j, m = mask.nonzero(as_tuple=True)
        dd = d[bi, i, j, m]
        centre = (b0 + bi) * n + i
        # order by (centre, distance): stable sort by distance, then stable sort by centre
        o = torch.argsort(dd, stable=True)
        o = o[torch.argsort(centre[o], stable=True)]
        cs.append(centre[o]); ns.append(((b0 + bi) * n + j)[o]); ss.append(imgs[m][o]); ds.append(dd[o])
    centre, neigh, shift, dist = torch.cat(cs), torch.cat(ns), torch.cat(ss), torch.cat(ds)
    if max_neighbors is not None:
        cnt = torch.bincount(centre, minlength=B * n)
        start = torch.cumsum(cnt, 0) - cnt
        rank = torch.arange(centre.numel(), device=dev) - start[centre]
        keep = rank < max_neighbors
        centre, neigh, shift, dist = centre[keep], neigh[keep], shift[keep], dist[keep]
    return centre, neigh, shift


def _finish(pos, z, box, centre, neigh, shift, n_per_cell):
    B = box.shape[0]
    lattice = torch.diag_embed(box)  # (B, 3, 3)
    return Batch(
        z=z.reshape(-1),
        pos=pos.reshape(-1, 3).contiguous(),
        edge_index=torch.stack([centre, neigh]).contiguous(),
        edge_shift=shift.contiguous(),
        lattice=lattice.contiguous(),
        batch=torch.arange(B, device=pos.device).repeat_interleave(n_per_cell),
    )


def random_periodic_cells(n_cells: int, n_atoms: int, box: float, cutoff_graph: float, max_z: int = 36,
                          min_dist: float = 1.0, seed: int = 1234, max_neighbors: int | None = None,
                          device="cpu", dtype=torch.float32) -> Batch:
    """BASELINE config 1 shape: `n_cells` cubic cells of edge `box` with `n_atoms` uniformly random
    atoms each (minimum-image rejection below `min_dist`), z ~ U{1..max_z}."""
    g = torch.Generator().manual_seed(seed)
    pos = torch.empty(n_cells, n_atoms, 3, dtype=torch.float64)
    for c in range(n_cells):
        pts = []
        while len(pts) < n_atoms:
            cand = torch.rand(3, generator=g, dtype=torch.float64) * box
            ok = True
            for q in pts:
                dv = cand - q
                dv = dv - box * torch.round(dv / box)
                if float(dv.norm()) < min_dist:
                    ok = False
                    break
            if ok:
                pts.append(cand)
        pos[c] = torch.stack(pts)
    z = torch.randint(1, max_z + 1, (n_cells, n_atoms), generator=g)
    pos = pos.to(dtype).to(device)
    z = z.to(device)
    bx = torch.full((n_cells, 3), box, dtype=dtype, device=device)
    centre, neigh, shift = _neighbor_list_cells(pos, bx, cutoff_graph, max_neighbors, True)
    return _finish(pos, z, bx, centre, neigh, shift, n_atoms)


def jittered_cubic_cells(n_cells: int, n_side: int = 4, a: float = 3.2, sigma: float = 0.05,
                         cutoff_graph: float = 4.95, max_z: int = 36, seed: int = 1234,
                         max_neighbors: int | None = None, device="cpu", dtype=torch.float32) -> Batch:
    """BASELINE config 2/3 shape: `n_cells` cells, each a jittered n_side^3 simple-cubic lattice
    (spacing `a`, gaussian jitter sigma*a), z ~ U{1..max_z}.  n_cells=1000, n_side=4, a=3.2,
    cutoff 4.95 gives N=64 000 and about 1.15 M edges (about 18 neighbours)."""
    g = torch.Generator().manual_seed(seed)
    n = n_side**3
    r = torch.arange(n_side, dtype=torch.float64)
    grid = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(1, n, 3) * a
    pos = grid + torch.randn(n_cells, n, 3, generator=g, dtype=torch.float64) * (sigma * a)
    z = torch.randint(1, max_z + 1, (n_cells, n), generator=g)
    pos = pos.to(dtype).to(device)
    z = z.to(device)
    bx = torch.full((n_cells, 3), n_side * a, dtype=dtype, device=device)
    centre, neigh, shift = _neighbor_list_cells(pos, bx, cutoff_graph, max_neighbors, True)
    return _finish(pos, z, bx, centre, neigh, shift, n)


def random_molecules(n_mol: int, n_atoms: int, extent: float, cutoff_graph: float, species=(1, 6, 7, 8, 9),
                     seed: int = 1234, device="cpu", dtype=torch.float32) -> Batch:
    """Non-periodic clusters (QM9-shaped): `n_atoms` atoms per molecule inside a cube of edge
    `extent`, zero shifts, 50 A dummy lattice as the reference's converter would write."""
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand(n_mol, n_atoms, 3, generator=g, dtype=torch.float64) * extent
    sp = torch.tensor(species)
    z = sp[torch.randint(0, len(species), (n_mol, n_atoms), generator=g)]
    pos = pos.to(dtype).to(device)
    z = z.to(device)
    bx = torch.full((n_mol, 3), 50.0, dtype=dtype, device=device)
    centre, neigh, shift = _neighbor_list_cells(pos, bx, cutoff_graph, None, False)
    return _finish(pos, z, bx, centre, neigh, shift, n_atoms)


def reference_fixture_graph(device="cpu"):
    """The hand-written 3-atom / 8-edge periodic graph the reference tests with
    (restated from tests/model/conftest.py:9-33; two PBC-shifted duplicates of edge 2->1)."""
    from .graph import Data

    shift = torch.zeros(8, 3)
    shift[6] = torch.tensor([0.0, 1.0, -1.0])
    shift[7] = torch.tensor([1.0, 0.0, 0.0])
    return Data(
        edge_index=torch.tensor([[0, 0, 1, 1, 2, 2, 2, 2], [1, 2, 0, 2, 0, 1, 1, 1]]),
        z=torch.tensor([1, 3, 5]),
        pos=torch.tensor([[5.187, 7.50, 7.50], [9.812, 7.50, 7.50], [5.187, 6.56, 7.50]]),
        edge_shift=shift,
        lattice=torch.eye(3).unsqueeze(0) * 15.0,
    ).to(device)


def jittered_cubic_crystal(n_side: int, a: float = 3.2, sigma: float = 0.05, cutoff_graph: float = 5.94,
                           max_neighbors: int | None = 32, max_z: int = 36, seed: int = 1234, device="cuda",
                           dtype=torch.float32) -> Batch:
    """BASELINE config 4 shape: ONE periodic crystal of n_side^3 atoms (jittered simple cubic), neighbour list built
    on the GPU by the cell-list kernels (data/neighbors.py), truncated to the `max_neighbors` nearest per centre as
    the reference's converter does (convert.py:40,145).  n_side = 126 gives the 2 M-atom crystal of config 4; atoms are
    in x-major grid order, so contiguous node ranges are spatial slabs (what shard.make_plans partitions by)."""
    from .neighbors import make_graph

    g = torch.Generator().manual_seed(seed)
    n = n_side**3
    r = torch.arange(n_side, dtype=torch.float64)
    grid = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(n, 3) * a
    pos = (grid + torch.randn(n, 3, generator=g, dtype=torch.float64) * (sigma * a)).to(dtype).to(device)
    z = torch.randint(1, max_z + 1, (n,), generator=g).to(device)
    lattice = (torch.eye(3, dtype=dtype) * (n_side * a)).unsqueeze(0).to(device)
    return make_graph(z, pos, lattice, cutoff_graph, None, None, max_neighbors)


def random_molecule_batch(n_mol: int, min_atoms: int = 3, max_atoms: int = 29, species=(1, 6, 7, 8, 9),
                          cutoff_graph: float = 5.94, density: float = 0.08, seed: int = 1234, device="cuda",
                          max_neighbors: int | None = None, dtype=torch.float32) -> Batch:
    """BASELINE config 5 shape (QM9-like): `n_mol` non-periodic molecules with U{min_atoms..max_atoms} atoms each,
    z drawn from `species`, atoms uniform in a cube sized for `density` atoms/A^3 (near-complete graphs at cutoff 6),
    50 A dummy lattice and zero shifts as the reference's converter writes for molecules."""
    from .neighbors import make_graph

    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(min_atoms, max_atoms + 1, (n_mol,), generator=g)
    batch = torch.repeat_interleave(torch.arange(n_mol), sizes)
    n = int(sizes.sum())
    extent = (sizes.double() / density) ** (1.0 / 3.0)
    pos = torch.rand(n, 3, generator=g, dtype=torch.float64) * extent[batch].unsqueeze(1)
    sp = torch.tensor(species)
    z = sp[torch.randint(0, len(species), (n,), generator=g)]
    lattice = (torch.eye(3, dtype=dtype) * 50.0).unsqueeze(0).repeat(n_mol, 1, 1)
    pbc = torch.zeros(n_mol, 3, dtype=torch.bool)
    return make_graph(z.to(device), pos.to(dtype).to(device), lattice.to(device), cutoff_graph, batch.to(device),
                      pbc.to(device), max_neighbors)
