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
        num_graphs=B,
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
                         max_neighbors: int | None = None, device="cpu", dtype=torch.float32,
                         cell_range: tuple[int, int] | None = None) -> Batch:
    """BASELINE config 2/3 shape: `n_cells` cells, each a jittered n_side^3 simple-cubic lattice
    (spacing `a`, gaussian jitter sigma*a), z ~ U{1..max_z}.  n_cells=1000, n_side=4, a=3.2,
    cutoff 4.95 gives N=64 000 and about 1.15 M edges (about 18 neighbours).
    cell_range = (c0, c1): only cells [c0, c1) of that batch (strong scaling: every rank a slice of the SAME batch)."""
    g = torch.Generator().manual_seed(seed)
    n = n_side**3
    r = torch.arange(n_side, dtype=torch.float64)
    grid = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(1, n, 3) * a
    pos = grid + torch.randn(n_cells, n, 3, generator=g, dtype=torch.float64) * (sigma * a)
    z = torch.randint(1, max_z + 1, (n_cells, n), generator=g)
    if cell_range is not None:
        pos, z = pos[cell_range[0]: cell_range[1]], z[cell_range[0]: cell_range[1]]
        n_cells = pos.shape[0]
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


def _crystal_atoms(n_side: int, a: float, sigma: float, max_z: int, seed: int, dtype=torch.float32):
    """pos (n, 3) and z (n) of jittered_cubic_crystal's atoms (x-major grid order), on the CPU: the same generator
    stream, so a slab cut from here is a piece of exactly that crystal."""
    g = torch.Generator().manual_seed(seed)
    n = n_side**3
    r = torch.arange(n_side, dtype=torch.float64)
    grid = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(n, 3) * a
    pos = (grid + torch.randn(n, 3, generator=g, dtype=torch.float64) * (sigma * a)).to(dtype)
    z = torch.randint(1, max_z + 1, (n,), generator=g)
    return pos, z


def slab_layers(n_side: int, world: int) -> list[tuple[int, int]]:
    """x-layers [x0, x1) owned by each rank: whole layers, sizes differing by at most one."""
    base, rem = divmod(n_side, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


def jittered_cubic_crystal_slab(n_side: int, rank: int, world: int, a: float = 3.2, sigma: float = 0.05,
                                cutoff_graph: float = 5.94, max_neighbors: int | None = 32, max_z: int = 36,
                                seed: int = 1234, device="cuda", dtype=torch.float32):
    """Rank `rank`'s piece of jittered_cubic_crystal(n_side, ...) cut into `world` slabs of whole x-layers, built
    WITHOUT the global graph: only the atom coordinates are global (one seeded stream, 24 B per atom); the neighbour
    search runs on the rank's window = its layers plus a halo of ceil((2 cutoff + 8 sigma a) / a) layers on each side
    (the sources k of in-edges k->s keep their `max_neighbors` nearest, which needs k's whole neighbourhood).

    Returns (graph, node_gid, n_own, owner_of): the local graph with owned atoms first then ghosts, every owned
    out-edge and every in-edge of an owned atom, original (wrapped) coordinates and lattice-unit shifts exactly as in the
    unsharded crystal; global atom ids; `owner_of(gids) -> ranks`.  Feed to shard.plan_from_local."""
    from .graph import Data

    pos_all, z_all = _crystal_atoms(n_side, a, sigma, max_z, seed, dtype)
    layers = slab_layers(n_side, world)
    x0, x1 = layers[rank]
    halo = int(math.ceil((2.0 * cutoff_graph + 8.0 * sigma * a) / a))
    L = n_side * a
    per_layer = n_side * n_side
    bounds = torch.tensor([e for _, e in layers])

    def owner_of(gid):
        return torch.searchsorted(bounds.to(gid.device), torch.div(gid, per_layer, rounding_mode="floor"), right=True)

    whole = (x1 - x0) + 2 * halo >= n_side      # the window would wrap onto itself: search the whole periodic crystal
    if whole:
        lay = torch.arange(n_side)
        wrap = torch.zeros(n_side, dtype=torch.long)
    else:
        lay_u = torch.arange(x0 - halo, x1 + halo)                  # unwrapped layer indices
        wrap = torch.div(lay_u, n_side, rounding_mode="floor")
        lay = lay_u - wrap * n_side
    gid = (lay.unsqueeze(1) * per_layer + torch.arange(per_layer).unsqueeze(0)).reshape(-1)      # window atoms
    wrap_a = wrap.repeat_interleave(per_layer)
    wpos = pos_all[gid].clone()
    zz = z_all[gid]
    if whole:
        lat_search = torch.eye(3, dtype=dtype) * L
    else:
        # unwrapped x, shifted into a cell whose x edge leaves > cutoff of vacuum: periodic images along x never touch
        wpos[:, 0] += (wrap_a.to(dtype) * L - float((x0 - halo) * a) + cutoff_graph + 1.0)
        ext = (x1 - x0 + 2 * halo) * a + 2.0 * (cutoff_graph + 1.0)
        lat_search = torch.diag(torch.tensor([ext, L, L], dtype=dtype))
    if torch.device(device).type == "cuda":
        from .neighbors import neighbor_list

        ei, shift, _ = neighbor_list(wpos.to(device), lat_search.unsqueeze(0).to(device), cutoff_graph, None, None,
                                     max_neighbors)
    else:   # brute force (toy sizes: CPU tests of the planner)
        c, nb, shift = _neighbor_list_cells(wpos.unsqueeze(0), torch.diagonal(lat_search).unsqueeze(0), cutoff_graph,
                                            max_neighbors, True)
        ei = torch.stack([c, nb])
    gid, wrap_a = gid.to(device), wrap_a.to(device)
    own_w = owner_of(gid) == rank          # the window never wraps onto itself, so every owned atom appears once
    keep = own_w[ei[0]] | own_w[ei[1]]
    es, et, sh = ei[0][keep], ei[1][keep], shift[keep].clone()
    if not whole:
        sh[:, 0] = (wrap_a[et] - wrap_a[es]).to(sh.dtype)         # x shift in units of the real lattice
    used = torch.zeros(gid.numel(), dtype=torch.bool, device=device)
    used[es] = True
    used[et] = True
    used &= ~own_w
    own_i = own_w.nonzero(as_tuple=True)[0]
    gh_i = used.nonzero(as_tuple=True)[0]
    nodes = torch.cat([own_i, gh_i])
    local_of = torch.full((gid.numel(),), -1, dtype=torch.long, device=device)
    local_of[nodes] = torch.arange(nodes.numel(), device=device)
    g = Data(z=zz.to(device)[nodes].contiguous(), pos=pos_all.to(device)[gid[nodes]].contiguous(),
             edge_index=torch.stack([local_of[es], local_of[et]]).contiguous(), edge_shift=sh.contiguous(),
             lattice=(torch.eye(3, dtype=dtype) * L).unsqueeze(0).to(device),
             batch=torch.zeros(nodes.numel(), dtype=torch.long, device=device))
    return g, gid[nodes].contiguous(), int(own_i.numel()), owner_of


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
