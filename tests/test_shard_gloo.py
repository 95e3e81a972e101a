"""World-size-2/3 CPU tests of the single-graph sharding logic (lcaonet_amd/shard.py) over the gloo backend.
The partition planner, the boundary all-to-all and the layer driver are device-agnostic; here the per-rank
layer arithmetic is the CPU oracle run on the rank's local (owned + ghost) graph, and the result must equal the
unsharded oracle forward.  Plans come both from a graph every rank holds (shard.make_plans) and from per-rank slabs
that never see the global graph (data/synthetic.py::jittered_cubic_crystal_slab + shard.plan_from_local: the route of
`bench.py --workload cfg4 --gpus N`)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lcaonet_amd import shard
from lcaonet_amd.data.synthetic import random_periodic_cells
from oracle import restatement as O

KW = dict(emb_size=16, emb_size_coeff=16, emb_size_conv=16, n_interaction=3, cutoff=4.0, max_z=20)


def _state_dict(seed=0):
    from lcaonet_amd.model import LCAONet
    torch.manual_seed(seed)
    return {k: v.clone() for k, v in LCAONet(**KW).eval().state_dict().items()}


def _graph():
    # ONE periodic cell (a single graph spanning all ranks), nodes in generator order
    return random_periodic_cells(1, 40, 9.0, 0.99 * 4.0, max_z=20, seed=3)


def _oracle_layers_on_local_graph(sd, cfg, plan):
    """Static per-rank quantities of the oracle on the local graph + a layer function with the driver's contract."""
    g = plan.graph.to_dict()
    z, ei = g["z"], g["edge_index"]
    tri_k, e_ks, e_st = O.get_triplets(ei, z.numel())
    d, vec = O.atomic_distances(g["pos"], ei, g["edge_shift"], g["lattice"], g["batch"])
    nl = O.nl_list(cfg.max_z, cfg.max_orb, cfg.n_per_orb)
    rb = O.hydrogen_radial_basis(d, nl, cfg.cutoff, cfg.cutoff_net)
    shb = O.spherical_harmonics_basis((vec[e_st] * vec[e_ks]).sum(-1), nl)
    x0, cst = O.embedding(sd, cfg, z, ei[0], ei[1])

    def layer_fn(i, x_local):
        return O.interaction(sd, f"int_layers.{i}.", cfg, x_local, cst, None, rb, shb, ei[0], ei[1], tri_k, e_ks, e_st)

    return x0, layer_fn


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        sd, cfg = _state_dict(), O.OracleConfig(**KW)
        plan = shard.make_plans(_graph(), world)[rank]
        with torch.no_grad():
            x0, layer_fn = _oracle_layers_on_local_graph(sd, cfg, plan)
            x_own = shard.run_sharded_layers(plan, x0, cfg.n_interaction, layer_fn,
                                             lambda xo: shard.exchange_ghosts(plan, xo))
            prop = O.out_layer(sd, cfg, x_own, None, None, None, None)        # owned nodes of the single graph, summed
            total = shard.reduce_graph_outputs(prop)
        ret[rank] = (plan.owned_global.clone(), x_own.clone(), total.clone(), plan.n_ghost, plan.n_send)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_layers_match_unsharded_oracle(world):
    sd, cfg = _state_dict(), O.OracleConfig(**KW)
    g = _graph().to_dict()
    tr = {}
    with torch.no_grad():
        want = O.forward(sd, cfg, g, trace=tr)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    seen = torch.zeros(g["z"].numel(), dtype=torch.bool)
    for r in range(world):
        owned, x_own, total, n_ghost, n_send = ret[r]
        assert n_ghost > 0 and n_send > 0          # a periodic cell: every rank has a halo
        seen[owned] = True
        ref = tr[f"x{cfg.n_interaction}"][owned]
        assert float((x_own - ref).abs().max() / ref.abs().max()) <= 2e-6
        assert float((total - want).abs().max() / want.abs().max()) <= 2e-6   # identical on every rank
    assert bool(seen.all())


def _reverse_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = shard.make_plans(_graph(), world)[rank]
        gid = plan.ghost_global.double()
        g_ghost = torch.stack([(gid + 1) * (rank + 1), gid * 0 + 10.0 ** rank], dim=1).float()
        ret[rank] = (plan.owned_global.clone(), shard.reverse_exchange(plan, g_ghost).clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reverse_exchange_sums_ghost_rows_at_their_owners(world):
    """shard.reverse_exchange (the halo all-to-all run backwards, split sizes swapped): every owned node receives the sum
    of the rows the OTHER ranks hold for it as a ghost; interior nodes receive zero."""
    plans = shard.make_plans(_graph(), world)
    N = _graph()["z"].numel()
    want = torch.zeros(N, 2, dtype=torch.float64)
    for p in plans:
        want[p.ghost_global, 0] += (p.ghost_global.double() + 1) * (p.rank + 1)
        want[p.ghost_global, 1] += 10.0 ** p.rank
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reverse_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        owned, got = ret[r]
        assert got.shape == (owned.numel(), 2)
        assert torch.allclose(got.double(), want[owned], rtol=1e-6, atol=0)
    assert float(want.abs().sum()) > 0


def test_plan_invariants():
    g = _graph()
    N = g["z"].numel()
    idx_s, idx_t = g["edge_index"]
    for world in (1, 2, 4):
        plans = shard.make_plans(g, world)
        assert sum(p.n_own for p in plans) == N
        assert sum(p.e_own for p in plans) == idx_s.numel()        # every edge is owned exactly once
        for p in plans:
            es, et = p.graph["edge_index"]
            own = es < p.n_own
            assert int(own.sum()) == p.e_own
            assert bool((et[~own] < p.n_own).all())                # replicated edges are in-edges of owned nodes
            assert sum(p.recv_splits) == p.n_ghost and sum(p.send_splits) == p.n_send
            assert p.send_splits[p.rank] == 0 and p.recv_splits[p.rank] == 0
            assert bool((p.send_idx < p.n_own).all())
            if world == 1:
                assert p.n_ghost == 0 and p.n_send == 0
        for p in plans:    # what r sends q is what q expects from r
            assert p.send_splits == [q.recv_splits[p.rank] for q in plans]
        # the exchange's data movement without a process group: rows land where the ghosts expect them
        x = torch.arange(N, dtype=torch.float32).unsqueeze(1).repeat(1, 3)
        got = shard.local_exchange(plans, [x[p.owned_global] for p in plans])
        for p, rows in zip(plans, got):
            assert torch.equal(rows[:, 0].long(), p.ghost_global)
        # and backwards: every owner gets the sum over the ranks that hold its node as a ghost
        back = shard.local_reverse_exchange(plans, [torch.ones(p.n_ghost, 1) for p in plans])
        held = torch.zeros(N)
        for p in plans:
            held[p.ghost_global] += 1
        for p, b in zip(plans, back):
            assert torch.equal(b[:, 0], held[p.owned_global])


# ---------------------------------------------------------------------------------------------------------------------
# per-rank slabs: the global graph is never built (bench.py --workload cfg4)
# ---------------------------------------------------------------------------------------------------------------------
SLAB = dict(n_side=12, a=3.2, sigma=0.05, cutoff_graph=3.6, max_neighbors=5, max_z=20, seed=7)
SLAB_KW = dict(KW, cutoff=3.64)


def _edge_keys(g, gid):
    ei, sh = g["edge_index"], g["edge_shift"].long()
    return set(zip(gid[ei[0]].tolist(), gid[ei[1]].tolist(), map(tuple, sh.tolist())))


def test_slabs_tile_the_crystal():
    """Every edge of the whole crystal is an owned out-edge of exactly one slab (same shifts, same truncation to the
    nearest neighbours); the replicated edges are in-edges of owned atoms; coordinates are the crystal's own."""
    from lcaonet_amd.data.synthetic import jittered_cubic_crystal_slab
    whole, gid_w, n_w, _ = jittered_cubic_crystal_slab(rank=0, world=1, device="cpu", **SLAB)
    assert n_w == SLAB["n_side"] ** 3 and 4.9 * n_w < whole["edge_index"].shape[1] <= 5 * n_w   # truncation is active
    want = _edge_keys(whole, gid_w)
    pos_of = torch.empty(n_w, 3)
    pos_of[gid_w] = whole["pos"]
    seen = set()
    for world in (3,):
        for r in range(world):
            g, gid, n_own, owner_of = jittered_cubic_crystal_slab(rank=r, world=world, device="cpu", **SLAB)
            assert bool((owner_of(gid[:n_own]) == r).all()) and bool((owner_of(gid[n_own:]) != r).all())
            assert torch.equal(g["pos"], pos_of[gid])
            ei = g["edge_index"]
            own = ei[0] < n_own
            assert bool((ei[1][~own] < n_own).all())
            keys = _edge_keys(g, gid)
            assert keys <= want
            mine = {k for k in keys if int(owner_of(torch.tensor([k[0]]))) == r}
            assert len(mine) == int(own.sum()) and not (mine & seen)
            seen |= mine
            # every in-edge of an owned atom is present
            assert {k for k in want if int(owner_of(torch.tensor([k[1]]))) == r} <= keys
    assert seen == want


def _slab_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lcaonet_amd.data.synthetic import jittered_cubic_crystal_slab
        torch.set_num_threads(1)
        cfg = O.OracleConfig(**SLAB_KW)
        torch.manual_seed(0)
        from lcaonet_amd.model import LCAONet
        sd = {k: v.clone() for k, v in LCAONet(**SLAB_KW).eval().state_dict().items()}
        g, gid, n_own, owner_of = jittered_cubic_crystal_slab(rank=rank, world=world, device="cpu", **SLAB)
        plan = shard.plan_from_local(rank, world, g, gid, n_own, owner_of)
        with torch.no_grad():
            x0, layer_fn = _oracle_layers_on_local_graph(sd, cfg, plan)
            x_own = shard.run_sharded_layers(plan, x0, cfg.n_interaction, layer_fn,
                                             lambda xo: shard.exchange_ghosts(plan, xo))
            total = shard.reduce_graph_outputs(O.out_layer(sd, cfg, x_own, None, None, None, None))
            back = shard.reverse_exchange(plan, torch.ones(plan.n_ghost, 1))
        ret[rank] = (plan.owned_global.clone(), x_own.clone(), total.clone(), plan.ghost_global.clone(), back.clone(),
                     list(plan.send_splits))
    finally:
        dist.destroy_process_group()


def test_per_rank_slab_plans_match_unsharded_oracle():
    """Three ranks, each building ONLY its slab and negotiating the send lists over gloo (shard.plan_from_local), run
    the oracle layers with the real all-to-all exchange: owned states and the reduced energy equal the unsharded oracle
    on the whole crystal."""
    from lcaonet_amd.data.synthetic import jittered_cubic_crystal_slab
    from lcaonet_amd.model import LCAONet
    world = 3
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in LCAONet(**SLAB_KW).eval().state_dict().items()}
    cfg = O.OracleConfig(**SLAB_KW)
    whole, gid_w, _, _ = jittered_cubic_crystal_slab(rank=0, world=1, device="cpu", **SLAB)
    tr = {}
    with torch.no_grad():
        want = O.forward(sd, cfg, whole.to_dict(), trace=tr)
    x_ref = torch.empty_like(tr[f"x{cfg.n_interaction}"])
    x_ref[gid_w] = tr[f"x{cfg.n_interaction}"]            # by global atom id
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_slab_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    held = torch.zeros(x_ref.shape[0])
    for r in range(world):
        held[ret[r][3]] += 1
    for r in range(world):
        owned, x_own, total, _, back, splits = ret[r]
        assert splits[r] == 0 and sum(1 for s in splits if s) == 2          # a slab talks to its two neighbours only
        assert float((x_own - x_ref[owned]).abs().max() / x_ref.abs().max()) <= 2e-6
        assert float((total - want).abs().max() / want.abs().max()) <= 2e-6
        assert torch.equal(back[:, 0], held[owned])
