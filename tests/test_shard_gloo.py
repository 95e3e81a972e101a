"""World-size-2/3 CPU tests of the single-graph sharding logic (lcaonet_amd/shard.py) over the gloo backend.
The partition planner, the packed boundary all-gather and the layer driver are device-agnostic; here the per-rank
layer arithmetic is the CPU oracle run on the rank's local (owned + ghost) graph, and the result must equal the
unsharded oracle forward."""
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
        ret[rank] = (plan.owned_global.clone(), x_own.clone(), total.clone(), plan.n_ghost, int(plan.send_idx.numel()))
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
    """shard.reverse_exchange (the halo all-gather run backwards; a reduce-scatter over RCCL, all-reduce + slice over gloo):
    every owned node receives the sum of the rows the OTHER ranks hold for it as a ghost; interior nodes receive zero."""
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
            assert p.ghost_flat.numel() == p.n_ghost and (p.n_ghost == 0 or int(p.ghost_flat.min()) >= 0)
            assert bool((p.send_idx < p.n_own).all())
            if world == 1:
                assert p.n_ghost == 0 and p.max_send == 0
        # pack / unpack round trip without a process group: rows land where the ghosts expect them
        x = torch.arange(N, dtype=torch.float32).unsqueeze(1).repeat(1, 3)
        bufs = torch.stack([shard.pack_boundary(p, x[p.owned_global]) for p in plans])
        for p in plans:
            got = shard.unpack_ghosts(p, bufs)
            assert torch.equal(got[:, 0].long(), p.ghost_global)
