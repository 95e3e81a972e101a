"""Multi-GPU sharding of ONE graph by centre node — one process per GPU, torch.distributed over RCCL.

The reference has no distributed code; this is the MI355X design for graphs that do not fit (or do not fill) one GPU
(BASELINE config 4: one 2 M-atom crystal).  Batches of independent crystals (config 2) shard by whole graphs and need
none of this: zero halo, no data-path collective (bench.py).

Partition.  Nodes are split into contiguous ranges (callers order nodes spatially / by graph); rank r OWNS the nodes of
its range and every edge whose centre (`edge_index[0]`, the aggregation index, lcaonet.py:207-213) it owns, so all
messages reduce locally — there is no cross-GPU reduction of node features.  To compute its owned nodes a rank also
needs, per layer, the state x of its GHOST nodes: neighbours t of owned centres (two-body term, lcaonet.py:209) and the
sources k of in-edges k->s of owned centres (three-body term, lcaonet.py:173-189).  Those in-edges are replicated on
the rank (geometry, radial basis and coefficient vectors are functions of the static inputs z, pos), so the ONLY
per-layer traffic is one all-gather of boundary node rows (n_boundary x H fp32), L-1 times per forward, plus one
all-reduce of the (B, out) per-graph partial sums.  xGMI is point to point (7 links/GPU): the all-gather of packed
boundary rows uses every link at once; nothing the size of x is ever ring-reduced.

Autograd forces (model/forces.py::sharded_force_steps) add the same exchange run backwards: dE/dx of ghost nodes per
layer and dE/dpos of ghost atoms return to their owners through ONE reduce-scatter of the slot buffer (reverse_exchange).

Local numbering: owned nodes first [0, n_own), ghosts after; edges sorted by centre then have the owned out-edges as a
prefix [0, e_own) of p-space, so the kernels simply run with smaller counts (triplet / message kernels over n_own,
e_own; the coefficient kernel over all local edges because ghost->owned edges feed the three-body term).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor

from .data.graph import Data
from .data.keys import GraphKeys


@dataclass
class ShardPlan:
    rank: int
    world: int
    n_own: int
    n_ghost: int
    e_own: int
    graph: Data                 # local graph (owned + ghost nodes; local node ids; `batch` = global graph ids)
    owned_global: Tensor        # (n_own) global node ids
    ghost_global: Tensor        # (n_ghost) global node ids (ascending)
    send_idx: Tensor            # (n_send) local ids (< n_own) of owned nodes some other rank holds as ghosts
    max_send: int               # max n_send over ranks (all-gather slot size)
    ghost_flat: Tensor          # (n_ghost) index into the gathered (world * max_send) rows
    n_graphs: int


def _node_ranges(n_nodes: int, world: int) -> list[tuple[int, int]]:
    base, rem = divmod(n_nodes, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


def make_plans(graph, world: int, ranges: list[tuple[int, int]] | None = None) -> list[ShardPlan]:
    """Build the plan of every rank from the global graph (host-side index work with torch ops; run once per graph).
    Every rank can call this on the replicated global graph and keep plans[rank]."""
    z = graph[GraphKeys.Z]
    N = z.numel()
    dev = z.device
    idx_s, idx_t = graph[GraphKeys.Edge_idx]
    batch = graph.get(GraphKeys.Batch_idx)
    lattice = graph[GraphKeys.Lattice]
    n_graphs = lattice.shape[0]
    ranges = ranges or _node_ranges(N, world)
    owner = torch.empty(N, dtype=torch.long, device=dev)
    for r, (a, b) in enumerate(ranges):
        owner[a:b] = r
    own_s, own_t = owner[idx_s], owner[idx_t]

    per_rank = []
    for r, (a, b) in enumerate(ranges):
        e_mask = (own_s == r) | (own_t == r)          # owned out-edges + every in-edge of an owned node
        es, et = idx_s[e_mask], idx_t[e_mask]
        touched = torch.zeros(N, dtype=torch.bool, device=dev)
        touched[es] = True
        touched[et] = True
        touched[a:b] = False
        ghosts = touched.nonzero(as_tuple=True)[0]     # ascending global ids
        per_rank.append((e_mask, ghosts))

    # boundary (send) sets: owned nodes that are ghosts anywhere else, in ascending global id
    is_boundary = torch.zeros(N, dtype=torch.bool, device=dev)
    for _, ghosts in per_rank:
        is_boundary[ghosts] = True
    send_global = [(is_boundary[a:b].nonzero(as_tuple=True)[0] + a) for (a, b) in ranges]
    max_send = max(int(s.numel()) for s in send_global) if world > 0 else 0
    slot_of = torch.full((N,), -1, dtype=torch.long, device=dev)   # global id -> flat slot in the gathered buffer
    for r, s in enumerate(send_global):
        slot_of[s] = r * max_send + torch.arange(s.numel(), device=dev)

    plans = []
    for r, (a, b) in enumerate(ranges):
        e_mask, ghosts = per_rank[r]
        n_own, n_ghost = b - a, int(ghosts.numel())
        local_of = torch.full((N,), -1, dtype=torch.long, device=dev)
        local_of[a:b] = torch.arange(n_own, device=dev)
        local_of[ghosts] = n_own + torch.arange(n_ghost, device=dev)
        nodes = torch.cat([torch.arange(a, b, device=dev), ghosts])
        es, et = local_of[idx_s[e_mask]], local_of[idx_t[e_mask]]
        g = Data(
            z=z[nodes].contiguous(), pos=graph[GraphKeys.Pos][nodes].contiguous(),
            edge_index=torch.stack([es, et]).contiguous(),
            edge_shift=graph[GraphKeys.Edge_shift][e_mask].contiguous(), lattice=lattice,
            batch=(batch[nodes].contiguous() if batch is not None else torch.zeros(nodes.numel(), dtype=torch.long, device=dev)),
        )
        plans.append(ShardPlan(
            rank=r, world=world, n_own=n_own, n_ghost=n_ghost, e_own=int((es < n_own).sum()), graph=g,
            owned_global=torch.arange(a, b, device=dev), ghost_global=ghosts, send_idx=(send_global[r] - a), max_send=max_send,
            ghost_flat=slot_of[ghosts], n_graphs=n_graphs))
    return plans


def pack_boundary(plan: ShardPlan, x_own: Tensor) -> Tensor:
    """(max_send, H) send buffer: this rank's boundary rows, zero padded to the common slot size."""
    buf = x_own.new_zeros((plan.max_send, x_own.shape[1]))
    if plan.send_idx.numel():
        buf[: plan.send_idx.numel()] = x_own[plan.send_idx]
    return buf


def unpack_ghosts(plan: ShardPlan, gathered: Tensor) -> Tensor:
    """gathered (world, max_send, H) -> (n_ghost, H) rows of this rank's ghosts."""
    return gathered.reshape(-1, gathered.shape[-1])[plan.ghost_flat]


def exchange_ghosts(plan: ShardPlan, x_own: Tensor, group=None) -> Tensor:
    """ONE collective: all-gather of packed boundary rows (backend 'nccl' = RCCL over xGMI on ROCm, 'gloo' on CPU)."""
    import torch.distributed as dist

    send = pack_boundary(plan, x_own)
    recv = send.new_empty((plan.world,) + tuple(send.shape))
    if plan.max_send > 0:
        dist.all_gather_into_tensor(recv.view(-1, send.shape[1]), send, group=group)
    return unpack_ghosts(plan, recv)


def scatter_ghost_rows(plan: ShardPlan, g_ghost: Tensor) -> Tensor:
    """(n_ghost, W) rows held for ghost nodes -> the (world * max_send, W) slot buffer of the forward all-gather with
    those rows at their owners' slots (zero elsewhere): the operand of the reverse collective."""
    buf = g_ghost.new_zeros((plan.world * plan.max_send, g_ghost.shape[1]))
    if plan.n_ghost:
        buf[plan.ghost_flat] = g_ghost          # ghost_flat is injective: one slot per ghost
    return buf


def own_addend(plan: ShardPlan, slots: Tensor) -> Tensor:
    """(max_send, W) reduced slot rows of THIS rank -> (n_own, W) addend for its owned nodes (zero for interior nodes)."""
    out = slots.new_zeros((plan.n_own, slots.shape[1]))
    n_send = plan.send_idx.numel()
    if n_send:
        out[plan.send_idx] = slots[:n_send]     # send_idx is injective
    return out


def reverse_exchange(plan: ShardPlan, g_ghost: Tensor, group=None) -> Tensor:
    """The forward halo all-gather run backwards: every rank contributes what it holds for its ghosts, every owner
    receives the sum for its boundary nodes — ONE reduce-scatter of the slot buffer (RCCL); gloo has no
    reduce-scatter, there an all-reduce of the slot buffer is sliced instead."""
    import torch.distributed as dist

    buf = scatter_ghost_rows(plan, g_ghost)
    if plan.max_send == 0:
        return g_ghost.new_zeros((plan.n_own, g_ghost.shape[1]))
    mine = buf.new_empty((plan.max_send, buf.shape[1]))
    if dist.get_backend(group) == "gloo":
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        mine.copy_(buf[plan.rank * plan.max_send: (plan.rank + 1) * plan.max_send])
    else:
        dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM, group=group)
    return own_addend(plan, mine)


def run_sharded_layers(plan: ShardPlan, x_local: Tensor, n_layers: int, layer_fn, exchange_fn) -> Tensor:
    """Drive L interaction layers on one rank: `layer_fn(i, x_local) -> x_new` (rows [:n_own] valid),
    `exchange_fn(x_own) -> x_ghost`.  L-1 exchanges: the last layer's ghosts are never read."""
    x = x_local
    for i in range(n_layers):
        x_new = layer_fn(i, x)
        if i + 1 < n_layers:
            ghosts = exchange_fn(x_new[: plan.n_own])
            x = torch.cat([x_new[: plan.n_own], ghosts], dim=0) if plan.n_ghost else x_new[: plan.n_own]
        else:
            x = x_new
    return x[: plan.n_own]


def reduce_graph_outputs(partial: Tensor, group=None) -> Tensor:
    """Per-graph partial sums (B, out) of the owned nodes -> global result: one small all-reduce."""
    import torch.distributed as dist

    out = partial.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out
