"""Multi-GPU sharding of ONE graph by centre node — one process per GPU, torch.distributed over RCCL.

The reference has no distributed code; this is the MI355X design for graphs that do not fit (or do not fill) one GPU
(BASELINE config 4: one 2 M-atom crystal).  Batches of independent crystals (config 2) shard by whole graphs and need
none of this: zero halo, no data-path collective (bench.py).

Partition.  Rank r OWNS a set of nodes (a contiguous range / a spatial slab) and every edge whose centre
(`edge_index[0]`, the aggregation index, lcaonet.py:207-213) it owns, so all messages reduce locally — there is no
cross-GPU reduction of node features.  To compute its owned nodes a rank also needs, per layer, the state x of its GHOST
nodes: neighbours t of owned centres (two-body term, lcaonet.py:209) and the sources k of in-edges k->s of owned centres
(three-body term, lcaonet.py:173-189).  Those in-edges are replicated on the rank (geometry, radial basis and
coefficient vectors are functions of the static inputs z, pos), so the ONLY per-layer traffic is the boundary node rows.

Exchange.  xGMI is point to point (7 links per GPU), so the halo exchange is ONE `all_to_all_single` with exact split
sizes per interaction (L-1 per forward): a rank sends each peer precisely the rows that peer holds as ghosts (a slab
talks to its <= 2 neighbours, 2 x 16.8 MB at config 4) and receives its own ghost rows already in local order —
ghosts are numbered by (owner rank, global id), which is the order the owners pack them in, so the receive buffer IS the
ghost block of x (no unpack pass).  Nothing is padded to a common slot size and nothing is replicated to ranks that do
not need it.  The exchange is asynchronous (`start_exchange`): the coefficient kernel of the next layer over the
owned-centre edges (96 % of the layer's matrix work) needs owned rows only and runs under it
(LCAOInteraction.forward_fused(ghost_fn=...)).  One all-reduce of the (B, out) per-graph partial sums closes a forward.

Autograd forces (model/forces.py::sharded_force_steps) add the same exchange run backwards (`reverse_exchange`): dE/dx
of ghost nodes per layer and dE/dpos of ghost atoms return to their owners with the split sizes swapped, and the owner
sums what it received per node in a fixed order (sorted segments: bitwise reproducible).

Plans.  `make_plans(graph, world)` cuts a graph every rank holds (tests, small graphs).  `plan_from_local(...)` builds a
rank's plan from ITS OWN slab only — the global graph never exists anywhere; the send lists are negotiated with two
plan-time all-to-alls (bench.py --workload cfg4, data/synthetic.py::jittered_cubic_crystal_slab).

Local numbering: owned nodes first [0, n_own), ghosts after; edges sorted by centre then have the owned out-edges as a
prefix [0, e_own) of p-space, so the kernels simply run with smaller counts.
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
    ghost_global: Tensor        # (n_ghost) global node ids, ordered by (owner rank, global id)
    send_idx: Tensor            # (n_send) local ids (< n_own) of the rows sent, concatenated by destination rank
    send_splits: list           # (world) rows sent to each rank (this rank's entry is 0)
    recv_splits: list           # (world) ghost rows received from each rank; sum = n_ghost
    n_graphs: int
    rev_perm: Tensor            # (n_send) stable argsort of send_idx: received reverse rows grouped by owned node
    rev_ptr: Tensor             # (n_own + 1) int32 segment pointers of rev_perm

    @property
    def n_send(self) -> int:
        return int(self.send_idx.numel())


def _node_ranges(n_nodes: int, world: int) -> list[tuple[int, int]]:
    base, rem = divmod(n_nodes, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


def _reverse_index(send_idx: Tensor, n_own: int) -> tuple[Tensor, Tensor]:
    perm = torch.argsort(send_idx, stable=True)
    cnt = torch.bincount(send_idx, minlength=n_own) if send_idx.numel() else torch.zeros(n_own, dtype=torch.long,
                                                                                         device=send_idx.device)
    ptr = torch.zeros(n_own + 1, dtype=torch.int32, device=send_idx.device)
    ptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
    return perm, ptr


def make_plans(graph, world: int, ranges: list[tuple[int, int]] | None = None) -> list[ShardPlan]:
    """Plans of every rank from a graph the caller holds whole (host-side index work with torch ops; run once per
    graph).  Contiguous node ranges; callers order nodes spatially / by graph."""
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
        ghosts = touched.nonzero(as_tuple=True)[0]     # ascending global ids = ordered by (owner, id) for ranges
        per_rank.append((e_mask, ghosts))

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
        # what this rank sends to q = q's ghosts that this rank owns, ascending (the order q numbers them in)
        sends = [per_rank[q][1][(per_rank[q][1] >= a) & (per_rank[q][1] < b)] - a if q != r
                 else torch.zeros(0, dtype=torch.long, device=dev) for q in range(world)]
        send_idx = torch.cat(sends) if world > 0 else torch.zeros(0, dtype=torch.long, device=dev)
        recv_splits = torch.bincount(owner[ghosts], minlength=world).tolist() if n_ghost else [0] * world
        perm, ptr = _reverse_index(send_idx, n_own)
        plans.append(ShardPlan(
            rank=r, world=world, n_own=n_own, n_ghost=n_ghost, e_own=int((es < n_own).sum()), graph=g,
            owned_global=torch.arange(a, b, device=dev), ghost_global=ghosts, send_idx=send_idx,
            send_splits=[int(s.numel()) for s in sends], recv_splits=[int(c) for c in recv_splits], n_graphs=n_graphs,
            rev_perm=perm, rev_ptr=ptr))
    return plans


def _a2a(out: Tensor, inp: Tensor, out_splits, in_splits, group=None, async_op: bool = False):
    """all_to_all_single; over gloo (CPU rehearsal of the RCCL path) device tensors are staged through the host."""
    import torch.distributed as dist

    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
        return None
    return dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=group, async_op=async_op)


def plan_from_local(rank: int, world: int, graph: Data, node_gid: Tensor, n_own: int, owner_of, n_graphs: int = 1,
                    group=None, negotiate: bool = True) -> ShardPlan:
    """A rank's plan from its OWN slab: `graph` holds the owned nodes first ([0, n_own)) then the ghosts in any order,
    with every owned out-edge and every in-edge of an owned node; `node_gid` (n_local) are global ids and
    `owner_of(gids) -> ranks` names their owners.  Ghosts are renumbered by (owner, global id) and the send lists come
    from two plan-time all-to-alls (counts, then the requested ids) — collective: every rank of `group` must call.
    negotiate=False (timing ONE rank's share without its peers, bench.py --emulate-world): no collective, empty send
    lists; the caller supplies the ghost rows itself (forward_sharded(exchange_fn=...))."""
    dev = node_gid.device
    n_loc = int(node_gid.numel())
    ghost_gid = node_gid[n_own:]
    ghost_owner = owner_of(ghost_gid).to(torch.long)
    key = ghost_owner * (int(node_gid.max().item()) + 1 if n_loc else 1) + ghost_gid
    o = torch.argsort(key)
    new_of_old = torch.arange(n_loc, device=dev)
    new_of_old[n_own + o] = n_own + torch.arange(o.numel(), device=dev)      # old local id -> new local id
    old_of_new = torch.cat([torch.arange(n_own, device=dev), n_own + o])
    ei = new_of_old[graph[GraphKeys.Edge_idx]]
    g = Data(**{k: v for k, v in graph.to_dict().items()})
    for k in (GraphKeys.Z, GraphKeys.Pos, GraphKeys.Batch_idx):
        if g.get(k) is not None:
            g[k] = g[k][old_of_new].contiguous()
    g[GraphKeys.Edge_idx] = ei.contiguous()
    ghost_gid, ghost_owner = ghost_gid[o], ghost_owner[o]
    recv_cnt = torch.bincount(ghost_owner, minlength=world) if o.numel() else torch.zeros(world, dtype=torch.long, device=dev)
    send_cnt = torch.zeros_like(recv_cnt)
    if world > 1 and negotiate:
        _a2a(send_cnt, recv_cnt, None, None, group)
    recv_splits, send_splits = [int(c) for c in recv_cnt.tolist()], [int(c) for c in send_cnt.tolist()]
    req = torch.empty(sum(send_splits), dtype=torch.long, device=dev)
    if world > 1 and negotiate:
        _a2a(req, ghost_gid.contiguous(), send_splits, recv_splits, group)     # ids the peers want from me
    own_gid = node_gid[:n_own]
    srt = torch.argsort(own_gid)
    pos_in = torch.searchsorted(own_gid[srt], req)
    if req.numel() and not bool((own_gid[srt][pos_in.clamp(max=max(n_own - 1, 0))] == req).all()):
        raise ValueError("plan_from_local: a peer requested a node this rank does not own (owner_of disagrees)")
    send_idx = srt[pos_in] if req.numel() else req
    perm, ptr = _reverse_index(send_idx, n_own)
    return ShardPlan(rank=rank, world=world, n_own=n_own, n_ghost=n_loc - n_own, e_own=int((ei[0] < n_own).sum()), graph=g,
                     owned_global=own_gid, ghost_global=ghost_gid, send_idx=send_idx, send_splits=send_splits,
                     recv_splits=recv_splits, n_graphs=n_graphs, rev_perm=perm, rev_ptr=ptr)


# --- forward halo exchange ---------------------------------------------------------------------------------------
class _Pending:
    """An exchange in flight: `wait()` returns the (n_ghost, H) ghost rows (the current stream waits, not the host)."""

    def __init__(self, recv: Tensor, work, keep):
        self.recv, self.work, self.keep = recv, work, keep

    def wait(self) -> Tensor:
        if self.work is not None:
            self.work.wait()
            self.work = None
        self.keep = None
        return self.recv


def start_exchange(plan: ShardPlan, x_own: Tensor, group=None, out: Tensor | None = None) -> _Pending:
    """Launch ONE all-to-all of boundary rows (backend 'nccl' = RCCL over xGMI on ROCm; 'gloo' on CPU).  `out`: the
    (n_ghost, H) block to receive into (e.g. the ghost rows of the next layer's x)."""
    recv = out if out is not None else x_own.new_empty((plan.n_ghost, x_own.shape[1]))
    if plan.world == 1:     # a rank of a larger group ALWAYS enters (all_to_all_single is a collective of the whole group,
        return _Pending(recv, None, None)   # also for a rank that neither sends nor receives: zero-size splits)
    send = x_own[plan.send_idx]
    work = _a2a(recv, send, plan.recv_splits, plan.send_splits, group, async_op=True)
    return _Pending(recv, work, send)


def exchange_ghosts(plan: ShardPlan, x_own: Tensor, group=None) -> Tensor:
    return start_exchange(plan, x_own, group).wait()


def _sum_by_owned_node(plan: ShardPlan, rows: Tensor) -> Tensor:
    """rows (n_send, W), row i belonging to owned node send_idx[i] -> (n_own, W) sums in a fixed order."""
    if rows.is_cuda:
        from . import ops

        return ops.segment_sum_gather(rows.contiguous(), plan.rev_perm.to(torch.int32), plan.rev_ptr)
    return rows.new_zeros((plan.n_own, rows.shape[1])).index_add_(0, plan.send_idx, rows)   # CPU rehearsal (gloo tests)


def reverse_exchange(plan: ShardPlan, g_ghost: Tensor, group=None) -> Tensor:
    """The forward halo exchange run backwards: every rank returns what it holds for its ghosts, every owner receives
    the rows for the nodes it sent (split sizes swapped) and sums them per node -> (n_own, W) addend (zero for interior
    nodes)."""
    if plan.world == 1:     # as start_exchange: every rank of a larger group enters the collective
        return g_ghost.new_zeros((plan.n_own, g_ghost.shape[1]))
    rows = g_ghost.new_empty((plan.n_send, g_ghost.shape[1]))
    _a2a(rows, g_ghost, plan.send_splits, plan.recv_splits, group)
    return _sum_by_owned_node(plan, rows)


# --- the same data movement without a process group: every rank's plan in one process (lock-step tests) ----------
def local_exchange(plans: list[ShardPlan], x_owns: list[Tensor]) -> list[Tensor]:
    """What start_exchange(...).wait() returns on every rank, computed in one process."""
    packed = [x[p.send_idx] for p, x in zip(plans, x_owns)]
    offs = [torch.tensor([0] + p.send_splits).cumsum(0).tolist() for p in plans]
    out = []
    for p in plans:
        parts = [packed[q][offs[q][p.rank]: offs[q][p.rank + 1]] for q in range(len(plans)) if q != p.rank]
        out.append(torch.cat(parts) if parts else x_owns[p.rank][:0])
    return out


def local_reverse_exchange(plans: list[ShardPlan], g_ghosts: list[Tensor]) -> list[Tensor]:
    """What reverse_exchange returns on every rank, computed in one process."""
    offs = [torch.tensor([0] + p.recv_splits).cumsum(0).tolist() for p in plans]
    out = []
    for p in plans:
        parts = [g_ghosts[q][offs[q][p.rank]: offs[q][p.rank + 1]] for q in range(len(plans)) if q != p.rank]
        rows = torch.cat(parts) if parts else g_ghosts[p.rank][:0]
        # rows arrive grouped by source rank in rank order, which is the layout of send_idx
        out.append(_sum_by_owned_node(p, rows) if rows.shape[0] else rows.new_zeros((p.n_own, rows.shape[1])))
    return out


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

    if partial.is_cuda and dist.get_backend(group) == "gloo":     # CPU rehearsal of the RCCL path
        out = partial.cpu()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out.to(partial.device)
    out = partial.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


# --- autograd-aware collectives (single-graph TRAINING, model/double.py::forward_sharded_differentiable) ----------
class _HaloFn(torch.autograd.Function):
    """ghost rows = exchange(x_own); backward: the exchange run backwards (what every rank holds for its ghosts returns
    to the owners, summed per node).  The two are adjoint, so each one's backward is the other — any order of derivative."""

    @staticmethod
    def forward(ctx, x_own, plan, group):
        ctx.plan, ctx.group = plan, group
        return exchange_ghosts(plan, x_own.contiguous(), group)

    @staticmethod
    def backward(ctx, g):
        return _HaloTFn.apply(g.contiguous(), ctx.plan, ctx.group), None, None


class _HaloTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_ghost, plan, group):
        ctx.plan, ctx.group = plan, group
        return reverse_exchange(plan, g_ghost.contiguous(), group)

    @staticmethod
    def backward(ctx, g):
        return _HaloFn.apply(g.contiguous(), ctx.plan, ctx.group), None, None


def halo_exchange(x_own: Tensor, plan: ShardPlan, group=None) -> Tensor:
    return _HaloFn.apply(x_own, plan, group)


class _AllReduceSumFn(torch.autograd.Function):
    """out = sum over ranks (identical everywhere).  The loss is then computed identically on every rank, so a rank's
    backward takes dL/dout as it is: the gradient of ITS addend (summing the ranks' parameter gradients afterwards gives
    the total)."""

    @staticmethod
    def forward(ctx, part, group):
        return reduce_graph_outputs(part, group)

    @staticmethod
    def backward(ctx, g):
        return g, None


def all_reduce_sum(part: Tensor, group=None) -> Tensor:
    return _AllReduceSumFn.apply(part, group)


def all_reduce_gradients(model, group=None) -> None:
    """Sum every parameter gradient over the ranks (after backward of a sharded differentiable forward)."""
    import torch.distributed as dist

    for p in model.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        if p.grad.is_cuda and dist.get_backend(group) == "gloo":     # CPU rehearsal of the RCCL path
            g = p.grad.cpu()
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
            p.grad.copy_(g)
        else:
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group)
