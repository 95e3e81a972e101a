"""Autograd forces at inference: F = -dE/dpos for `regress_forces=True, direct_forces=False`
(reference lcaonet/model/lcaonet.py:497-498, 310-317).

The reference records an autograd graph through every op (create_graph=True).  Here the forward pass of the
species-pair-table form (model/lcaonet.py, cst_mode="pair") keeps the few per-layer tensors the chain rule needs and a
hand-written reverse sweep runs on the HIP kernels of csrc/lcao_backward.hip: dense layers are differentiated by
`lcao_dense` on transposed weights, everything else by small vector-Jacobian kernels.  Only activation gradients are
propagated (inference); parameter gradients — training — are the remaining part of SURVEY.md §8f-3.

Restrictions (each raises NotImplementedError): eval mode, SiLU activations, built-in cutoff / basis.
add_valence=True runs the unfused two-body kernels and three pair-table passes per layer.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..data.keys import GraphKeys
from ..nn.scatter import scatter

ACT = ops.ACT_SILU


def _t(w: Tensor) -> Tensor:
    return w.detach().t().contiguous()


def _in_sum(rows: Tensor, ctx, n_nodes: int) -> Tensor:
    """sum of per-edge rows over the IN-edges of every node (deterministic: in_pos order)."""
    return ops.segment_sum_gather(rows, ctx.in_pos, ctx.in_ptr[: n_nodes + 1])


@torch.no_grad()
def layer_forward_saving(layer, x, Tf_in, rb, uvec_p, ctx, l_of_d, n_l, z, zdim, mask_pair=None, n_own=None, e_own=None):
    """One interaction layer in pair-table form; returns (x_new, saved tensors for the reverse sweep).
    n_own / e_own (sharded graphs): x holds owned nodes first then ghosts; only the first n_own nodes are centres and
    only the first e_own edges of p-space carry messages (as LCAOInteraction.forward_fused); x_new has n_own rows."""
    V = layer.emb_size_conv
    N, E = x.shape[0], ctx.n_edges
    n_own = N if n_own is None else n_own
    e_own = E if e_own is None else e_own
    out_ptr, in_ptr = ctx.out_ptr[: n_own + 1], ctx.in_ptr[: n_own + 1]
    src_own, dst_own = ctx.src_p[:e_own], ctx.dst_p[:e_own]
    wn, bn_ = layer.node_weight.weight.detach(), layer.node_weight.bias.detach()
    xs = ops.dense(x, wn[:V], bn_[:V])
    xk = ops.dense(x, wn[V:], bn_[V:])
    gate = torch.sigmoid(xk)
    w1 = layer.f_node[0].weight.detach()
    Ps = ops.dense(xs[:n_own], w1[:, :V].contiguous(), layer.f_node[0].bias.detach())
    Pt = ops.dense(xs, w1[:, V:].contiguous())
    Tf = ops.dense(ops.dense(Tf_in, layer.f_coeffs[0].weight.detach(), None, ACT), layer.f_coeffs[2].weight.detach(),
                   None, ACT)
    pc = lambda T: ops.pair_contract(T, rb, ctx.order, l_of_d, z, ctx.src_p, ctx.dst_p, gate, n_l, zdim)
    Bv = None
    if layer.add_valence:
        from .lcaonet import split_valence_table
        Tc, Tv = split_valence_table(Tf, mask_pair, V)
        Tf = (Tc + Tv, Tc, Tv)
        Ag, gram, _ = pc(Tf[0])
        B, Bv = pc(Tc)[2][:e_own], pc(Tv)[2][:e_own]
    else:
        Ag, gram, B = pc(Tf)
        B = B[:e_own]
    tbw = ops.triplet_gemm(Ag, gram, uvec_p, in_ptr, ctx.in_pos, out_ptr, e_own)
    if V in ops.FUSED_WIDTHS and not layer.add_valence:
        # the forward's fused two-body kernels, keeping ft and pre2; pre1 = Ps[s] + Pt[t] is recomputed in the sweep
        bw, ft = ops.edge_lcao_weight(tbw, B, layer.f_three[0].weight.detach(), layer.basis_weight.weight.detach(),
                                      save=True)
        agg, pre2 = ops.edge_node_message_agg(bw, Ps, Pt, src_own, dst_own, out_ptr, layer.f_node[2].weight.detach(),
                                              layer.f_node[2].bias.detach(), ctx.column_hint(out_ptr, e_own), save=True)
        pre1 = None
    else:
        ft = ops.dense(tbw, layer.f_three[0].weight.detach())
        lw = ops.edge_combine(ft, B, Bv)
        bw = ops.dense(lw, layer.basis_weight.weight.detach())
        pre1 = ops.pair_pre(Ps, Pt, src_own, dst_own)
        h1 = torch.nn.functional.silu(pre1)
        pre2 = ops.dense(h1, layer.f_node[2].weight.detach(), layer.f_node[2].bias.detach())
        agg = ops.segment_sum(bw, out_ptr, torch.nn.functional.silu(pre2))
    x_new = ops.dense(agg, layer.out_weight.weight.detach(), None, ops.ACT_NONE, x[:n_own])
    saved = dict(xk=xk, gate=gate, Tf=Tf, Ag=Ag, gram=gram, B=B, Bv=Bv, ft=ft, bw=bw, pre1=pre1, pre2=pre2, Ps=Ps, Pt=Pt,
                 n_own=n_own, e_own=e_own, n_local=N)
    return x_new, saved


@torch.no_grad()
def layer_backward(layer, sv, g_xnew, rb, uvec_p, ctx, l_of_d, n_l, z, zdim, tbase, n_trip_dense, g_rb, g_u_p,
                   first: bool = False):
    """Reverse sweep of one layer: g_xnew = dE/dx_out (n_own rows) -> dE/dx_in (all local rows: owned nodes, then the
    ghosts of a sharded graph); accumulates dE/drb (orig order) and dE/du (p-space) of every local edge.
    first: the first interaction layer — its input x depends on z only, so nothing on the node side (dE/dPs, dE/dPt,
    dE/dgate, dE/dx) is formed and None is returned."""
    V = layer.emb_size_conv
    E = ctx.n_edges
    n_own, e_own, N = sv["n_own"], sv["e_own"], sv["n_local"]
    out_own = ctx.out_ptr[: n_own + 1]
    src_own, dst_own = ctx.src_p[:e_own], ctx.dst_p[:e_own]
    # x' = x + agg Wout^T
    wt = layer.transposed_weights()
    g_agg = ops.dense(g_xnew, wt["out"])
    # msg = bw * silu(pre2), agg = segment sum
    # bw = lw Wb^T ; lw = normalize(B (1 + ft)) ; ft = tbw W3^T
    g_Bv = None
    if V in ops.FUSED_WIDTHS and not layer.add_valence:
        g_bw, g_pre1, _ = ops.message_bwd_fused(g_agg, src_own, dst_own, sv["bw"], sv["pre2"], sv["Ps"], sv["Pt"],
                                                wt["node2"], want_g_pre1=not first)
        g_B, g_tbw, _ = ops.lcao_weight_bwd_fused(g_bw, sv["B"], sv["ft"], wt["basis"], wt["three"])
    else:
        g_bw, g_pre2 = ops.message_bwd(g_agg, src_own, sv["bw"], sv["pre2"])
        g_h1 = ops.dense(g_pre2, wt["node2"])
        g_pre1 = ops.act_bwd(g_h1, sv["pre1"], ACT)
        g_lw = ops.dense(g_bw, wt["basis"])
        if layer.add_valence:
            g_B, g_Bv, g_ft = ops.combine_bwd(g_lw, sv["B"], sv["ft"], sv["Bv"])
        else:
            g_B, g_ft = ops.combine_bwd(g_lw, sv["B"], sv["ft"])
        g_tbw = ops.dense(g_ft, wt["three"])
    if not first:
        g_Ps = ops.segment_sum(g_pre1, out_own)              # pre1 = Ps[src] + Pt[dst]
        if e_own < E:   # ghost-centred edges carry no two-body message: their rows of dE/dpre1 are zero
            g_pre1 = torch.cat([g_pre1, g_pre1.new_zeros(E - e_own, V)], dim=0)
        g_Pt = _in_sum(g_pre1, ctx, N)
    # three-body term
    g_Ag, g_gram, g_uq, g_up = ops.triplet_bwd(g_tbw, sv["Ag"], sv["gram"], uvec_p, ctx.src_p, ctx.dst_p, ctx.in_ptr,
                                               ctx.in_pos, ctx.out_ptr, tbase, n_trip_dense, e_own, n_own)
    g_u_p += g_uq
    g_u_p += g_up
    # pair-table contraction: A from rb and Tf; Ag = A gate; B; gram
    pcb = lambda T, gA, gB, gG: ops.pair_contract_bwd(T, rb, ctx.order, l_of_d, z, ctx.src_p, ctx.dst_p, sv["gate"], gA, gB,
                                                      gG, e_own, n_l, zdim, g_rb, want_gate=not first)
    if layer.add_valence:
        # A (three-body) comes from Tc + Tv, B from Tc, Bv from Tv: three passes, each adding its part of dE/drb
        Te, Tc, Tv = sv["Tf"]
        zA, zB, zG = torch.zeros_like(g_Ag), torch.zeros_like(g_B), torch.zeros_like(g_gram)
        g_gate_part = pcb(Te, g_Ag, zB, g_gram)
        pcb(Tc, zA, g_B, zG)
        pcb(Tv, zA, g_Bv, zG)
    else:
        g_gate_part = pcb(sv["Tf"], g_Ag, g_B, g_gram)
    if first:
        return None
    g_gate = ops.segment_sum(g_gate_part, ctx.out_ptr[: N + 1])     # gate[src] of every local edge
    g_xk = ops.act_bwd(g_gate, sv["xk"], ops.ACT_SIGMOID)
    # node level: Ps = xs Wa^T + b (owned), Pt = xs Wb^T (all local) ; [xs | xk] = x Wnw^T + b
    g_xs = ops.dense(g_Pt, wt["node1_t"])
    g_xs_own = ops.dense(g_Ps, wt["node1_s"], None, ops.ACT_NONE, g_xs[:n_own])
    if n_own < N:
        g_xs = torch.cat([g_xs_own, g_xs[n_own:]], dim=0)
        res = torch.cat([g_xnew, g_xnew.new_zeros(N - n_own, g_xnew.shape[1])], dim=0)   # residual path: owned rows
    else:
        g_xs, res = g_xs_own, g_xnew
    g_x = ops.dense(g_xs, wt["nw_s"], None, ops.ACT_NONE, res)
    g_x = ops.dense(g_xk, wt["nw_k"], None, ops.ACT_NONE, g_x)
    return g_x


def _check_supported(model, have_tables: bool = False):
    if model.training and not have_tables:
        raise NotImplementedError("autograd forces: eval mode only (BatchNorm batch statistics are not differentiated)")
    acts = [model.int_layers[0].f_coeffs[1], model.int_layers[0].f_node[1], model.out_layer.out_lin[1]]
    if not all(type(a) is nn.SiLU for a in acts):
        raise NotImplementedError("autograd forces are built for the default SiLU activation")
    if getattr(model.rbf, "rbf_kind", None) is None or getattr(model.rbf.cutoff_net, "kind", None) is None:
        raise NotImplementedError("autograd forces need a built-in radial basis and cutoff")


@torch.no_grad()
def forward_with_autograd_forces(model, graph, tables=None, ctx=None):
    """Returns (out (B, out_size), force) with force = -d(sum_b out[b, i])/dpos: (N, 3) for out_size == 1,
    (N, out_size, 3) otherwise — the shapes of lcaonet.py:310-317.
    tables = (x_tab (max_z+1, H), cst_pair, zdim): embedding tables supplied by the caller (model/train.py: train-mode
    BatchNorm batch statistics) instead of the eval-mode ones."""
    from .lcaonet import EdgeContext
    from .base import BaseMPNN, batch_plan, num_graphs

    try:
        _check_supported(model, tables is not None)
    except NotImplementedError:
        if tables is not None or ctx is not None:
            raise
        # other activations / user cutoffs: the generic twice-differentiable graph (model/double.py), results detached
        from .double import forward_exact_forces

        with torch.enable_grad():
            out, force = forward_exact_forces(model, graph, update_stats=False)
        return out.detach(), force.detach()
    batch_idx = graph.get(GraphKeys.Batch_idx)
    z, pos = model.checked_z(graph), graph[GraphKeys.Pos]
    idx_s, idx_t = graph[GraphKeys.Edge_idx]
    ops._dev(pos, "graph.pos")
    N = z.size(0)
    ctx = ctx if ctx is not None else EdgeContext.build(graph[GraphKeys.Edge_idx], N)
    graph = BaseMPNN.calc_atomic_distances(graph, return_vec=True)
    dist, uvec = graph[GraphKeys.Edge_dist], graph[GraphKeys.Edge_vec_st]
    uvec_p = ctx.uvec_p(uvec)
    rb = model.rbf(dist)
    if tables is None:
        x, cst_pair, zdim = model.emb_layer.forward_pair_table(z)
    else:
        x, cst_pair, zdim = ops.gather_rows(tables[0], z), tables[1], tables[2]
    l_of_d, n_l = model.shbf.l_of_d, model.n_l

    mask_pair = model.pair_valence_mask(zdim) if model.add_valence else None
    saved, x_in = [], x
    for layer in model.int_layers:
        x_in, sv = layer_forward_saving(layer, x_in, cst_pair, rb, uvec_p, ctx, l_of_d, n_l, z, zdim, mask_pair)
        saved.append(sv)
    xL = x_in

    # output head (lcaonet.py:290-298) with the pre-activations kept
    lin = model.out_layer.out_lin
    pre_a = ops.dense(xL, lin[0].weight.detach(), lin[0].bias.detach())
    a = torch.nn.functional.silu(pre_a)
    pre_b = ops.dense(a, lin[2].weight.detach(), lin[2].bias.detach())
    b = torch.nn.functional.silu(pre_b)
    prop = ops.dense(b, lin[4].weight.detach())
    n_graphs = num_graphs(graph, batch_idx)
    red = "sum" if model.out_layer.is_extensive else "mean"
    plan = batch_plan(graph, batch_idx, n_graphs)
    if batch_idx is not None:
        out = scatter(prop, batch_idx, dim=0, dim_size=n_graphs, reduce=red, plan=plan)
        cnt = (plan[1][1:] - plan[1][:-1]).clamp(min=1).to(prop.dtype)           # atoms per graph
        node_scale = torch.ones(N, device=prop.device) if red == "sum" else (1.0 / cnt)[batch_idx]
    else:
        out = prop.sum(0, keepdim=True) if red == "sum" else prop.mean(0, keepdim=True)
        node_scale = torch.full((N,), 1.0 if red == "sum" else 1.0 / N, device=prop.device)
    out = model.pp_layer(out, z, batch_idx, n_graphs, batch_plan=plan)     # atomref / mean: no position dependence

    # dense per-node scratch layout for the per-triplet dE/dcos
    m_in = (ctx.in_ptr[1:] - ctx.in_ptr[:-1]).long()
    m_out = (ctx.out_ptr[1:] - ctx.out_ptr[:-1]).long()
    prod = m_in * m_out
    tbase = (torch.cumsum(prod, 0) - prod).contiguous()
    # only the any-V fallback of the three-body backward needs the scratch size (a host sync); the MFMA kernel does not
    n_trip_dense = int(prod.sum().item()) if model.emb_size_conv not in ops.FUSED_WIDTHS else 1

    forces = []
    for i in range(model.out_size):
        # E_i = sum_b out[b, i]: dE_i/dprop[n, j] = node_scale[n] * delta_ij
        g_prop = torch.zeros_like(prop)
        g_prop[:, i] = node_scale
        g_b = ops.dense(g_prop, _t(lin[4].weight))
        g_a = ops.dense(ops.act_bwd(g_b, pre_b, ACT), _t(lin[2].weight))
        g_x = ops.dense(ops.act_bwd(g_a, pre_a, ACT), _t(lin[0].weight))
        g_rb = torch.zeros_like(rb)
        g_u_p = torch.zeros(ctx.n_edges, 3, device=pos.device)
        for li in reversed(range(len(saved))):
            g_x = layer_backward(model.int_layers[li], saved[li], g_x, rb, uvec_p, ctx, l_of_d, n_l, z, zdim, tbase, n_trip_dense,
                                 g_rb, g_u_p, first=li == 0)
        # x0 depends on z only.  Geometry: rb = R(dist), u = vec / dist, vec = pos[t] - pos[s] + shift @ lattice
        g_u = torch.empty_like(g_u_p)
        g_u[ctx.order.long()] = g_u_p                       # p-space -> caller's edge order
        g_vec = ops.geometry_bwd(dist, uvec, g_rb, g_u, model.rbf.orb_table, model.rbf.rbf_kind,
                                 model.rbf.cutoff_net.kind, model.cutoff)
        g_vec_p = ops.gather_rows(g_vec, ctx.order)
        g_pos = _in_sum(g_vec_p, ctx, N) - ops.segment_sum(g_vec_p, ctx.out_ptr)
        forces.append(-g_pos)
    force = forces[0] if model.out_size == 1 else torch.stack(forces, dim=1)
    return out, force


# ---------------------------------------------------------------------------------------------------------------------
# Autograd forces of ONE graph sharded over ranks (lcaonet_amd/shard.py): every rank differentiates the energy of its
# owned nodes, E_r, with respect to everything it holds; E = sum_r E_r.  Two things cross ranks in the reverse sweep,
# both along the forward halo exchange run backwards (all-gather -> reduce-scatter):
#   * dE_r/dx of a GHOST node at the input of layer l belongs to the node's owner, which adds it to its own dE/dx before
#     it differentiates layer l-1 (L-1 reverse exchanges; the layer-0 input depends on z only and is dropped);
#   * dE_r/dpos of ghost atoms (they are end points of local edges) goes to the owners once, at the end.
# The computation is written as a generator that yields its communication requests, so that the same code runs under
# torch.distributed (run_sharded_forces) and in lock step for several ranks on one device (tests).
#   yields ("exchange", x_own (n_own, H))        expects the (n_ghost, H) ghost rows          [shard.exchange_ghosts]
#          ("reduce", partial (B, k))            expects the sum over ranks                   [shard.reduce_graph_outputs]
#          ("reverse", g_ghost (n_ghost, W))     expects the (n_own, W) sum of what every rank holds for the owned nodes
#                                                                                             [shard.reverse_exchange]
# ---------------------------------------------------------------------------------------------------------------------
@torch.no_grad()   # on a generator function torch re-enters no_grad at every resume and leaves it at every yield
def sharded_force_steps(model, plan):
    from .base import BaseMPNN
    from .lcaonet import EdgeContext

    _check_supported(model)
    graph = plan.graph
    z, pos = model.checked_z(graph), graph[GraphKeys.Pos]
    ops._dev(pos, "plan.graph.pos")
    n_own, e_own, n_loc = plan.n_own, plan.e_own, z.size(0)
    ctx = EdgeContext.build(graph[GraphKeys.Edge_idx], n_loc)
    graph = BaseMPNN.calc_atomic_distances(graph, return_vec=True)
    dist, uvec = graph[GraphKeys.Edge_dist], graph[GraphKeys.Edge_vec_st]
    uvec_p = ctx.uvec_p(uvec)
    rb = model.rbf(dist)
    x_loc, cst_pair, zdim = model.emb_layer.forward_pair_table(z)      # static inputs: ghosts need no exchange for x0
    l_of_d, n_l = model.shbf.l_of_d, model.n_l
    mask_pair = model.pair_valence_mask(zdim) if model.add_valence else None
    L = model.n_interaction
    saved = []
    for i, layer in enumerate(model.int_layers):
        x_own, sv = layer_forward_saving(layer, x_loc, cst_pair, rb, uvec_p, ctx, l_of_d, n_l, z, zdim, mask_pair,
                                         n_own, e_own)
        saved.append(sv)
        if i + 1 < L:
            ghosts = yield ("exchange", x_own)
            x_loc = torch.cat([x_own, ghosts], dim=0) if plan.n_ghost else x_own
    # output head on the owned nodes (lcaonet.py:290-298), pre-activations kept
    lin = model.out_layer.out_lin
    pre_a = ops.dense(x_own, lin[0].weight.detach(), lin[0].bias.detach())
    a = torch.nn.functional.silu(pre_a)
    pre_b = ops.dense(a, lin[2].weight.detach(), lin[2].bias.detach())
    b = torch.nn.functional.silu(pre_b)
    prop = ops.dense(b, lin[4].weight.detach())
    batch_own = graph[GraphKeys.Batch_idx][:n_own]
    z_own = z[:n_own]
    osz = model.out_size
    parts = [scatter(prop, batch_own, dim=0, dim_size=plan.n_graphs, reduce="sum"),
             scatter(torch.ones(n_own, 1, device=prop.device), batch_own, dim=0, dim_size=plan.n_graphs, reduce="sum")]
    if model.pp_layer.atomref is not None:
        parts.append(scatter(model.pp_layer.atomref[z_own].float().contiguous(), batch_own, dim=0,
                             dim_size=plan.n_graphs, reduce="sum"))
    tot = yield ("reduce", torch.cat(parts, dim=1))
    out, cnt = tot[:, :osz], tot[:, osz: osz + 1]
    ext = model.out_layer.is_extensive
    if not ext:
        out = out / cnt.clamp(min=1)
    if model.pp_layer.atomref is not None:
        aref = tot[:, osz + 1:]
        out = out + (aref if model.pp_layer.is_extensive else aref / cnt.clamp(min=1))
    if model.pp_layer.mean is not None:
        out = out + (model.pp_layer.mean.unsqueeze(0) * cnt if model.pp_layer.is_extensive
                     else model.pp_layer.mean.unsqueeze(0))
    node_scale = torch.ones(n_own, device=prop.device) if ext else (1.0 / cnt.clamp(min=1))[batch_own, 0]

    m_in = (ctx.in_ptr[1:] - ctx.in_ptr[:-1]).long()
    m_out = (ctx.out_ptr[1:] - ctx.out_ptr[:-1]).long()
    prod = m_in * m_out
    tbase = (torch.cumsum(prod, 0) - prod).contiguous()
    n_trip_dense = int(prod.sum().item()) if model.emb_size_conv not in ops.FUSED_WIDTHS else 1

    forces = []
    for i in range(osz):
        g_prop = torch.zeros_like(prop)
        g_prop[:, i] = node_scale
        g_b = ops.dense(g_prop, _t(lin[4].weight))
        g_a = ops.dense(ops.act_bwd(g_b, pre_b, ACT), _t(lin[2].weight))
        g_x = ops.dense(ops.act_bwd(g_a, pre_a, ACT), _t(lin[0].weight))
        g_rb = torch.zeros_like(rb)
        g_u_p = torch.zeros(ctx.n_edges, 3, device=pos.device)
        for l in reversed(range(L)):
            g_loc = layer_backward(model.int_layers[l], saved[l], g_x, rb, uvec_p, ctx, l_of_d, n_l, z, zdim, tbase,
                                   n_trip_dense, g_rb, g_u_p, first=l == 0)
            if l > 0:
                g_x = g_loc[:n_own] + (yield ("reverse", g_loc[n_own:].contiguous()))
        g_u = torch.empty_like(g_u_p)
        g_u[ctx.order.long()] = g_u_p
        g_vec = ops.geometry_bwd(dist, uvec, g_rb, g_u, model.rbf.orb_table, model.rbf.rbf_kind,
                                 model.rbf.cutoff_net.kind, model.cutoff)
        g_vec_p = ops.gather_rows(g_vec, ctx.order)
        g_pos = _in_sum(g_vec_p, ctx, n_loc) - ops.segment_sum(g_vec_p, ctx.out_ptr)      # every local atom
        g_own = g_pos[:n_own] + (yield ("reverse", g_pos[n_own:].contiguous()))
        forces.append(-g_own)
    force = forces[0] if osz == 1 else torch.stack(forces, dim=1)
    return out, force


def run_sharded_forces(model, plan, group=None, exchange_fn=None, reduce_fn=None, reverse_fn=None):
    """Drive sharded_force_steps on this rank with torch.distributed collectives (or the given stand-ins).
    Returns ((B, out) global result, forces of the OWNED atoms)."""
    from .. import shard

    fns = {"exchange": exchange_fn or (lambda x_own: shard.exchange_ghosts(plan, x_own, group)),
           "reduce": reduce_fn or (lambda part: shard.reduce_graph_outputs(part, group)),
           "reverse": reverse_fn or (lambda g_ghost: shard.reverse_exchange(plan, g_ghost, group))}
    steps = sharded_force_steps(model, plan)
    try:
        kind, payload = next(steps)
        while True:
            kind, payload = steps.send(fns[kind](payload))
    except StopIteration as done:
        return done.value
