"""Differentiable forward (training): parameter gradients of LCAONet through the HIP kernels.

Structure (all of it is exact, summation order aside):
  * Everything the embedding computes is a function of the atomic number (node vectors) or of the species pair
    (coefficient vectors, embed.py:246-249) — including BatchNorm in TRAIN mode, whose batch statistics over N nodes /
    E edges are count-weighted statistics over <= max_z species / (max_z+1)^2 pairs.  Those tiny table computations
    (embeddings, f_z, f_e, both BatchNorms, and f_coeffs of every layer on the ~1.4 k-row pair table) run as ordinary
    torch ops under autograd.
  * The L interaction layers on N nodes / E edges — the hot path — are ONE torch.autograd.Function.  Its forward is the
    pair-table HIP pipeline of model/lcaonet.py; its backward is the reverse sweep of model/forces.py extended with
    weight gradients.  Weight-gradient products G^T X are plain GEMMs and go through the vendor library (torch.mm).
  * The output heads (N-level MLPs) and the per-graph reduction are torch ops under autograd.
Training on autograd forces (create_graph=True) runs the twice-differentiable graph of model/double.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import ops
from ..data.keys import GraphKeys
from ..nn.base import Dense
from ..nn.scatter import csr_ptr

ACT = ops.ACT_SILU
PER_LAYER = ("node_weight.weight", "node_weight.bias", "f_three.0.weight", "basis_weight.weight", "f_node.0.weight",
             "f_node.0.bias", "f_node.2.weight", "f_node.2.bias", "out_weight.weight")


def _mlp(seq: nn.Sequential, x: Tensor) -> Tensor:
    """nn.Sequential of Dense / activation modules as plain torch ops (autograd-visible); used on species tables only."""
    for m in seq:
        x = F.linear(x, m.weight, m.bias) if isinstance(m, Dense) else m(x)
    return x


def _bn_weighted(bn: nn.BatchNorm1d, h: Tensor, w: Tensor, n_samples: int, training: bool,
                 update_stats: bool = True) -> Tensor:
    """BatchNorm1d over a population in which table row r occurs with relative frequency w[r] (sum w = 1):
    identical to applying the module to the expanded (n_samples, F) tensor, running-stat update included."""
    if training:
        mean = (w.unsqueeze(1) * h).sum(0)
        var = (w.unsqueeze(1) * (h - mean) ** 2).sum(0)
        with torch.no_grad():
            if bn.track_running_stats and update_stats:
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                bn.running_mean.mul_(1 - mom).add_(mom * mean)
                bn.running_var.mul_(1 - mom).add_(mom * var * (n_samples / max(n_samples - 1, 1)))
                bn.num_batches_tracked += 1
    else:
        mean, var = bn.running_mean, bn.running_var
    return (h - mean) / torch.sqrt(var + bn.eps) * bn.weight + bn.bias


def species_tables_autograd(model, z: Tensor, idx_s: Tensor, idx_t: Tensor, update_stats: bool = True, count_reduce=None,
                            n_own: int | None = None):
    """x_tab (max_z+1, H) and cst_pair ((max_z+1)^2, D, C), differentiable w.r.t. every embedding parameter.
    update_stats=False: train-mode batch statistics without touching the BatchNorm running buffers (extra passes).
    count_reduce / n_own (one graph sharded over ranks): z holds this rank's local atoms (the first n_own are OWNED),
    idx_s / idx_t its owned edges; count_reduce(counts) sums the species / pair histograms over the ranks — the BatchNorm batch statistics of the whole
    graph are functions of those counts alone, so every rank gets the same tables (embed.py:175,232)."""
    emb = model.emb_layer
    dev = z.device
    Z = emb.max_z
    zz = torch.arange(1, Z + 1, device=dev)
    z_embed = emb.z_embed(zz)
    node_z, coeff_z = torch.split(z_embed, [emb.emb_size, emb.emb_size_coeff], dim=-1)
    e_embed = emb.e_embed(zz)
    D = e_embed.shape[1]
    if emb.use_elec:
        node_e, coeff_e = torch.split(e_embed, [emb.emb_size_node_e, emb.emb_size_coeff], dim=-1)
        node_in = torch.cat([node_z, node_e.sum(1) / math.sqrt(D)], dim=-1)
    else:
        coeff_e, node_in = e_embed, node_z
    zd = Z + 1
    z_cnt = z if n_own is None else z[:n_own]
    N, E = z_cnt.numel(), idx_s.numel()
    # histograms by integer counting (torch.bincount): exact, the same in every run — no floating-point atomics
    cnt_z = torch.bincount(z_cnt, minlength=zd).to(torch.float32)
    pair = z[idx_s] * zd + z[idx_t]
    cnt_p = torch.bincount(pair, minlength=zd * zd).to(torch.float32)
    if count_reduce is not None:
        tot = count_reduce(torch.cat([cnt_z, cnt_p]))
        cnt_z, cnt_p = tot[:zd], tot[zd:]
        N, E = int(round(float(cnt_z.sum()))), int(round(float(cnt_p.sum())))
    # node table
    h = _mlp(emb.node_embed.f_enc, node_in)                                    # (Z, H)
    x_tab = _bn_weighted(emb.node_embed.bn, h, cnt_z[1:] / N, N, model.training, update_stats)
    x_tab = torch.cat([torch.zeros_like(x_tab[:1]), x_tab], dim=0)
    # pair table
    ce = emb.coeff_embed
    C = emb.emb_size_coeff
    wz = ce.f_z[0].weight
    U, Vt = F.linear(coeff_z, wz[:, :C]), F.linear(coeff_z, wz[:, C:])          # (Z, C) each
    fe = _mlp(ce.f_e, coeff_e)                                                   # (Z, D, C)
    raw = fe.unsqueeze(0) * (1.0 + U.unsqueeze(1).unsqueeze(2) + Vt.unsqueeze(0).unsqueeze(2))   # (Zs, Zt, D, C)
    wp = (cnt_p.view(zd, zd)[1:, 1:] / max(E, 1)).reshape(-1)
    cst = _bn_weighted(ce.bn, raw.reshape(Z * Z, D * C), wp, E, model.training, update_stats).reshape(Z, Z, D, C)
    cst_pair = torch.cat([torch.zeros(1, zd, D, C, device=dev),
                          torch.cat([torch.zeros(Z, 1, D, C, device=dev), cst], dim=1)], dim=0)
    return x_tab, cst_pair.reshape(zd * zd, D, C), pair, zd


class _Geom:
    """Static (non-differentiated) per-forward data handed to the autograd Function."""

    def __init__(self, model, graph):
        from .base import BaseMPNN
        from .lcaonet import EdgeContext

        z = model.checked_z(graph)
        self.z = z
        self.N = z.numel()
        self.ctx = EdgeContext.build(graph[GraphKeys.Edge_idx], self.N)
        graph = BaseMPNN.calc_atomic_distances(graph, return_vec=True)
        self.dist, self.uvec = graph[GraphKeys.Edge_dist], graph[GraphKeys.Edge_vec_st]
        self.uvec_p = self.ctx.uvec_p(self.uvec)
        self.rb = model.rbf(self.dist)
        self.l_of_d, self.n_l = model.shbf.l_of_d, model.n_l
        # nodes grouped by species (gradient of the x_tab gather)
        self.order_z = torch.argsort(z, stable=True)
        self.ptr_z = csr_ptr(z[self.order_z], model.emb_layer.max_z + 1)

    def pair_lists(self, pair: Tensor, zd: int, chunk: int = 64):
        """p-space edges sorted by species pair, cut into chunks of <= `chunk` edges that never span two pairs."""
        ctx = self.ctx
        pair_p = pair[ctx.order.long()]
        perm = torch.argsort(pair_p, stable=True)
        ptr = csr_ptr(pair_p[perm], zd * zd).long()
        n_chunks = (ptr[1:] - ptr[:-1] + chunk - 1) // chunk
        cptr = torch.zeros(zd * zd + 1, dtype=torch.long, device=pair.device)
        cptr[1:] = torch.cumsum(n_chunks, 0)
        total = (pair.numel() + chunk - 1) // chunk + zd * zd             # static upper bound, no sync
        cid = torch.arange(total + 1, device=pair.device)
        seg = torch.searchsorted(cptr[1:], cid, right=True).clamp(max=zd * zd - 1)
        lo = torch.minimum(ptr[seg] + (cid - cptr[seg]) * chunk, ptr[seg + 1])
        lo = torch.where(cid < cptr[-1], lo, ptr[-1].expand_as(lo))
        self.perm_pair = perm.to(torch.int32)
        self.chunk_ptr = torch.cat([lo[:total], ptr[-1:]]).to(torch.int32)
        self.pair_cptr = torch.clamp(cptr, max=total).to(torch.int32)
        self.zd = zd


def _mm_t(g: Tensor, x: Tensor) -> Tensor:
    """weight gradient of y = x W^T:  dW = g^T x  (lcao_wgrad: fp32 MFMA, deterministic; widths that are not multiples
    of 64 use the vendor GEMM)."""
    return ops.wgrad(g, x)


class LCAOStackFn(torch.autograd.Function):
    """x_L = interaction_L(... interaction_1(x_tab[z]; Tf_1) ...; Tf_L) with hand-written HIP forward and backward."""

    @staticmethod
    def forward(fctx, geom: _Geom, n_layers: int, x_tab: Tensor, *args: Tensor):
        Tfs, params = args[:n_layers], args[n_layers:]
        ctx, rb, z = geom.ctx, geom.rb, geom.z
        x = ops.gather_rows(x_tab.detach().contiguous(), z)
        saved = []
        for i in range(n_layers):
            Wnw, bnw, W3, Wb, Wn1, bn1, Wn2, bn2, Wout = [p.detach() for p in params[i * 9:(i + 1) * 9]]
            V = Wb.shape[0]
            xs = ops.dense(x, Wnw[:V], bnw[:V])
            xk = ops.dense(x, Wnw[V:], bnw[V:])
            gate = torch.sigmoid(xk)
            Ps = ops.dense(xs, Wn1[:, :V].contiguous(), bn1)
            Pt = ops.dense(xs, Wn1[:, V:].contiguous())
            Tf = Tfs[i].detach().contiguous()
            pc = lambda T: ops.pair_contract(T, rb, ctx.order, geom.l_of_d, z, ctx.src_p, ctx.dst_p, gate, geom.n_l, geom.zd,
                                             visit=geom.perm_pair)   # edges walked pair by pair: table blocks stay in L1
            valence = Tf.shape[-1] == 2 * V     # add_valence: [plain half | masked valence half] (lcaonet.py:174-183)
            Bv = None
            if valence:
                Tc, Tv = Tf[..., :V].contiguous(), Tf[..., V:].contiguous()
                Tf = (Tc + Tv, Tc, Tv)
                Ag, gram, _ = pc(Tf[0])
                B, Bv = pc(Tc)[2], pc(Tv)[2]
            else:
                Ag, gram, B = pc(Tf)
            tbw = ops.triplet_gemm(Ag, gram, geom.uvec_p, ctx.in_ptr, ctx.in_pos, ctx.out_ptr)
            if V in ops.FUSED_WIDTHS and not valence:
                bw, ft = ops.edge_lcao_weight(tbw, B, W3, Wb, save=True)
                agg, pre2 = ops.edge_node_message_agg(bw, Ps, Pt, ctx.src_p, ctx.dst_p, ctx.out_ptr, Wn2, bn2,
                                                      ctx.column_hint(ctx.out_ptr, ctx.n_edges), save=True)
                pre1 = None
            else:
                ft = ops.dense(tbw, W3)
                lw = ops.edge_combine(ft, B, Bv)
                bw = ops.dense(lw, Wb)
                pre1 = ops.pair_pre(Ps, Pt, ctx.src_p, ctx.dst_p)
                h1 = F.silu(pre1)
                pre2 = ops.dense(h1, Wn2, bn2)
                agg = ops.segment_sum(bw, ctx.out_ptr, F.silu(pre2))
            x_new = ops.dense(agg, Wout, None, ops.ACT_NONE, x)
            saved.append(dict(x=x, xs=xs, xk=xk, gate=gate, Tf=Tf, Ag=Ag, gram=gram, B=B, Bv=Bv, tbw=tbw, ft=ft, bw=bw,
                              pre1=pre1, pre2=pre2, agg=agg, Ps=Ps, Pt=Pt))
            x = x_new
        fctx.geom, fctx.n_layers, fctx.saved = geom, n_layers, saved
        fctx.params = [p.detach() for p in params]
        fctx.x_tab_shape = x_tab.shape
        return x

    @staticmethod
    def backward(fctx, g_x: Tensor):
        geom, L, saved, params = fctx.geom, fctx.n_layers, fctx.saved, fctx.params
        ctx, rb, z = geom.ctx, geom.rb, geom.z
        N, E = geom.N, ctx.n_edges
        g_x = g_x.contiguous().float()
        g_Tf = [None] * L
        g_params = [None] * (9 * L)
        g_rb_sink = torch.zeros_like(rb)   # dE/drb is produced by the kernel; positions are not differentiated here
        m_in = (ctx.in_ptr[1:] - ctx.in_ptr[:-1]).long()
        m_out = (ctx.out_ptr[1:] - ctx.out_ptr[:-1]).long()
        prod = m_in * m_out
        tbase = (torch.cumsum(prod, 0) - prod).contiguous()
        n_dense = None   # only the any-V fallback of the three-body backward needs the scratch size (host sync)
        for i in reversed(range(L)):
            Wnw, bnw, W3, Wb, Wn1, bn1, Wn2, bn2, Wout = params[i * 9:(i + 1) * 9]
            sv = saved[i]
            V = Wb.shape[0]
            valence = sv["Bv"] is not None
            gp = g_params
            # x' = x + agg Wout^T
            gp[i * 9 + 8] = _mm_t(g_x, sv["agg"])
            g_agg = ops.dense(g_x, Wout.t().contiguous())
            fused = V in ops.FUSED_WIDTHS and not valence
            if fused:
                g_bw, g_pre1, g_pre2 = ops.message_bwd_fused(g_agg, ctx.src_p, ctx.dst_p, sv["bw"], sv["pre2"], sv["Ps"],
                                                             sv["Pt"], Wn2.t().contiguous(), want_g_pre2=True)
                h1 = ops.pair_silu(sv["Ps"], sv["Pt"], ctx.src_p, ctx.dst_p)
            else:
                g_bw, g_pre2 = ops.message_bwd(g_agg, ctx.src_p, sv["bw"], sv["pre2"])
                h1 = F.silu(sv["pre1"])
                g_h1 = ops.dense(g_pre2, Wn2.t().contiguous())
                g_pre1 = ops.act_bwd(g_h1, sv["pre1"], ACT)
            gp[i * 9 + 6] = _mm_t(g_pre2, h1)
            gp[i * 9 + 7] = g_pre2.sum(0)
            del h1, g_pre2
            g_Ps = ops.segment_sum(g_pre1, ctx.out_ptr)
            g_Pt = ops.segment_sum_gather(g_pre1, ctx.in_pos, ctx.in_ptr)
            lw = ops.edge_combine(sv["ft"], sv["B"], sv["Bv"])
            gp[i * 9 + 3] = _mm_t(g_bw, lw)
            del lw
            if fused:
                g_B, g_tbw, g_ft = ops.lcao_weight_bwd_fused(g_bw, sv["B"], sv["ft"], Wb.t().contiguous(),
                                                             W3.t().contiguous(), want_g_ft=True)
            else:
                g_lw = ops.dense(g_bw, Wb.t().contiguous())
                if valence:
                    g_B, g_Bv, g_ft = ops.combine_bwd(g_lw, sv["B"], sv["ft"], sv["Bv"])
                else:
                    g_B, g_ft = ops.combine_bwd(g_lw, sv["B"], sv["ft"])
                g_tbw = ops.dense(g_ft, W3.t().contiguous())
            gp[i * 9 + 2] = _mm_t(g_ft, sv["tbw"])
            if V not in ops.FUSED_WIDTHS and n_dense is None:
                n_dense = int(prod.sum().item())
            g_Ag, g_gram, _, _ = ops.triplet_bwd(g_tbw, sv["Ag"], sv["gram"], geom.uvec_p, ctx.src_p, ctx.dst_p, ctx.in_ptr,
                                                 ctx.in_pos, ctx.out_ptr, tbase, n_dense or 1, E, N)
            D = rb.shape[1]

            def table_grad(g_A):
                partial = ops.pair_table_grad(g_A, rb, ctx.order, geom.l_of_d, geom.perm_pair, geom.chunk_ptr)
                return ops.segment_sum(partial.reshape(partial.shape[0], D * V), geom.pair_cptr).reshape(-1, D, V)

            Te = sv["Tf"][0] if valence else sv["Tf"]
            if not valence and V <= 128 and D <= ops.ASIDE_PAIRS_MAX_ORB:
                # contraction backward and the table gradient in ONE pass over the pair-sorted chunks (dE/dA never written)
                _, g_gate_part, partial = ops.aside_pairs(1, Te, rb, geom.l_of_d, z, ctx.src_p, ctx.dst_p, sv["gate"], geom.n_l,
                                                          geom.zd, geom.perm_pair, geom.chunk_ptr, g_Ag, g_B, g_gram,
                                                          order=ctx.order, gram_tri=True)
                g_Tf[i] = ops.segment_sum(partial.reshape(partial.shape[0], D * V), geom.pair_cptr).reshape(-1, D, V)
                g_A = None
            else:
                g_gate_part, g_A = ops.pair_contract_bwd(Te, rb, ctx.order, geom.l_of_d, z, ctx.src_p, ctx.dst_p, sv["gate"],
                                                         g_Ag, torch.zeros_like(g_B) if valence else g_B, g_gram, E, geom.n_l,
                                                         geom.zd, g_rb_sink, want_gA=True)
            if valence:   # A sees Tc + Tv, B sees Tc, Bv sees Tv: the shared part of g_A plus each half's own g_B
                g_Tf[i] = torch.cat([table_grad(g_A + g_B.unsqueeze(1)), table_grad(g_A + g_Bv.unsqueeze(1))], dim=-1)
            elif g_A is not None:
                g_Tf[i] = table_grad(g_A)
            g_gate = ops.segment_sum(g_gate_part, ctx.out_ptr)
            g_xk = ops.act_bwd(g_gate, sv["xk"], ops.ACT_SIGMOID)
            xs = sv["xs"]
            gp[i * 9 + 4] = torch.cat([_mm_t(g_Ps, xs), _mm_t(g_Pt, xs)], dim=1)
            gp[i * 9 + 5] = g_Ps.sum(0)
            g_xs = ops.dense(g_Ps, Wn1[:, :V].t().contiguous())
            g_xs = ops.dense(g_Pt, Wn1[:, V:].t().contiguous(), None, ops.ACT_NONE, g_xs)
            g_nw = torch.cat([g_xs, g_xk], dim=1)
            gp[i * 9 + 0] = _mm_t(g_nw, sv["x"])
            gp[i * 9 + 1] = g_nw.sum(0)
            g_x = ops.dense(g_nw, Wnw.t().contiguous(), None, ops.ACT_NONE, g_x)
        g_x_tab = ops.segment_sum_gather(g_x, geom.order_z.to(torch.int32), geom.ptr_z)
        return (None, None, g_x_tab.reshape(fctx.x_tab_shape), *g_Tf, *g_params)


def unsupported_reason(model) -> str | None:
    """Why the differentiable path cannot serve this model, or None."""
    for layer in model.int_layers:
        if not all(type(layer.f_node[k]) is nn.SiLU for k in (1, 3)):
            return "the differentiable path is built for the default SiLU activation"
    return None


def forward_differentiable(model, graph, update_stats: bool = True, energy_only: bool = False):
    """LCAONet.forward with an autograd graph to every parameter (training).  Same outputs as the inference path.
    energy_only: return the (B, out) property alone (no force head); update_stats as species_tables_autograd."""
    if model.regress_forces and not model.direct_forces and not energy_only:
        return forward_force_training(model, graph)
    if unsupported_reason(model) is not None and not energy_only:
        from .double import forward_generic       # e.g. non-SiLU activations: the generic twice-differentiable graph

        return forward_generic(model, graph, update_stats)
    if unsupported_reason(model) is not None:
        raise NotImplementedError(unsupported_reason(model))
    z = model.checked_z(graph)
    idx_s, idx_t = graph[GraphKeys.Edge_idx]
    ops._dev(graph[GraphKeys.Pos], "graph.pos")
    batch_idx = graph.get(GraphKeys.Batch_idx)
    with torch.no_grad():
        geom = _Geom(model, graph)
    x_tab, cst_pair, pair, zd = species_tables_autograd(model, z, idx_s, idx_t, update_stats)
    with torch.no_grad():
        geom.pair_lists(pair, zd)
    L = model.n_interaction
    Tfs = [_mlp(layer.f_coeffs, cst_pair) for layer in model.int_layers]
    if model.add_valence:   # mask the valence half by the neighbour species (differentiable: plain torch on the table)
        V = model.emb_size_conv
        mp = model.pair_valence_mask(zd).unsqueeze(-1)
        Tfs = [torch.cat([T[..., :V], T[..., V:] * mp], dim=-1) for T in Tfs]
    params = []
    for layer in model.int_layers:
        sd = dict(layer.named_parameters())
        params += [sd[k] for k in PER_LAYER]
    x = LCAOStackFn.apply(geom, L, x_tab, *Tfs, *params)
    # heads: N-level MLPs and the per-graph reduction as torch ops under autograd
    ol = model.out_layer
    prop = _mlp(ol.out_lin, x)
    # Per-graph sums and the direct-force scatter as sorted-segment reductions (double.py: gather <-> scatter_sum, each the
    # other's backward): rows are added in a fixed order, so outputs AND gradients are the same bits in every run.  (These
    # were index_add_ calls — atomicAdd on ROCm — until round 3.)
    from .double import IndexPlan, gather, scatter_sum
    if batch_idx is not None:
        from .base import num_graphs
        B = num_graphs(graph, batch_idx)
        bplan = IndexPlan(batch_idx, B)
        out = scatter_sum(prop, bplan)
        cnt = (bplan.ptr[1:] - bplan.ptr[:-1]).to(prop.dtype)
        if not ol.is_extensive:
            out = out / cnt.clamp(min=1).unsqueeze(1)
    else:
        B, bplan = 1, None
        out = prop.sum(0, keepdim=True) if ol.is_extensive else prop.mean(0, keepdim=True)
        cnt = torch.full((1,), float(z.numel()), device=prop.device)
    pp = model.pp_layer
    if pp.atomref is not None:
        aref = pp.atomref[z].float().contiguous()
        a = scatter_sum(aref, bplan) if bplan is not None else aref.sum(0, keepdim=True)
        out = out + (a if pp.is_extensive else a / cnt.clamp(min=1).unsqueeze(1))
    if pp.mean is not None:
        out = out + (pp.mean.unsqueeze(0) * cnt.unsqueeze(1) if pp.is_extensive else pp.mean.unsqueeze(0))
    if model.regress_forces and not energy_only:  # direct forces head (lcaonet.py:303-308)
        splan, tplan = IndexPlan(idx_s, z.numel()), IndexPlan(idx_t, z.numel())
        f = _mlp(ol.out_lin_force, torch.cat([gather(x, splan), gather(x, tplan)], dim=-1)) * geom.uvec
        return out, scatter_sum(f.contiguous(), splan)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Training on autograd forces (regress_forces=True, direct_forces=False with autograd on; reference lcaonet.py:310-317
# with create_graph=True).  Default: the EXACT double backward — the energy is assembled from twice-differentiable
# primitives (model/double.py) and F = -dE/dpos is taken with create_graph=True, so forces carry a real graph in train and
# eval mode.  `model.force_training = "central-difference"` selects the older cross-check instead: the force loss
# reaches the parameters through a Hessian-vector product by CENTRAL DIFFERENCES of first-order passes,
#     dL/dtheta (force part) = -d/dtheta [ E(x + h v) - E(x - h v) ] / (2 h)  + O(h^2),   v = dL/dF,
# two more differentiable forwards at displaced positions, h putting the largest displacement at 0.01 A (an
# approximation with a step size: 2e-6 (eval) / 2e-5 (train mode) from the exact result).  `None` makes the
# combination raise.
# ---------------------------------------------------------------------------------------------------------------------
FORCE_TRAINING_MODES = ("exact", "central-difference")
_GRAPH_KEYS = (GraphKeys.Z, GraphKeys.Pos, GraphKeys.Edge_idx, GraphKeys.Edge_shift, GraphKeys.Lattice, GraphKeys.Batch_idx)


def _displaced(graph, pos: Tensor):
    from ..data.graph import Data

    g = Data(**{k: graph[k] for k in _GRAPH_KEYS if graph.get(k) is not None})
    g[GraphKeys.Pos] = pos
    return g


class _ForceHVP(torch.autograd.Function):
    """Identity on the (detached) forces whose backward hands the parameters the central-difference HVP above."""

    @staticmethod
    def forward(ctx, force: Tensor, handle, *params: Tensor):
        ctx.handle = handle
        return force.clone()

    @staticmethod
    def backward(ctx, g_force: Tensor):
        model, graph, params, max_disp = ctx.handle
        pos = graph[GraphKeys.Pos].detach()
        total = None
        with torch.enable_grad():
            for i in range(model.out_size):
                v = (g_force if model.out_size == 1 else g_force[:, i]).detach().to(pos.dtype)
                vmax = float(v.abs().max())
                if vmax == 0.0 or not math.isfinite(vmax):
                    continue
                h = max_disp / vmax
                e_p = forward_differentiable(model, _displaced(graph, pos + h * v), False, True)[:, i].sum()
                e_m = forward_differentiable(model, _displaced(graph, pos - h * v), False, True)[:, i].sum()
                s_i = -(e_p - e_m) / (2.0 * h)
                total = s_i if total is None else total + s_i
            if total is None:
                grads = [None] * len(params)
            else:
                grads = torch.autograd.grad(total, params, allow_unused=True)
        return (None, None, *grads)


def forward_force_training(model, graph, max_displacement: float = 0.01):
    """(out, force) in train mode with gradients to every parameter from both; see the block comment above."""
    from .forces import forward_with_autograd_forces

    mode = getattr(model, "force_training", FORCE_TRAINING_MODES[0])
    if mode not in FORCE_TRAINING_MODES:
        raise NotImplementedError(
            f"training on autograd forces is switched off (model.force_training = {mode!r}; allowed: {FORCE_TRAINING_MODES})")
    if mode == "exact":
        from .double import forward_exact_forces

        return forward_exact_forces(model, graph)
    out = forward_differentiable(model, graph, True, True)          # energies: exact first-order path, BN stats updated
    z = model.checked_z(graph)
    idx_s, idx_t = graph[GraphKeys.Edge_idx]
    with torch.no_grad():                                           # forces at the same train-mode batch statistics
        x_tab, cst_pair, _, zd = species_tables_autograd(model, z, idx_s, idx_t, False)
        _, force = forward_with_autograd_forces(model, _displaced(graph, graph[GraphKeys.Pos]),
                                                tables=(x_tab.contiguous(), cst_pair.contiguous(), zd))
    params = [p for p in model.parameters() if p.requires_grad]
    force = _ForceHVP.apply(force, (model, graph, params, max_displacement), *params)
    return out, force
