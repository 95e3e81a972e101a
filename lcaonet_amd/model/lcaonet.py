"""LCAONet on MI355X: same constructor, forward(graph) contract, graph keys and state_dict names as the
reference's lcaonet/model/lcaonet.py, with the per-edge interaction hot path executed by hand-written HIP
kernels (liblcaonet_hip.so) in a restructured dataflow:

  reference (lcaonet.py:130-216)                       here
  ---------------------------------------------------  ------------------------------------------------------
  f_coeffs(cst) -> (E,D,V) tensor, gathered per        lcao_coeff_mlp_contract: both f_coeffs layers on the
  triplet as cks (T,D,V) = 65-80 GB at 1 M edges       matrix cores, folded with rb into A (E,n_l,V), B (E,V);
                                                       (E,D,V) and every T-sized tensor never exist
  rb[ks]*shb, einsum, normalize, sigmoid(xk[k]),       lcao_triplet_sum: per centre node, Y_l from unit vectors
  scatter T->E                                         on the fly, exact L2 norm, gate, segmented sum
  f_three, cst + cst*ft, einsum, normalize,            lcao_edge_lcao_weight (register-to-register MFMA chain)
  basis_weight
  f_node(cat[x_s, x_t]) * ..., scatter E->N            node-level split of f_node.0 (Ps/Pt), lcao_edge_node_message,
                                                       lcao_segment_sum (edges sorted by centre, no atomics)

Identities used (summation order changes only): shb[t,d] depends on d through l(d) alone (shbf.py:54-63), so
sum_d rb*shb*c = sum_l Y_l * A[e',l,:];  cst + cst*ft summed against rb = B * (1 + ft);  a Dense on
cat[x_s, x_t] = x_s W_a^T + x_t W_b^T.
"""
from __future__ import annotations

from collections.abc import Callable
from dataclasses import dataclass

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..atomistic.info import ElecInfo
from ..data.keys import GraphKeys
from ..nn.base import Dense, act_code, drop_weight_caches, inference_only, run_mlp
from ..nn.cutoff import BaseCutoff
from ..nn.embed import EmbedCoeffs, EmbedElec, EmbedNode, EmbedZ, ValenceMask
from ..nn.post import PostProcess
from ..nn.rbf import BaseRadialBasis
from ..nn.scatter import csr_ptr, scatter
from ..nn.shbf import SphericalHarmonicsBasis
from ..utils.resolve import activation_resolver, cutoffnet_resolver, init_resolver, rbf_resolver
from .base import BaseMPNN, batch_plan, num_graphs


@dataclass
class EdgeContext:
    """Per-forward index structures in HBM (all int32, built without a host sync).

    p-space = edges stably sorted by centre node (`order[p]` = caller's edge id); reference-produced graphs
    (data/convert.py:135-153) are already grouped by centre, so `order` is then the identity."""
    n_nodes: int
    n_edges: int
    order: Tensor      # (E)   p -> orig edge id
    src_p: Tensor      # (E)   centre node of edge p (ascending)
    dst_p: Tensor      # (E)   neighbour node of edge p
    out_ptr: Tensor    # (N+1) edges p with centre s: [out_ptr[s], out_ptr[s+1])
    in_ptr: Tensor     # (N+1) in-edge list of node s: in_pos[in_ptr[s] : in_ptr[s+1]]
    in_pos: Tensor     # (E)   p-indices of edges k->s grouped by s, ordered by (k, orig id)
    src_o: Tensor      # (E)   idx_s as int32, orig order
    dst_o: Tensor      # (E)   idx_t as int32, orig order
    _uvec_p: Tensor | None = None
    _uvec_src: Tensor | None = None
    _col_hints: dict | None = None

    def column_hint(self, out_ptr: Tensor, n_edges: int) -> Tensor:
        """Device-side flag for ops.edge_node_message_agg (per-node-column form when the degrees are even); depends on the
        edge list only: one tiny launch per (context, number of centres), no host synchronisation."""
        if self._col_hints is None:
            self._col_hints = {}
        key = (out_ptr.numel(), n_edges)
        if key not in self._col_hints:
            self._col_hints[key] = ops.node_column_hint(out_ptr, n_edges)
        return self._col_hints[key]

    def uvec_p(self, uvec: Tensor) -> Tensor:
        """Unit vectors permuted to p-space (computed once per forward and shared by the layers; a context that
        outlives a forward — model/capture.py — sees a new `uvec` tensor next time and permutes again)."""
        if self._uvec_p is None or self._uvec_src is not uvec:
            self._uvec_p = ops.gather_rows(uvec, self.order)
            self._uvec_src = uvec
        return self._uvec_p

    @staticmethod
    def build(edge_index: Tensor, n_nodes: int) -> "EdgeContext":
        idx_s, idx_t = edge_index[0], edge_index[1]
        E = idx_s.numel()
        if E >= 2**31 or n_nodes >= 2**31:
            raise ValueError("graph too large for int32 indexing on one GPU")
        c = ops.build_csr(idx_s, idx_t, n_nodes)   # lcao_build_csr: two stable radix sorts + lower bounds, no sync
        return EdgeContext(n_nodes=n_nodes, n_edges=E, **c)


class LCAOEmbedding(nn.Module):
    """Node vectors x (N, H) and per-edge coefficient vectors cst (E, D, C)  (reference lcaonet.py:30-74)."""

    def __init__(self, emb_size: int, emb_size_coeff: int, elec_info: ElecInfo, max_z: int, use_elec: bool,
                 extend_orb: bool, activation: nn.Module = nn.SiLU(),
                 weight_init: Callable[[Tensor], Tensor] | None = None):
        super().__init__()
        self.emb_size = emb_size
        self.emb_size_coeff = emb_size_coeff
        self.use_elec = use_elec
        self.max_z = max_z
        self.z_embed = EmbedZ(emb_size + emb_size_coeff, max_z)
        self.emb_size_node_e = emb_size if use_elec else 0
        self.e_embed = EmbedElec(self.emb_size_node_e + emb_size_coeff, elec_info, extend_orb)
        self.node_embed = EmbedNode(emb_size, emb_size, use_elec, self.emb_size_node_e, activation, weight_init)
        self.coeff_embed = EmbedCoeffs(emb_size_coeff, emb_size_coeff, emb_size_coeff, elec_info.n_orb, activation,
                                       weight_init)

    # weight-derived caches (see LCAONet.invalidate_caches): dropped by everything that rewrites weights wholesale
    def train(self, mode: bool = True):
        drop_weight_caches(self)
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        drop_weight_caches(self)
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, z: Tensor, idx_s: Tensor, idx_t: Tensor) -> tuple[Tensor, Tensor]:
        z_embed = self.z_embed(z)
        node_z, coeff_z = torch.split(z_embed, [self.emb_size, self.emb_size_coeff], dim=-1)
        e_embed = self.e_embed(z)
        if self.use_elec:
            node_e, coeff_e = torch.split(e_embed, [self.emb_size_node_e, self.emb_size_coeff], dim=-1)
            x = self.node_embed(node_z, node_e)
        else:
            coeff_e = e_embed
            x = self.node_embed(node_z)
        cst = self.coeff_embed(coeff_z, coeff_e, idx_s, idx_t)
        return x, cst

    @torch.no_grad()
    def species_tables(self, dev) -> dict:
        """Eval-mode per-species tables (rows indexed by atomic number 0..max_z, row 0 unused): every quantity of
        this module is a function of the atomic number (x) or of the (z_s, z_t) pair (cst), so the MLPs run on max_z
        rows:  cst = bn(fe[z_t] * (1 + f_z([zc[z_s], zc[z_t]])))  (embed.py:246-249)."""
        assert not self.training
        # weights only (no input reaches these tables): kept until a parameter or buffer of this module changes
        key = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        cached = self.__dict__.get("_species")
        if cached is not None and cached[0] == key:
            return cached[1]
        zz = torch.arange(1, self.max_z + 1, device=dev)
        z_embed = self.z_embed(zz)
        node_z, coeff_z = torch.split(z_embed, [self.emb_size, self.emb_size_coeff], dim=-1)
        e_embed = self.e_embed(zz)
        if self.use_elec:
            node_e, coeff_e = torch.split(e_embed, [self.emb_size_node_e, self.emb_size_coeff], dim=-1)
            x_tab = self.node_embed(node_z.contiguous(), node_e.contiguous())
        else:
            coeff_e = e_embed
            x_tab = self.node_embed(node_z.contiguous())
        C = self.emb_size_coeff
        ce = self.coeff_embed
        U, Vt = ce.node_halves(coeff_z.contiguous())      # centre / neighbour half of f_z
        FE = run_mlp(ce.f_e, coeff_e.contiguous())        # (max_z, D, C)
        bn = ce.bn
        scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias.detach() - bn.running_mean * scale
        pad = lambda t: torch.cat([torch.zeros_like(t[:1]), t], dim=0).contiguous()
        tables = dict(x=pad(x_tab), FE=pad(FE), U=pad(U), Vt=pad(Vt), scale=scale.contiguous(), shift=shift.contiguous())
        if not torch.cuda.is_current_stream_capturing():     # a capture must record the kernels that fill its own copy
            self.__dict__["_species"] = (key, tables)
        return tables

    @torch.no_grad()
    def forward_species(self, z: Tensor, ctx: EdgeContext) -> tuple[Tensor, Tensor]:
        """Eval-mode forward: node vectors by a row gather, cst (E, D, C) by ONE fused kernel (4*D*C bytes/edge)."""
        t = self.species_tables(z.device)
        x = ops.gather_rows(t["x"], z)
        cst = ops.embed_coeffs(t["FE"], t["U"], t["Vt"], t["scale"], t["shift"], z, ctx.src_o, ctx.dst_o)
        return x, cst

    @torch.no_grad()
    def forward_pair_table(self, z: Tensor) -> tuple[Tensor, Tensor, int]:
        """Species-pair form: x (N, H) and cst_pair ((max_z+1)^2, D, C) with row z_s*(max_z+1)+z_t —
        the same kernel as forward_species run on the (max_z+1)^2 synthetic 'edges' (z_s, z_t)."""
        dev = z.device
        t = self.species_tables(dev)
        x = ops.gather_rows(t["x"], z)
        zd = self.max_z + 1
        cst_pair = t.get("cst_pair")          # weights only, like the rest of `t`: kept with it
        if cst_pair is None:
            zfake = torch.arange(zd, device=dev)
            src = torch.arange(zd, device=dev, dtype=torch.int32).repeat_interleave(zd)
            dst = torch.arange(zd, device=dev, dtype=torch.int32).repeat(zd)
            cst_pair = ops.embed_coeffs(t["FE"], t["U"], t["Vt"], t["scale"], t["shift"], zfake, src, dst)
            if not torch.cuda.is_current_stream_capturing():
                t["cst_pair"] = cst_pair
        return x, cst_pair, zd


def split_valence_table(Tf: Tensor, mask_pair: Tensor, V: int) -> tuple[Tensor, Tensor]:
    """f_coeffs(pair table) (P, D, 2V) -> (Tc, Tv) = (plain half, valence half masked by the neighbour species)."""
    return Tf[..., :V].contiguous(), (Tf[..., V:] * mask_pair.unsqueeze(-1)).contiguous()


class LCAOInteraction(nn.Module):
    """One LCAOConv message-passing layer (reference lcaonet.py:77-216); same parameters and names."""

    def __init__(self, emb_size: int, emb_size_coeff: int, emb_size_conv: int, add_valence: bool = False,
                 activation: nn.Module = nn.SiLU(), weight_init: Callable[[Tensor], Tensor] | None = None):
        super().__init__()
        self.emb_size = emb_size
        self.emb_size_coeff = emb_size_coeff
        self.emb_size_conv = emb_size_conv
        self.add_valence = add_valence
        self.node_weight = Dense(emb_size, 2 * emb_size_conv, True, weight_init)
        out_size = 2 * emb_size_conv if add_valence else emb_size_conv
        self.f_coeffs = nn.Sequential(  # no bias: zero coefficient vectors stay zero
            Dense(emb_size_coeff, emb_size_conv, False, weight_init), activation,
            Dense(emb_size_conv, out_size, False, weight_init), activation,
        )
        self.f_three = nn.Sequential(Dense(emb_size_conv, out_size, False, weight_init))
        self.basis_weight = Dense(emb_size_conv, emb_size_conv, False, weight_init)
        self.f_node = nn.Sequential(
            Dense(2 * emb_size_conv, emb_size_conv, True, weight_init), activation,
            Dense(emb_size_conv, emb_size_conv, True, weight_init), activation,
        )
        self.out_weight = Dense(emb_size_conv, emb_size, False, weight_init)

    # ------------------------------------------------------------------ reference operator signature
    @inference_only
    def forward(self, x: Tensor, cst: Tensor, valence_mask: Tensor | None, rb: Tensor, shb: Tensor, idx_s: Tensor,
                idx_t: Tensor, tri_idx_k: Tensor, edge_idx_ks: Tensor, edge_idx_st: Tensor) -> Tensor:
        """Literal operator: arbitrary cst / shb / triplet lists as in the reference (lcaonet.py:130-142).
        Materialises f_coeffs(cst) (E,D,V) and a (T,V) buffer like the reference does; LCAONet.forward uses
        `forward_fused`, which does not."""
        if self.add_valence and valence_mask is None:
            raise ValueError("valence_mask must be provided when add_valence=True")
        V, N, E = self.emb_size_conv, x.size(0), rb.size(0)
        nw = self.node_weight(x)
        xs, xk = nw[:, :V], nw[:, V:]
        cstf = run_mlp(self.f_coeffs, cst)
        w = ops.triplet_literal(cstf, rb, shb, xk, tri_idx_k, edge_idx_ks, valence_mask, V, self.add_valence)
        three_body_w = scatter(w, edge_idx_st, dim=0, dim_size=E)
        ft = run_mlp(self.f_three, three_body_w)
        l0 = torch.zeros(rb.size(1), dtype=torch.int32, device=rb.device)
        _, B, Bv = ops.coeff_contract(cstf, rb, None, l0, valence_mask, V, 1, self.add_valence)
        lw = ops.edge_combine(ft, B, Bv)
        xs = xs.contiguous()
        h = run_mlp(self.f_node, torch.cat([xs[idx_s], xs[idx_t]], dim=-1))
        msg = self.basis_weight(lw) * h
        return self.out_weight(scatter(msg, idx_s, dim=0, dim_size=N), residual=x)

    # ------------------------------------------------------------------ fused graph path
    # weight-derived caches (see LCAONet.invalidate_caches): dropped by everything that rewrites weights wholesale
    def train(self, mode: bool = True):
        drop_weight_caches(self)
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        drop_weight_caches(self)
        return super()._load_from_state_dict(*args, **kwargs)

    def _fusable(self, C: int) -> bool:
        V = self.emb_size_conv
        acts_silu = all(type(m) is nn.SiLU for m in (self.f_coeffs[1], self.f_coeffs[3], self.f_node[1], self.f_node[3]))
        return acts_silu and (C, V) in ops.FUSED_COEFF_SHAPES

    # node-level tables of the layer: gate = sigmoid(xk), Ps / Pt = the two halves of f_node.0 applied to xs
    def _folded_node_weights(self) -> tuple[Tensor, Tensor]:
        """[Wk; Wa Ws; Wb Ws] (3V, H) and [bk; Wa bs + b1; Wb bs] (3V) for lcao_node_tables — node_weight and the first
        f_node layer are both linear, so Ps = (x Ws^T + bs) Wa^T + b1 = x (Wa Ws)^T + (Wa bs + b1).  Weights only;
        cached until a parameter changes (data pointer or version counter)."""
        ps = (self.node_weight.weight, self.node_weight.bias, self.f_node[0].weight, self.f_node[0].bias)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        cached = self.__dict__.get("_fold")
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        V = self.emb_size_conv
        wn, bn_, w1, b1 = (p.detach() for p in ps)
        Ws, Wk, bs, bk = wn[:V], wn[V:], bn_[:V], bn_[V:]
        Wa, Wb = w1[:, :V].contiguous(), w1[:, V:].contiguous()
        WsT = Ws.t().contiguous()
        w_cat = torch.cat([Wk, ops.dense(WsT, Wa).t(), ops.dense(WsT, Wb).t()], dim=0).contiguous()
        b_cat = torch.cat([bk, ops.dense(bs.unsqueeze(0), Wa, b1)[0], ops.dense(bs.unsqueeze(0), Wb)[0]]).contiguous()
        if not torch.cuda.is_current_stream_capturing():   # tensors made inside a capture live in the graph's pool
            self.__dict__["_fold"] = (key, w_cat, b_cat)
        return w_cat, b_cat

    def transposed_weights(self) -> dict:
        """Contiguous transposes of this layer's weights as the reverse sweeps use them (model/forces.py); cached like the
        folded node weights until a parameter changes."""
        ps = (self.node_weight.weight, self.f_node[0].weight, self.f_node[2].weight, self.basis_weight.weight,
              self.f_three[0].weight, self.out_weight.weight)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        cached = self.__dict__.get("_wt")
        if cached is not None and cached[0] == key:
            return cached[1]
        V = self.emb_size_conv
        wn, w1, wn2, wb, w3, wout = (p.detach() for p in ps)
        t = lambda w: w.t().contiguous()
        wt = {"out": t(wout), "node2": t(wn2), "basis": t(wb), "three": t(w3), "node1_s": t(w1[:, :V]), "node1_t": t(w1[:, V:]),
              "nw_s": t(wn[:V]), "nw_k": t(wn[V:])}
        if not torch.cuda.is_current_stream_capturing():
            self.__dict__["_wt"] = (key, wt)
        return wt

    def _node_level(self, x: Tensor, gate: Tensor, Ps: Tensor | None, Pt: Tensor) -> None:
        """Fill gate / Ps / Pt (row blocks matching x's rows; Ps may be None) — one launch when the widths allow."""
        V, H = self.emb_size_conv, x.shape[1]
        if V % 32 == 0 and H % 16 == 0 and H <= 512:
            w_cat, b_cat = self._folded_node_weights()
            ops.node_tables(x, w_cat, b_cat, V, out=(gate, Ps, Pt))
            return
        wn, bn_ = self.node_weight.weight.detach(), self.node_weight.bias.detach()
        w1 = self.f_node[0].weight.detach()
        xs = ops.dense(x, wn[:V], bn_[:V])
        ops.dense(x, wn[V:], bn_[V:], ops.ACT_SIGMOID, out=gate)
        if Ps is not None:
            ops.dense(xs, w1[:, :V].contiguous(), self.f_node[0].bias.detach(), out=Ps)
        ops.dense(xs, w1[:, V:].contiguous(), out=Pt)

    @torch.no_grad()
    def forward_fused(self, x: Tensor, cst: Tensor, valence_ed: Tensor | None, rb: Tensor, uvec: Tensor,
                      ctx: EdgeContext, l_of_d: Tensor, n_l: int, fused: bool | None = None,
                      trace: dict | None = None, exact_norm: bool = False, n_own: int | None = None,
                      e_own: int | None = None, pair: tuple | None = None, precision: str = "fp32",
                      ghost_fn=None, cst_row: Tensor | None = None, cst16: Tensor | None = None) -> Tensor:
        """x (N,H), cst (E,D,C) and rb (E,D), uvec (E,3) in the caller's edge order; valence_ed (E,D) 0/1.

        cst16 (16-bit precisions): ops.lowp_pack_rows(cst, precision), packed once per forward by the caller.

        cst_row (int32, p-space; fused kernels only): cst is then a TABLE of coefficient rows and edge p reads row
        cst_row[p] — the gather is fused into lcao_coeff_mlp_contract and nothing of size (E, D, C) exists.

        n_own / e_own (sharded graphs, lcaonet_amd/shard.py): only the first n_own nodes are centres to update and
        only the first e_own edges of p-space carry messages; the remaining (ghost) nodes and their out-edges feed the
        three-body and two-body terms.  Returns the new state of the n_own centre nodes.

        ghost_fn (sharded graphs): x then holds the n_own OWNED rows only and ghost_fn() returns the ghost rows — it is
        called as late as possible: the node tables of the owned rows and the coefficient kernel over the owned-centre
        edges [0, e_own) are enqueued first, so a halo exchange still in flight overlaps with ~96 % of the layer's
        matrix work."""
        V = self.emb_size_conv
        N = ctx.n_nodes
        n_own = N if n_own is None else n_own
        e_own = ctx.n_edges if e_own is None else e_own
        E, dev = ctx.n_edges, x.device
        if fused is None:
            fused = self._fusable(V if pair is not None else cst.shape[-1])
        if cst_row is not None and (pair is not None or not fused or (precision in ops.LOWP and self.add_valence)):
            raise ValueError("forward_fused: cst_row needs the fused per-edge coefficient kernel")
        split = (ghost_fn is not None and fused and pair is None and precision == "fp32" and not exact_norm
                 and not self.add_valence and N > n_own)
        if ghost_fn is not None and not split:
            xg = ghost_fn()
            x = torch.cat([x, xg], dim=0) if xg.shape[0] else x
        rows = x.shape[0]            # n_own when the ghost rows are still in flight, else every local row
        new = lambda n: torch.empty(n, V, device=dev, dtype=torch.float32)
        gate, Pt, Ps = new(N), new(N), new(rows)
        self._node_level(x, gate[:rows], Ps, Pt[:rows])
        k2_out = None
        if split:
            k2_out = ops.coeff_mlp_buffers(E, V, n_l, dev, True, False)
            k2 = lambda rng: ops.coeff_mlp_contract(
                cst, rb, ctx.order, l_of_d, None, self.f_coeffs[0].weight.detach(), self.f_coeffs[2].weight.detach(),
                n_l, False, gate=gate, src_p=ctx.src_p, p_range=rng, out=k2_out, cst_row=cst_row)
            k2((0, e_own))                                           # owned centres: gate rows [0, n_own) only
            xg = ghost_fn()
            self._node_level(xg, gate[rows:], None, Pt[rows:])
            Ag, gram, B, Bv = k2((e_own, E))                         # ghost-centred in-edges of owned nodes
        out_ptr, in_ptr = ctx.out_ptr[: n_own + 1], ctx.in_ptr[: n_own + 1]
        src_own, dst_own = ctx.src_p[:e_own], ctx.dst_p[:e_own]
        # coefficient MLP + radial contraction (all local edges), then the three-body term (owned centres)
        if pair is not None:
            # cst here is the ((max_z+1)^2, D, C) species-pair table: f_coeffs runs on ~1.4 k x D rows, and the
            # per-edge work is a contraction of the cache-resident table with rb (no 2*D*(C*V+V*V) flop per edge)
            z, zdim = pair[0], pair[1]
            Tf = run_mlp(self.f_coeffs, cst)
            pc = lambda T: ops.pair_contract(T, rb, ctx.order, l_of_d, z, ctx.src_p, ctx.dst_p, gate, n_l, zdim)
            if self.add_valence:
                # lcaonet.py:174-183,195-203 on the table: the valence half is masked by the neighbour species, the
                # three-body term sees the sum of both halves, the two-body term keeps them apart (B, Bv)
                Tc, Tv = split_valence_table(Tf, pair[2], V)
                Ag, gram, _ = pc(Tc + Tv)
                B, Bv = pc(Tc)[2], pc(Tv)[2]
            else:
                Ag, gram, B = pc(Tf)
                Bv = None
            tbw = ops.triplet_gemm(Ag, gram, ctx.uvec_p(uvec), in_ptr, ctx.in_pos, out_ptr, e_own)
        elif fused and precision in ops.LOWP and not self.add_valence and cst.shape[-1] == V and rb.shape[1] <= ops.LOWP_MAX_ORBITALS:
            # BASELINE config 3: 16-bit MFMA on the per-edge coefficient GEMMs (bf16 split in two terms, or plain fp16)
            Ag, gram, B, Bv = ops.coeff_mlp_contract_bf16x3(
                cst, rb, ctx.order, l_of_d, self.f_coeffs[0].weight.detach(), self.f_coeffs[2].weight.detach(), n_l,
                gate=None if exact_norm else gate, src_p=ctx.src_p, cst_row=cst_row, precision=precision, cst16=cst16)
            if exact_norm:
                tbw = ops.triplet_sum(Ag, gate, uvec, ctx.order, ctx.src_p, in_ptr, ctx.in_pos, e_own)
            else:
                tbw = ops.triplet_gemm(Ag, gram, ctx.uvec_p(uvec), in_ptr, ctx.in_pos, out_ptr, e_own)
        elif fused:
            if not split:
                Ag, gram, B, Bv = ops.coeff_mlp_contract(
                    cst, rb, ctx.order, l_of_d, valence_ed, self.f_coeffs[0].weight.detach(),
                    self.f_coeffs[2].weight.detach(), n_l, self.add_valence,
                    gate=None if exact_norm else gate, src_p=ctx.src_p, cst_row=cst_row)
            if exact_norm:   # per-triplet V-long norm (validation of the Gram-matrix form)
                tbw = ops.triplet_sum(Ag, gate, uvec, ctx.order, ctx.src_p, in_ptr, ctx.in_pos, e_own)
            else:
                tbw = ops.triplet_gemm(Ag, gram, ctx.uvec_p(uvec), in_ptr, ctx.in_pos, out_ptr, e_own)
        else:
            if cst_row is not None:
                raise ValueError("forward_fused: cst_row needs the fused coefficient kernel (widths in ops.FUSED_COEFF_SHAPES)")
            cstf = run_mlp(self.f_coeffs, cst)
            A, B, Bv = ops.coeff_contract(cstf, rb, ctx.order, l_of_d, valence_ed, V, n_l, self.add_valence)
            tbw = ops.triplet_sum(A, gate, uvec, ctx.order, ctx.src_p, in_ptr, ctx.in_pos, e_own)
        if trace is not None:
            trace["three_body_w"] = tbw  # p-space
        B = B[:e_own]
        Bv = Bv[:e_own] if Bv is not None else None
        # two-body term and messages
        if fused and not self.add_valence:
            # basis weights, then the message rows summed per centre inside the kernel (msg (E, V) exists only as the scratch
            # of the 16-bit forms' fallback for uneven degrees); precision "f16": both in one launch, bw never written
            agg = ops.edge_two_body_agg(tbw, B, self.f_three[0].weight.detach(), self.basis_weight.weight.detach(), Ps, Pt,
                                        src_own, dst_own, out_ptr, self.f_node[2].weight.detach(),
                                        self.f_node[2].bias.detach(), ctx.column_hint(out_ptr, e_own), precision)
        else:
            ft = run_mlp(self.f_three, tbw)
            lw = ops.edge_combine(ft, B, Bv)
            bw = self.basis_weight(lw)
            code = act_code(self.f_node[1])
            if code == ops.ACT_SILU:
                h1 = ops.pair_silu(Ps, Pt, src_own, dst_own)
            else:
                h1 = self.f_node[1](Ps[src_own.long()] + Pt[dst_own.long()])
            code2 = act_code(self.f_node[3])
            h2 = self.f_node[2](h1, act=code2) if code2 is not None else self.f_node[3](self.f_node[2](h1))
            agg = ops.segment_sum(bw, out_ptr, h2)
        return self.out_weight(agg, residual=x[:n_own])


class LCAOOut(nn.Module):
    """Output head (reference lcaonet.py:219-319): node MLP, per-graph reduction, optional direct forces."""

    def __init__(self, emb_size: int, out_size: int, is_extensive: bool = True, regress_forces: bool = False,
                 direct_forces: bool = True, activation: nn.Module = nn.SiLU(),
                 weight_init: Callable[[Tensor], Tensor] | None = None):
        super().__init__()
        self.emb_size = emb_size
        self.out_size = out_size
        self.is_extensive = is_extensive
        self.regress_forces = regress_forces
        self.direct_forces = direct_forces
        self.out_lin = nn.Sequential(
            Dense(emb_size, emb_size, True, weight_init), activation,
            Dense(emb_size, emb_size // 2, True, weight_init), activation,
            Dense(emb_size // 2, out_size, False, weight_init),
        )
        if regress_forces and direct_forces:
            self.out_lin_force = nn.Sequential(
                Dense(2 * emb_size, emb_size, True, weight_init), activation,
                Dense(emb_size, emb_size // 2, True, weight_init), activation,
                Dense(emb_size // 2, 1, False, weight_init),
            )

    @inference_only
    def forward(self, x: Tensor, batch_idx: Tensor | None, idx_s: Tensor, idx_t: Tensor, edge_vec_st: Tensor,
                pos: Tensor, n_graphs: int | None = None, batch_plan=None):
        prop = run_mlp(self.out_lin, x)
        if batch_idx is not None:
            prop = scatter(prop, batch_idx, dim=0, dim_size=n_graphs, reduce="sum" if self.is_extensive else "mean",
                           plan=batch_plan if n_graphs is not None else None)
        else:
            prop = prop.sum(0, keepdim=True) if self.is_extensive else prop.mean(0, keepdim=True)
        if not self.regress_forces:
            return prop
        if self.direct_forces:
            x = x.contiguous()
            force_st = run_mlp(self.out_lin_force, torch.cat([x[idx_s], x[idx_t]], dim=-1)) * edge_vec_st
            return prop, scatter(force_st, idx_s, dim=0, dim_size=x.size(0))
        # F = -dE/dpos needs the whole model, not this head: LCAONet.forward routes it to model/forces.py
        raise NotImplementedError("LCAOOut alone cannot produce autograd forces; call LCAONet.forward")


class LCAONet(BaseMPNN):
    """Drop-in for lcaonet.model.LCAONet (reference lcaonet.py:322-540).  Also accepts the README spellings
    hidden_dim / coeffs_dim / conv_dim / out_dim."""

    def __init__(self, emb_size: int = 128, emb_size_coeff: int = 128, emb_size_conv: int = 128, out_size: int = 1,
                 n_interaction: int = 3, n_per_orb: int = 1, cutoff: float = 6.0,
                 rbf_type: str | type[BaseRadialBasis] = "hydrogen", cutoff_net: str | type[BaseCutoff] = "envelope",
                 max_z: int = 36, min_orb: str | None = None, max_orb: str | None = None, elec_to_node: bool = True,
                 add_valence: bool = False, extend_orb: bool = False, is_extensive: bool = True,
                 activation: str = "SiLU", weight_init: str | None = "glorotorthogonal",
                 atomref: Tensor | None = None, mean: Tensor | None = None, regress_forces: bool = False,
                 direct_forces: bool = True, hidden_dim: int | None = None, coeffs_dim: int | None = None,
                 conv_dim: int | None = None, out_dim: int | None = None):
        super().__init__()
        emb_size = hidden_dim if hidden_dim is not None else emb_size
        emb_size_coeff = coeffs_dim if coeffs_dim is not None else emb_size_coeff
        emb_size_conv = conv_dim if conv_dim is not None else emb_size_conv
        out_size = out_dim if out_dim is not None else out_size
        wi = init_resolver(weight_init) if weight_init is not None else None
        act = activation_resolver(activation)
        self.emb_size = emb_size
        self.emb_size_coeff = emb_size_coeff
        self.emb_size_conv = emb_size_conv
        self.out_size = out_size
        self.n_interaction = n_interaction
        self.cutoff = cutoff
        self.cutoff_net = cutoff_net
        self.elec_to_node = elec_to_node
        self.add_valence = add_valence
        self.regress_forces = regress_forces
        self.direct_forces = direct_forces

        elec_info = ElecInfo(max_z, max_orb, min_orb, n_per_orb)
        self.elec_info = elec_info
        cn = cutoffnet_resolver(cutoff_net, cutoff=cutoff)
        self.rbf = rbf_resolver(rbf_type, cutoff=cutoff, elec_info=elec_info, cutoff_net=cn)
        self.shbf = SphericalHarmonicsBasis(elec_info)
        self.n_l = int(elec_info.nl_list[:, 1].max()) + 1

        self.emb_layer = LCAOEmbedding(emb_size, emb_size_coeff, elec_info, max_z, elec_to_node, extend_orb, act, wi)
        if add_valence:
            self.valence_mask = ValenceMask(emb_size_conv, elec_info)
        self.int_layers = nn.ModuleList(
            [LCAOInteraction(emb_size, emb_size_coeff, emb_size_conv, add_valence, act, wi) for _ in range(n_interaction)])
        self.out_layer = LCAOOut(emb_size, out_size, is_extensive, regress_forces, direct_forces, act, wi)
        self.pp_layer = PostProcess(out_size, is_extensive, atomref, mean)

    # --- weight-derived caches ------------------------------------------------------------------------------
    # The eval-mode species / pair tables (LCAOEmbedding.species_tables) and the folded node weights
    # (LCAOInteraction._folded_node_weights) are functions of the weights only and are kept between forwards, keyed on
    # (data_ptr, _version) of every tensor they were built from.  Optimiser steps, load_state_dict and the package's
    # initialisers write in place on the parameter and bump the version; a write THROUGH `.data` (`p.data.copy_(ema)`)
    # does not, so everything that replaces weights wholesale drops the caches here — and code that edits `.data`
    # directly calls invalidate_caches().
    def checked_z(self, graph) -> Tensor:
        """The graph's atomic numbers range-checked and CLAMPED to [1, max_z] on the device (the reference indexes its
        embedding tables with z - 1 and raises IndexError, embed.py:41); a value outside sets the device error flag that
        forward() publishes.  Every entry point — inference, forces, training, sharded — reads z through this."""
        return ops.checked_index(graph[GraphKeys.Z], 1, self.emb_layer.max_z, ops.FLAG_ATOMIC_NUMBER)

    def invalidate_caches(self) -> None:
        """Forget the weight-derived tables; the next forward rebuilds them.  Call after writing weights through `.data`."""
        drop_weight_caches(self)

    def train(self, mode: bool = True):
        drop_weight_caches(self)
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        drop_weight_caches(self)
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):     # .to(device) / .float() / .half(): the tables live on the old device
        drop_weight_caches(self)
        return super()._apply(fn, *args, **kwargs)

    def reset_parameters(self) -> None:
        """Re-initialise every sub-module that has a reset_parameters() (the reference's modules initialise themselves
        in __init__; this is the same call repeated) and drop the weight-derived caches."""
        for m in self.modules():
            if m is not self and hasattr(m, "reset_parameters"):
                m.reset_parameters()
        drop_weight_caches(self)

    # --- index / geometry helpers with the reference's names -------------------------------------------
    @torch.no_grad()
    def get_triplets(self, graph):
        """Materialise the reference's triplet lists (lcaonet.py:439-486) into the graph: for each edge
        e=(s->t) in order, every edge e'=(k->s) ordered by k, except e itself.  The fused forward never needs
        them (it walks the in-edge CSR instead); this exists for the operator-level API and for tests."""
        idx_s, idx_t = graph[GraphKeys.Edge_idx]
        N = graph[GraphKeys.Z].size(0)
        E = idx_s.numel()
        order = torch.argsort(idx_t * N + idx_s, stable=True)          # in-edges grouped by target, by source
        ptr = torch.searchsorted(idx_t[order], torch.arange(N + 1, device=idx_s.device))
        n_in = (ptr[1:] - ptr[:-1])[idx_s]
        e_st = torch.repeat_interleave(torch.arange(E, device=idx_s.device), n_in)
        start = torch.cumsum(n_in, 0) - n_in
        offs = torch.arange(e_st.numel(), device=idx_s.device) - start[e_st]
        e_ks = order[ptr[idx_s[e_st]] + offs]
        keep = e_ks != e_st
        e_ks, e_st = e_ks[keep], e_st[keep]
        graph[GraphKeys.Idx_k_3b] = idx_s[e_ks]
        graph[GraphKeys.Edge_idx_ks_3b] = e_ks
        graph[GraphKeys.Edge_idx_st_3b] = e_st
        return graph

    @torch.no_grad()
    def calc_3body_angles(self, graph):
        vec = graph.get(GraphKeys.Edge_vec_st)
        if vec is None:
            raise ValueError("edge_vec_st is not calculated. Please run calc_atomic_distances(return_vec=True) first.")
        graph[GraphKeys.Angles_3b] = ops.triplet_angles(vec, graph[GraphKeys.Edge_idx_st_3b],
                                                        graph[GraphKeys.Edge_idx_ks_3b])
        return graph

    # --- one graph sharded over several GPUs (lcaonet_amd/shard.py, model/sharded.py) ------------------------------
    @torch.no_grad()
    def forward_sharded(self, plan, exchange_fn=None, reduce_fn=None, group=None, fused: bool | None = None,
                        reverse_fn=None):
        """This rank's part of forward() on a graph partitioned by centre node.  `plan` = shard.make_plans(...)[rank] or
        shard.plan_from_local(...).  Collectives: L-1 all-to-alls of boundary node rows (asynchronous, overlapped with
        the next layer's coefficient kernel; exchange_fn(x_own) -> ghost rows replaces them, synchronously) and one
        all-reduce of the (B, out) per-graph partial sums (reduce_fn).  Returns the global (B, out) result (identical
        on every rank) or ((B, out), forces of the owned nodes)."""
        from .. import shard
        from .sharded import ShardedForward

        # NOTE: this decorated method is the graph-less path; LCAONet.forward_sharded_train carries the autograd graph
        if self.training:
            raise NotImplementedError("forward_sharded is the inference path (eval mode); single-graph training: "
                                      "forward_sharded_train(plan, group)")
        dev = plan.graph[GraphKeys.Pos].device
        if dev.type == "cuda":
            ops.raise_pending_errors(dev)
        if self.regress_forces and not self.direct_forces:
            # F = -dE/dpos of the owned atoms: sharded reverse sweep with L reverse halo exchanges (model/forces.py)
            from .forces import run_sharded_forces

            res = run_sharded_forces(self, plan, group, exchange_fn, reduce_fn, reverse_fn)
            self._publish_index_errors(dev)
            return res
        reduce_fn = reduce_fn or (lambda part: shard.reduce_graph_outputs(part, group))
        sf = ShardedForward(self, plan, fused)
        pending = None
        for i in range(self.n_interaction):
            if i == 0:
                x_own = sf.layer(0, None)
            elif exchange_fn is not None:
                x_own = sf.layer(i, lambda x_prev=x_own: exchange_fn(x_prev))
            else:
                x_own = sf.layer(i, pending.wait)
            if i + 1 < self.n_interaction and exchange_fn is None:
                pending = shard.start_exchange(plan, x_own, group)
        out = sf.finish(reduce_fn(sf.head_partial()))
        if self.regress_forces:   # direct forces of the owned centres (their out-edges are all local)
            ghosts = exchange_fn(x_own) if exchange_fn is not None else shard.exchange_ghosts(plan, x_own, group)
            out = (out, sf.direct_forces(ghosts))
        self._publish_index_errors(dev)
        return out

    def forward_sharded_train(self, plan, group=None, **kw):
        """This rank's part of a DIFFERENTIABLE forward on one graph partitioned by centre node: returns the global
        (B, out) result with an autograd graph (train or eval mode; BatchNorm batch statistics of the whole graph through
        one all-reduce of species / pair counts).  After loss.backward() call shard.all_reduce_gradients(model, group).
        See model/double.py::forward_sharded_differentiable."""
        from .double import forward_sharded_differentiable

        dev = plan.graph[GraphKeys.Pos].device
        if dev.type == "cuda":
            ops.raise_pending_errors(dev)
        out = forward_sharded_differentiable(self, plan, group, **kw)
        self._publish_index_errors(dev)
        return out

    # --- forward ------------------------------------------------------------------------------------------
    def forward(self, graph, fused: bool | None = None, materialize_triplets: bool = False,
                trace: dict | None = None, exact_norm: bool = False, cst_mode: str = "edge",
                differentiable: bool | None = None, precision: str = "fp32", edge_context: EdgeContext | None = None):
        """graph: PyG-Batch-like object with z, pos, edge_index [centre, neighbour], edge_shift, lattice and
        optionally batch.  Returns (B, out_size) or ((B, out_size), (N, 3)) exactly like the reference.

        cst_mode: "edge" (default) — the general per-edge coefficient MLP: every (edge, orbital) row runs through both
        f_coeffs layers; in eval mode with the fused kernel the rows are gathered from the species-pair table inside the
        kernel, so the (E, D, C) tensor is never materialised.  "edge_tensor": the same with cst materialised per edge as
        the reference holds it.  "pair": f_coeffs evaluated on the pair table itself (fewer flops).

        precision: "fp32" (default; exact fp32 MFMA); BASELINE config 3 modes, the per-edge GEMMs on the 16-bit matrix
        path: "bf16x3" (bf16 MFMA with two-term operand splitting, three MFMAs per product: ~1e-5 from fp32, where plain
        bf16 would be 4e-3) or "f16" (operands rounded once to fp16, one MFMA per product: ~3e-4, inside the config's
        stated 1e-3).

        differentiable (default: autograd is enabled and some parameter requires grad — what the reference does in
        train AND eval mode, lcaonet.py:488-540): record an autograd graph to every parameter (model/train.py: HIP
        forward + hand-written HIP backward of the interaction stack; autograd forces then carry a graph as with the
        reference's create_graph=True).  Otherwise — under torch.no_grad(), with frozen parameters, with
        differentiable=False, or when an inference-only option (fused, trace, exact_norm, cst_mode, precision,
        edge_context, materialize_triplets) is given — the inference pipeline runs and the result carries no graph."""
        if differentiable is None:
            inference_only = (fused is not None or materialize_triplets or trace is not None or exact_norm
                              or cst_mode != "edge" or precision != "fp32" or edge_context is not None)
            differentiable = (torch.is_grad_enabled() and not inference_only
                              and any(p.requires_grad for p in self.parameters()))
        if precision not in ("fp32",) + ops.LOWP:
            raise ValueError(f"precision={precision!r}: expected 'fp32', 'bf16x3' or 'f16'")
        dev = graph[GraphKeys.Pos].device
        if dev.type == "cuda":
            ops.raise_pending_errors(dev)            # an earlier forward that has finished and flagged a bad index
        if differentiable:
            from .train import forward_differentiable

            out = forward_differentiable(self, graph)
        else:
            out = self._forward_inference(graph, fused, materialize_triplets, trace, exact_norm, cst_mode, precision,
                                          edge_context)
        self._publish_index_errors(dev)
        return out

    strict_indices: bool = False

    def _publish_index_errors(self, dev) -> None:
        """This call's index-error flag: an asynchronous copy to pinned memory, turned into LcaoHipError by the next call
        once it has landed — or, with `model.strict_indices = True`, before this call returns (one device sync)."""
        if dev.type != "cuda":
            return
        ops.publish_errors(dev)
        if self.strict_indices:
            ops.raise_pending_errors(dev, sync=True)

    @torch.no_grad()
    def pair_valence_mask(self, zdim: int) -> Tensor:
        """(zdim^2, D) 0/1: row z_s*zdim + z_t = valence orbitals of the NEIGHBOUR species z_t (embed.py:102-133 through
        ValenceMask.forward(z, idx_t))."""
        return self.valence_mask.valence[:zdim].to(torch.float32).repeat(zdim, 1).contiguous()

    @torch.no_grad()
    def geometry_and_basis(self, graph, ctx: EdgeContext):
        """(dist, uvec, rb): calc_atomic_distances (base.py:15-43; writes edge_dist / edge_vec into the graph as the
        reference does) and the radial basis — ONE launch for the built-in bases and cutoffs, which also leaves the unit
        vectors in p-space in the context."""
        kind = getattr(self.rbf.cutoff_net, "kind", None)
        rkind = getattr(self.rbf, "rbf_kind", None)
        if kind is None or rkind is None or type(self.rbf).forward is not BaseRadialBasis.forward:
            graph = BaseMPNN.calc_atomic_distances(graph, return_vec=True)      # user-defined basis / envelope
            dist, uvec = graph[GraphKeys.Edge_dist], graph[GraphKeys.Edge_vec_st]
            return dist, uvec, self.rbf(dist)
        ops._dev(graph[GraphKeys.Pos], "graph.pos")
        dist, uvec, uvec_p, rb = ops.edge_geometry_basis(
            graph[GraphKeys.Pos], ctx.order, ctx.src_p, ctx.dst_p, graph[GraphKeys.Edge_shift], graph[GraphKeys.Lattice],
            graph.get(GraphKeys.Batch_idx), self.rbf.orb_table, rkind, kind, self.cutoff)
        graph[GraphKeys.Edge_dist], graph[GraphKeys.Edge_vec_st] = dist, uvec
        ctx._uvec_p, ctx._uvec_src = uvec_p, uvec
        return dist, uvec, rb

    def _gathers_cst(self, fused: bool | None, precision: str) -> bool:
        """Eval mode with the fused coefficient kernel on every layer: the coefficient rows are gathered in-kernel."""
        if self.training or fused is False:
            return False
        if precision in ops.LOWP and (self.add_valence or self.emb_size_coeff != self.emb_size_conv):
            return False
        return all(layer._fusable(self.emb_size_coeff) for layer in self.int_layers)

    @torch.no_grad()
    def _forward_inference(self, graph, fused: bool | None = None, materialize_triplets: bool = False,
                           trace: dict | None = None, exact_norm: bool = False, cst_mode: str = "edge",
                           precision: str = "fp32", edge_context: EdgeContext | None = None):
        if self.regress_forces and not self.direct_forces:
            # F = -dE/dpos (lcaonet.py:310-317): species-pair forward + hand-written reverse sweep (model/forces.py)
            from .forces import forward_with_autograd_forces

            return forward_with_autograd_forces(self, graph, ctx=edge_context)
        batch_idx = graph.get(GraphKeys.Batch_idx)
        pos = graph[GraphKeys.Pos]
        edge_index = graph[GraphKeys.Edge_idx]
        idx_s, idx_t = edge_index[0], edge_index[1]
        ops._dev(pos, "graph.pos")
        # atomic numbers range-checked and clamped on the device (embed.py:41 indexes z - 1; the reference raises)
        z = self.checked_z(graph)
        N = z.size(0)

        # edge_context: the index structures of this edge list built earlier (EdgeContext.build) — they do not depend on
        # positions, so a caller that keeps the edge list (model/capture.py) builds them once
        ctx = edge_context if edge_context is not None else EdgeContext.build(edge_index, N)
        dist, uvec, rb = self.geometry_and_basis(graph, ctx)
        if materialize_triplets:
            graph = self.calc_3body_angles(self.get_triplets(graph))
        pair = cst_row = None
        if cst_mode == "pair":
            # cst_mode="pair": species-pair table instead of the per-edge (E, D, C) tensor.  Same
            # outputs; valid when cst comes from this model's own embedding in eval mode.
            if self.training:
                raise NotImplementedError("cst_mode='pair' needs eval mode (train mode: differentiable=True)")
            x, cst, zdim = self.emb_layer.forward_pair_table(z)
            pair = (z, zdim, self.pair_valence_mask(zdim) if self.add_valence else None)
        elif cst_mode not in ("edge", "edge_tensor"):
            raise ValueError(f"cst_mode={cst_mode!r}: expected 'edge', 'edge_tensor' or 'pair'")
        elif self.training:
            x, cst = self.emb_layer(z, idx_s, idx_t)  # BatchNorm batch statistics: module path
        elif cst_mode == "edge" and self._gathers_cst(fused, precision):
            # every (edge, orbital) row still goes through both f_coeffs layers, but the rows are read from the
            # species-pair table inside the kernel (cst[e] is a function of (z_s, z_t), embed.py:246-249): the (E, D, C)
            # tensor — 4.6 GB at the 1 M-edge config — is never written or re-read
            x, cst, zdim = self.emb_layer.forward_pair_table(z)
            cst_row = ops.pair_rows(z, ctx.src_p, ctx.dst_p, zdim)
        else:
            x, cst = self.emb_layer.forward_species(z, ctx)
        valence_ed = self.valence_mask.per_edge(z, idx_t) if self.add_valence else None
        # 16-bit coefficient kernels: the MFMA operands of cst, prepared once (cst is the same in every interaction)
        cst16 = None
        if (precision in ops.LOWP and pair is None and fused is not False and not self.add_valence
                and self.emb_size_coeff == self.emb_size_conv and self.elec_info.n_orb <= ops.LOWP_MAX_ORBITALS
                and all(layer._fusable(self.emb_size_coeff) for layer in self.int_layers)):
            cst16 = ops.lowp_pack_rows(cst, precision)

        if trace is not None:
            # cst: the per-edge tensor the reference holds (materialised for the trace only); pair table in cst_mode="pair"
            cst_tr = cst if cst_row is None else self.emb_layer.forward_species(z, ctx)[1]
            trace.update(rb=rb, x0=x, cst=cst_tr, order=ctx.order)
        for i, inte in enumerate(self.int_layers):
            tr = {} if trace is not None else None
            x = inte.forward_fused(x, cst, valence_ed, rb, uvec, ctx, self.shbf.l_of_d, self.n_l, fused, tr, exact_norm,
                                   pair=pair, precision=precision, cst_row=cst_row, cst16=cst16)
            if trace is not None:
                trace[f"x{i + 1}"] = x
                trace[f"three_body_w{i + 1}"] = tr["three_body_w"]

        n_graphs = num_graphs(graph, batch_idx)      # B = batch.max() + 1 as lcaonet.py:292, without a per-call sync
        plan = batch_plan(graph, batch_idx, n_graphs)   # one sort of the batch index per graph, shared by the three sums below
        out = self.out_layer(x, batch_idx, idx_s, idx_t, uvec, pos, n_graphs, batch_plan=plan)
        return self.pp_layer(out, z, batch_idx, n_graphs, batch_plan=plan)
