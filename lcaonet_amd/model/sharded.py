"""One rank's share of LCAONet.forward on a graph partitioned by centre node (lcaonet_amd/shard.py).

`ShardedForward` holds what does not change between layers (index structures, geometry, radial basis, coefficient
vectors of every local edge — functions of the static inputs z and pos, so ghosts need no exchange for them) and steps
through the interaction layers one at a time.  The caller owns the communication: LCAONet.forward_sharded drives it over
torch.distributed (RCCL) with the exchange of layer i's output in flight under layer i+1's owned-edge coefficient
kernel; the GPU tests drive several ranks in lock step on one device with shard.local_exchange."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import ops
from ..data.keys import GraphKeys
from ..nn.base import run_mlp
from ..nn.scatter import scatter


class ShardedForward:
    @torch.no_grad()
    def __init__(self, model, plan, fused: bool | None = None):
        from .lcaonet import EdgeContext

        self.model, self.plan, self.fused = model, plan, fused
        graph = plan.graph
        z = model.checked_z(graph)
        self.z = z
        self.ctx = EdgeContext.build(graph[GraphKeys.Edge_idx], z.size(0))
        _, self.uvec, self.rb = model.geometry_and_basis(graph, self.ctx)
        # static inputs: every local row (ghosts included) is known without an exchange
        self.cst_row = None
        if model._gathers_cst(fused, "fp32"):
            self.x0, self.cst, zdim = model.emb_layer.forward_pair_table(z)
            self.cst_row = ops.pair_rows(z, self.ctx.src_p, self.ctx.dst_p, zdim)
        else:
            self.x0, self.cst = model.emb_layer.forward_species(z, self.ctx)
        idx_t = graph[GraphKeys.Edge_idx][1]
        self.valence_ed = model.valence_mask.per_edge(z, idx_t) if model.add_valence else None
        self.x_own = None

    @torch.no_grad()
    def layer(self, i: int, ghost_fn) -> Tensor:
        """Interaction layer i.  ghost_fn() -> the (n_ghost, H) ghost rows of this layer's input (None for layer 0,
        whose input is a function of z).  Returns (and keeps) the new state of the owned nodes."""
        m, p = self.model, self.plan
        if i == 0:
            x, gf = self.x0, None
        else:
            x, gf = self.x_own, ghost_fn
        self.x_own = m.int_layers[i].forward_fused(x, self.cst, self.valence_ed, self.rb, self.uvec, self.ctx,
                                                   m.shbf.l_of_d, m.n_l, self.fused, None, False, p.n_own, p.e_own,
                                                   ghost_fn=gf, cst_row=self.cst_row)
        return self.x_own

    @torch.no_grad()
    def head_partial(self) -> Tensor:
        """(B, out + 1 [+ out]) partial sums over the OWNED nodes: property, node count[, atomref]."""
        m, p = self.model, self.plan
        batch_own = p.graph[GraphKeys.Batch_idx][: p.n_own]
        prop = run_mlp(m.out_layer.out_lin, self.x_own)
        parts = [scatter(prop, batch_own, dim=0, dim_size=p.n_graphs, reduce="sum"),
                 scatter(torch.ones(p.n_own, 1, device=prop.device), batch_own, dim=0, dim_size=p.n_graphs, reduce="sum")]
        if m.pp_layer.atomref is not None:
            parts.append(scatter(m.pp_layer.atomref[self.z[: p.n_own]].float().contiguous(), batch_own, dim=0,
                                 dim_size=p.n_graphs, reduce="sum"))
        return torch.cat(parts, dim=1)

    @torch.no_grad()
    def finish(self, tot: Tensor) -> Tensor:
        """The all-reduced partial sums -> (B, out) result (LCAOOut reduction + PostProcess, lcaonet.py:290-298)."""
        return finish_outputs(self.model, tot)

    @torch.no_grad()
    def direct_forces(self, ghosts: Tensor) -> Tensor:
        """Direct-force head on the owned centres (lcaonet.py:303-308); `ghosts` = the ghost rows of the final state."""
        m, p = self.model, self.plan
        idx_s, idx_t = p.graph[GraphKeys.Edge_idx]
        own_e = idx_s < p.n_own
        es, et = idx_s[own_e], idx_t[own_e]
        xl = torch.cat([self.x_own, ghosts], dim=0) if p.n_ghost else self.x_own
        f = run_mlp(m.out_layer.out_lin_force, torch.cat([xl[es], xl[et]], dim=-1)) * self.uvec[own_e]
        return scatter(f, es, dim=0, dim_size=p.n_own)


def finish_outputs(model, tot: Tensor) -> Tensor:
    osz = model.out_size
    out, cnt = tot[:, :osz], tot[:, osz: osz + 1]
    if not model.out_layer.is_extensive:
        out = out / cnt.clamp(min=1)
    pp = model.pp_layer
    if pp.atomref is not None:
        aref = tot[:, osz + 1:]
        out = out + (aref if pp.is_extensive else aref / cnt.clamp(min=1))
    if pp.mean is not None:
        out = out + (pp.mean.unsqueeze(0) * cnt if pp.is_extensive else pp.mean.unsqueeze(0))
    return out
