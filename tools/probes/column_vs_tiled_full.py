"""Full-size consistency of the two forms of the message kernel: the cfg2 batch (1.13 M edges) and a 40^3 crystal with 32
neighbours, forward in every precision with the per-node-column form (device hint) and with the hint forced to 0."""
import sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells, jittered_cubic_crystal
from lcaonet_amd.model.lcaonet import EdgeContext

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
orig = EdgeContext.column_hint
def forced_zero(self, out_ptr, n_edges):
    return torch.zeros(1, dtype=torch.int32, device=out_ptr.device)
for name, batch, mk in (("cfg2 1000 crystals", jittered_cubic_cells(seed=1234, device=dev, **gen_kw), model_kw),
                        ("40^3 crystal, 32 neighbours", jittered_cubic_crystal(40, cutoff_graph=5.94, max_neighbors=32, device=dev),
                         dict(model_kw, cutoff=6.0))):
    model = build_model(mk, 1234, dev)
    with torch.no_grad():
        for prec in ("fp32", "bf16x3", "f16"):
            kw = {} if prec == "fp32" else {"precision": prec}
            EdgeContext.column_hint = orig
            tr_c = {}
            a = model(batch, trace=tr_c, **kw).clone()
            EdgeContext.column_hint = forced_zero
            tr_t = {}
            b = model(batch, trace=tr_t, **kw).clone()
            EdgeContext.column_hint = orig
            ctx = EdgeContext.build(batch["edge_index"], batch["z"].numel())
            hint = int(ctx.column_hint(ctx.out_ptr, ctx.n_edges)[0])
            worst = max(float((tr_c[k] - tr_t[k]).abs().max() / tr_t[k].abs().max()) for k in tr_c
                        if isinstance(tr_c[k], torch.Tensor) and tr_c[k].dtype.is_floating_point)
            print(f"{name:30s} {prec:7s} hint {hint}  out rel diff {float((a - b).abs().max() / b.abs().max()):.2e}  "
                  f"worst traced tensor {worst:.2e}", flush=True)
