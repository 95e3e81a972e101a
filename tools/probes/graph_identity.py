"""Is a HIP-graph replay of the forward bitwise equal to the eager forward?  (and eager to itself)"""
import sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model import CapturedForward

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
for n in (8, 32, 64, 128):
    batch = jittered_cubic_cells(seed=1234, device=dev, **dict(gen_kw, n_cells=n))
    model = build_model(model_kw, 1234, dev)
    with torch.no_grad():
        a = model(batch).clone()
        b = model(batch).clone()
        step = CapturedForward(model, batch)
        c = step(batch["pos"]).clone()
        d = step(batch["pos"]).clone()
        tr1, tr2 = {}, {}
        model(batch, trace=tr1)
        e = model(batch, cst_mode="edge_tensor")
    print(n, "eager==eager", torch.equal(a, b), "replay==replay", torch.equal(c, d), "eager==replay", torch.equal(a, c),
          "max|diff|/max", float((a - c).abs().max() / a.abs().max()), "edge_tensor==eager", torch.equal(a, e))
