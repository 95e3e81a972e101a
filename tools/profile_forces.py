"""cfg2-shaped batch with regress_forces=True, direct_forces=False: per-kernel ms of forward + reverse sweep."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model
from lcaonet_amd import ops
from lcaonet_amd.data.synthetic import jittered_cubic_cells
dev = torch.device("cuda:0")
gen_kw, model_kw = WORKLOADS["cfg2"]
model = build_model(dict(model_kw, regress_forces=True, direct_forces=False), 1234, dev)
batch = jittered_cubic_cells(seed=1234, device=dev, **gen_kw)
E = batch["edge_index"].shape[1]
with torch.no_grad():
    for _ in range(2):
        out, f = model(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with ops.profile() as prof:
        for _ in range(5):
            out, f = model(batch)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5 * 1e3
        ks = prof.summary()
print(f"energy+forces: wall {wall:.2f} ms/step = {E / wall * 1e3 / 1e6:.1f} M edges/s; force norm {float(f.norm()):.4f}, net force {f.sum(0).tolist()}")
for k, (n, ms) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k:28s} calls/step={n/5:5.1f}  {ms/5:8.3f} ms/step  {ms/n:8.4f} ms/call")
