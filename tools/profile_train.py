"""cfg2-shaped batch, one training step (forward + loss.backward()): wall time and per-kernel ms of the HIP calls."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model
from lcaonet_amd import ops
from lcaonet_amd.data.synthetic import jittered_cubic_cells
dev = torch.device("cuda:0")
gen_kw, model_kw = WORKLOADS["cfg2"]
model = build_model(model_kw, 1234, dev).train()
batch = jittered_cubic_cells(seed=1234, device=dev, **gen_kw)
E = batch["edge_index"].shape[1]
target = torch.randn(1000, 1, device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.mse_loss(model(batch), target)
    loss.backward()
    opt.step()
    return loss
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
with ops.profile() as prof:
    for _ in range(5):
        loss = step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    ks = prof.summary()
tot = sum(v[1] for v in ks.values()) / 5
print(f"training step (fwd + bwd + Adam): wall {wall:.2f} ms = {E / wall * 1e3 / 1e6:.1f} M edges/s; HIP entry points {tot:.2f} ms, "
      f"torch (weight-grad GEMMs, table autograd, optimiser) {wall - tot:.2f} ms; loss {float(loss.detach()):.4f}")
for k, (n, ms) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"   {k:28s} calls/step={n/5:5.1f}  {ms/5:8.3f} ms/step  {ms/n:8.4f} ms/call")
