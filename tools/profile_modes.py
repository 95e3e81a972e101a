"""Per-kernel ms/step of the cfg2 forward in both cst modes (HIP events around every C-ABI call)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model
from lcaonet_amd import ops
from lcaonet_amd.data.synthetic import jittered_cubic_cells
dev = torch.device("cuda:0")
gen_kw, model_kw = WORKLOADS["cfg2"]
model = build_model(model_kw, 1234, dev)
batch = jittered_cubic_cells(seed=1234, device=dev, **gen_kw)
for mode in ("edge", "bf16x3", "pair"):
    kw = dict(precision="bf16x3") if mode == "bf16x3" else dict(cst_mode=mode)
    with torch.no_grad():
        for _ in range(3):
            model(batch, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with ops.profile() as prof:
            for _ in range(10):
                model(batch, **kw)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 10 * 1e3
            ks = prof.summary()
    tot = sum(v[1] for v in ks.values()) / 10
    print(f"mode={mode}: wall {wall:.2f} ms/step, kernels {tot:.2f} ms, glue {wall - tot:.2f} ms")
    for k, (n, ms) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:28s} calls/step={n/10:5.1f}  {ms/10:8.3f} ms/step  {ms/n:8.4f} ms/call")
