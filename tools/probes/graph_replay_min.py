"""Minimal: capture the forward at a size with dynamic K2 tiles, replay a few times, compare with eager.  Stops at the
first mismatch."""
import sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model import CapturedForward

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
batch = jittered_cubic_cells(seed=1234, device=dev, **dict(gen_kw, n_cells=n))
model = build_model(model_kw, 1234, dev)
with torch.no_grad():
    ref = model(batch).clone()
    step = CapturedForward(model, batch)
    for i in range(6):
        b = step(batch["pos"]).clone()
        torch.cuda.synchronize()
        ok = torch.equal(b, ref)
        print(f"replay {i}: ok {ok}", flush=True)
        if not ok:
            break
        x = torch.empty((i + 1) * 2**28, dtype=torch.uint8, device=dev)   # new allocator segments between replays
        x.zero_()
        a = model(batch)
        if not torch.equal(a, ref):
            print("eager differs", flush=True)
            break
print("done", flush=True)
