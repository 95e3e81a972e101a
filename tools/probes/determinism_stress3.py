"""Interleave eager forwards and HIP-graph replays; stop at the first replay that differs.  argv: cst_mode [n_cells]"""
import sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model import CapturedForward

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250
kw = {} if mode == "default" else ({"fused": False} if mode == "unfused" else {"cst_mode": mode})
batch = jittered_cubic_cells(seed=1234, device=dev, **dict(gen_kw, n_cells=n))
model = build_model(model_kw, 1234, dev)
with torch.no_grad():
    ref = model(batch, **kw).clone()
    step = CapturedForward(model, batch, **kw)
    for i in range(60):
        a = model(batch, **kw).clone()
        b = step(batch["pos"]).clone()
        ea, eb = not torch.equal(a, ref), not torch.equal(b, ref)
        if ea or eb:
            print(mode, "first mismatch at", i, "eager" if ea else "", "replay" if eb else "", flush=True)
            break
    else:
        print(mode, "no mismatch in 60 interleaved iterations", flush=True)
