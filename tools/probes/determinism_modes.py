"""Bitwise run-to-run stability of every mode on one batch (n_cells crystals of cfg2 shape)."""
import sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
batch = jittered_cubic_cells(seed=1234, device=dev, **dict(gen_kw, n_cells=n))
N = batch["z"].numel()

def flat(o):
    return torch.cat([t.reshape(-1).float() for t in (o if isinstance(o, (tuple, list)) else [o])])

def stable(fn, reps=10):
    ref = flat(fn()).clone()
    return sum(int(not torch.equal(flat(fn()), ref)) for _ in range(reps))

model = build_model(model_kw, 1234, dev)
with torch.no_grad():
    for name, kw in (("edge", {}), ("edge_tensor", {"cst_mode": "edge_tensor"}), ("pair", {"cst_mode": "pair"}),
                     ("bf16x3", {"precision": "bf16x3"}), ("f16", {"precision": "f16"}), ("unfused", {"fused": False})):
        print(f"forward {name:12s} mismatching runs: {stable(lambda: model(batch, **kw))}/10", flush=True)
    fm = build_model(dict(model_kw, regress_forces=True, direct_forces=False), 1234, dev)
    print(f"energy+forces (sweep)  mismatching runs: {stable(lambda: fm(batch))}/10", flush=True)

def grads(m, forces):
    m.zero_grad(set_to_none=True)
    out = m(batch)
    loss = (out[0].abs().sum() + out[1].abs().sum()) if forces else out.abs().sum()
    loss.backward()
    return [p.grad for p in m.parameters() if p.grad is not None]

tm = build_model(model_kw, 1234, dev).train()
print(f"training gradients     mismatching runs: {stable(lambda: grads(tm, False), 5)}/5", flush=True)
ft = build_model(dict(model_kw, regress_forces=True, direct_forces=False), 1234, dev).train()
print(f"force-training grads   mismatching runs: {stable(lambda: grads(ft, True), 3)}/3", flush=True)
