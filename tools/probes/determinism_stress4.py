"""Interleave eager forwards and HIP-graph replays (default cst mode); at the first differing replay report which traced
stage differs first, then stop."""
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
    tr0 = {}
    ref = model(batch, trace=tr0).clone()
    tr0 = {k: v.clone() for k, v in tr0.items() if isinstance(v, torch.Tensor)}
    trg = {}
    step = CapturedForward(model, batch, trace=trg)
    print("allocated GiB", torch.cuda.memory_allocated() / 2**30, "reserved", torch.cuda.memory_reserved() / 2**30, flush=True)
    for i in range(45):
        a = model(batch).clone()
        b = step(batch["pos"]).clone()
        if not torch.equal(b, ref) or not torch.equal(a, ref):
            torch.cuda.synchronize()
            diff = [(k, int((trg[k] != tr0[k]).sum()), trg[k].numel()) for k in tr0 if k in trg and trg[k].shape == tr0[k].shape
                    and not torch.equal(trg[k], tr0[k])]
            print("first mismatch at", i, "eager ok" if torch.equal(a, ref) else "EAGER BAD", "differing traces:", diff, flush=True)
            print("reserved", torch.cuda.memory_reserved() / 2**30, flush=True)
            break
    else:
        print("no mismatch", flush=True)
