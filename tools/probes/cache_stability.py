import sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells
dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
batch = jittered_cubic_cells(seed=1234, device=dev, **dict(gen_kw, n_cells=8))
model = build_model(model_kw, 1234, dev)
def ptrs():
    out = {}
    for name, mod in model.named_modules():
        for k in ("_species", "_fold"):
            c = mod.__dict__.get(k)
            if c is not None:
                ts = c[1].values() if isinstance(c[1], dict) else c[1:]
                out[f"{name}.{k}"] = tuple(t.data_ptr() for t in ts if isinstance(t, torch.Tensor))
    return out
with torch.no_grad():
    model(batch); a = ptrs()
    for i in range(3):
        model(batch)
        b = ptrs()
        print(i, "caches stable:", a == b, {k: (a.get(k) == b.get(k)) for k in b})
        a = b
    keys = {n: (t.data_ptr(), t._version) for n, t in list(model.named_parameters()) + list(model.named_buffers())}
    model(batch)
    keys2 = {n: (t.data_ptr(), t._version) for n, t in list(model.named_parameters()) + list(model.named_buffers())}
    print("changed keys:", [n for n in keys if keys[n] != keys2[n]])
