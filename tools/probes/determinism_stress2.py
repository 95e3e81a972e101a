import sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model import CapturedForward

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
n = 250
batch = jittered_cubic_cells(seed=1234, device=dev, **dict(gen_kw, n_cells=n))
model = build_model(model_kw, 1234, dev)
with torch.no_grad():
    ref = model(batch).clone()
    step = CapturedForward(model, batch)
    bad_a = bad_b = 0
    mags = []
    for i in range(40):
        a = model(batch).clone()
        b = step(batch["pos"]).clone()
        ea, eb = not torch.equal(a, ref), not torch.equal(b, ref)
        bad_a += ea; bad_b += eb
        if ea or eb:
            mags.append((i, "eager" if ea else "", "replay" if eb else "", float((a - ref).abs().max()), float((b - ref).abs().max()),
                         int((a != ref).sum()), int((b != ref).sum())))
    print("interleaved without sync: eager bad", bad_a, "replay bad", bad_b, mags[:6])
    bad_a = bad_b = 0
    for i in range(40):
        a = model(batch).clone(); torch.cuda.synchronize()
        b = step(batch["pos"]).clone(); torch.cuda.synchronize()
        bad_a += not torch.equal(a, ref); bad_b += not torch.equal(b, ref)
    print("interleaved with sync between: eager bad", bad_a, "replay bad", bad_b)
    for mode in ("pair", "edge_tensor"):
        refm = model(batch, cst_mode=mode).clone()
        stepm = CapturedForward(model, batch, cst_mode=mode)
        bad = 0
        for i in range(40):
            a = model(batch, cst_mode=mode).clone()
            b = stepm(batch["pos"]).clone()
            bad += (not torch.equal(a, refm)) + (not torch.equal(b, refm))
        print("cst_mode", mode, "interleaved mismatches", bad)
