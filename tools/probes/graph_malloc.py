"""Does a device allocation (new allocator segment) between replays change what a captured forward computes?"""
import sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model import CapturedForward

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
batch = jittered_cubic_cells(seed=1234, device=dev, **dict(gen_kw, n_cells=250))
model = build_model(model_kw, 1234, dev)
with torch.no_grad():
    ref = model(batch).clone()
    step = CapturedForward(model, batch)
    def check(tag):
        b = step(batch["pos"]).clone()
        torch.cuda.synchronize()
        print(f"{tag:46s} replay ok: {torch.equal(b, ref)}   reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
        return torch.equal(b, ref)
    ok = check("after capture")
    for i in range(3):
        ok = ok and check(f"replay {i}")
    x = torch.empty(3 * 2**30, dtype=torch.uint8, device=dev)
    ok = ok and check("after a new 3 GiB segment (uninitialised)")
    if ok:
        x.zero_()
        ok = check("after writing that segment")
    if ok:
        del x
        torch.cuda.empty_cache()
        ok = check("after empty_cache")
    if ok:
        for i in range(3):
            a = model(batch).clone()
            ok = ok and check(f"after eager forward {i} (eager ok: {torch.equal(a, ref)})")
