"""Long mixed stress of the tile queue: several captured graphs (different sizes, cst modes, precisions) replayed in random
order, interleaved with eager forwards of other sizes and allocator churn; every result compared bitwise with its reference."""
import random, sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model import CapturedForward

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
model = build_model(model_kw, 1234, dev)
cases = []
with torch.no_grad():
    for n, kw in ((40, {}), (250, {}), (120, {"cst_mode": "edge_tensor"}), (90, {"precision": "f16"}), (64, {"precision": "bf16x3"}),
                  (300, {"cst_mode": "pair"})):
        batch = jittered_cubic_cells(seed=100 + n, device=dev, **dict(gen_kw, n_cells=n))
        ref = model(batch, **kw).clone()
        cases.append((batch, kw, ref, CapturedForward(model, batch, **kw)))
    rng = random.Random(0)
    bad = 0
    keep = []
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    for i in range(iters):
        batch, kw, ref, step = rng.choice(cases)
        if rng.random() < 0.5:
            out = step(batch["pos"])
        else:
            out = model(batch, **kw)
        if not torch.equal(out, ref):
            bad += 1
            print("mismatch at", i, kw, flush=True)
            if bad > 3:
                break
        if i % 37 == 0:
            keep.append(torch.zeros(rng.randrange(1, 64) * 2**20, dtype=torch.uint8, device=dev))
            if len(keep) > 6:
                keep.pop(0)
    print(f"{iters} mixed iterations, mismatches: {bad}", flush=True)
