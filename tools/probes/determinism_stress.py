"""Run the forward many times (eager and HIP-graph replay) and count runs that are not bitwise equal to the first."""
import sys
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model import CapturedForward

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for n in (8, 64, 250):
    batch = jittered_cubic_cells(seed=1234, device=dev, **dict(gen_kw, n_cells=n))
    model = build_model(model_kw, 1234, dev)
    with torch.no_grad():
        tr0 = {}
        ref = model(batch, trace=tr0).clone()
        bad = 0
        first_bad = None
        for i in range(reps):
            tr = {}
            out = model(batch, trace=tr)
            if not torch.equal(out, ref):
                bad += 1
                if first_bad is None:
                    first_bad = [k for k in tr0 if isinstance(tr0[k], torch.Tensor) and tr0[k].shape == tr[k].shape
                                 and not torch.equal(tr0[k], tr[k])]
        step = CapturedForward(model, batch)
        badg = sum(int(not torch.equal(step(batch["pos"]), ref)) for _ in range(reps))
        # interleave eager and replay like tools/try_graph.py does
        badi = 0
        for i in range(reps // 2):
            a = model(batch)
            b = step(batch["pos"])
            badi += int(not torch.equal(a, b))
    print(f"n_cells {n}: eager mismatches {bad}/{reps} (first differing traces: {first_bad}), replay mismatches {badg}/{reps}, "
          f"interleaved {badi}/{reps // 2}", flush=True)
