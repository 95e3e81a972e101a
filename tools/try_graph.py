"""Eager launches vs HIP-graph replay (lcaonet_amd.model.CapturedForward) of one whole forward, and of energy + forces.
usage: python tools/try_graph.py [n_cells]   (cfg2 cells of 64 atoms; default 8)"""
import sys, time
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model import CapturedForward

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
gen_kw = dict(gen_kw, n_cells=int(sys.argv[1]) if len(sys.argv) > 1 else 8)
batch = jittered_cubic_cells(seed=1234, device=dev, **gen_kw)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for label, kw in (("forward", {}), ("energy+forces", dict(regress_forces=True, direct_forces=False))):
    model = build_model(dict(model_kw, **kw), 1234, dev)
    with torch.no_grad():
        eager = timeit(lambda: model(batch))
        step = CapturedForward(model, batch)
        replay = timeit(lambda: step(batch["pos"]))
        ref, got = model(batch), step(batch["pos"])
        same = all(torch.equal(a, b) for a, b in zip(ref, got)) if isinstance(ref, tuple) else torch.equal(ref, got)
    print(f"n_cells {gen_kw['n_cells']:5d} {label:14s}: eager {eager:8.3f} ms   HIP-graph replay {replay:8.3f} ms   identical {same}")
