"""Capture one whole LCAONet forward in a HIP graph (torch.cuda.CUDAGraph) and compare replay time with eager launches."""
import sys, time
sys.path.insert(0, ".")
import torch
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells

dev = torch.device("cuda", 0)
gen_kw, model_kw = WORKLOADS["cfg2"]
if len(sys.argv) > 1:
    gen_kw = dict(gen_kw, n_cells=int(sys.argv[1]))
model = build_model(model_kw, 1234, dev)
batch = jittered_cubic_cells(seed=1234, device=dev, **gen_kw)
with torch.no_grad():
    for _ in range(3):
        ref = model(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        model(batch)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20 * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model(batch)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = model(batch)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 20 * 1e3
print(f"n_cells {gen_kw['n_cells']}: eager {eager:.3f} ms  graph {graph:.3f} ms  max diff {float((out - ref).abs().max()):.3e}")
