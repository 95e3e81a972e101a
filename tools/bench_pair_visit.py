"""lcao_pair_contract on the cfg2 graph: p-space order vs edges walked sorted by species pair."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS
from lcaonet_amd import ops
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model.lcaonet import EdgeContext
dev = torch.device("cuda:0")
g = jittered_cubic_cells(seed=1234, device=dev, **WORKLOADS["cfg2"][0])
z = g["z"]; N, E = z.numel(), g["edge_index"].shape[1]
ctx = EdgeContext.build(g["edge_index"], N)
D, V, NL, zd = 8, 128, 3, 37
gen = torch.Generator(device=dev).manual_seed(0)
Tf = torch.randn(zd * zd, D, V, device=dev, generator=gen)
rb = torch.randn(E, D, device=dev, generator=gen)
gate = torch.rand(N, V, device=dev, generator=gen)
l_of_d = torch.tensor([0, 0, 1, 0, 1, 0, 2, 1], dtype=torch.int32, device=dev)
pair_p = z[ctx.src_p.long()] * zd + z[ctx.dst_p.long()]
visit = torch.argsort(pair_p, stable=True).to(torch.int32)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a = ops.pair_contract(Tf, rb, ctx.order, l_of_d, z, ctx.src_p, ctx.dst_p, gate, NL, zd)
b = ops.pair_contract(Tf, rb, ctx.order, l_of_d, z, ctx.src_p, ctx.dst_p, gate, NL, zd, visit=visit)
print("identical:", all(bool((x == y).all()) for x, y in zip(a, b)))
print("p order      %.3f ms" % t(lambda: ops.pair_contract(Tf, rb, ctx.order, l_of_d, z, ctx.src_p, ctx.dst_p, gate, NL, zd)))
print("pair-sorted  %.3f ms" % t(lambda: ops.pair_contract(Tf, rb, ctx.order, l_of_d, z, ctx.src_p, ctx.dst_p, gate, NL, zd, visit=visit)))
print("argsort      %.3f ms" % t(lambda: torch.argsort(pair_p, stable=True)))
