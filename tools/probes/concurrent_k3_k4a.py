"""Can the HBM-bound three-body kernel (lcao_triplet_gemm) and the fp32-MFMA-bound basis-weight kernel (lcao_edge_lcao_weight)
share the chip?  Both at config-2 size on independent data: each alone, back to back on one stream, and concurrently on two
streams.  If concurrent ~ max(alone) a half-and-half split of a layer's edges could overlap them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import WORKLOADS
from lcaonet_amd import ops
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model.lcaonet import EdgeContext

dev = torch.device("cuda:0")
g = jittered_cubic_cells(seed=1234, device=dev, **WORKLOADS["cfg2"][0])
N, E = g["z"].numel(), g["edge_index"].shape[1]
ctx = EdgeContext.build(g["edge_index"], N)
V, NL = 128, 3
gen = torch.Generator(device=dev).manual_seed(0)
Ag = torch.randn(E, NL, V, device=dev, generator=gen)
fac = torch.zeros(E, 10, device=dev); fac[:, [0, 4, 7]] = 1.0
u = torch.nn.functional.normalize(torch.randn(E, 3, device=dev, generator=gen), dim=1)
tbw_in = torch.randn(E, V, device=dev, generator=gen); B = torch.randn(E, V, device=dev, generator=gen)
W3 = torch.randn(V, V, device=dev, generator=gen) / V**0.5; Wb = torch.randn(V, V, device=dev, generator=gen) / V**0.5
k3 = lambda: ops.triplet_gemm(Ag, fac, u, ctx.in_ptr, ctx.in_pos, ctx.out_ptr)
k4 = lambda: ops.edge_lcao_weight(tbw_in, B, W3, Wb)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

def both_streams():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa): k3()
    with torch.cuda.stream(sb): k4()
    cur.wait_stream(sa); cur.wait_stream(sb)

t3, t4 = timeit(k3), timeit(k4)
tseq = timeit(lambda: (k3(), k4()))
tcon = timeit(both_streams)
print(f"triplet_gemm alone {t3:.3f} ms | edge_lcao_weight alone {t4:.3f} ms | back to back {tseq:.3f} | two streams {tcon:.3f} (sum {t3 + t4:.3f}, max {max(t3, t4):.3f})")
for r in (32, 64, 96, 128):
    os.environ["LCAO_CU_RESERVE"] = str(r)
print("(persistent grids fill every CU: see LCAO_CU_RESERVE for leaving room)")
