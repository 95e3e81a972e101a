"""Time the weight-gradient product dW = G^T X (G, X: E x 128) as torch.mm does it (rocBLAS) — the baseline for an own kernel."""
import torch, time
dev = torch.device("cuda:0")
E = 1128478
g = torch.randn(E, 128, device=dev); x = torch.randn(E, 128, device=dev); x2 = torch.randn(E, 256, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("mm(g.t(), x)          %.3f ms" % t(lambda: torch.mm(g.t(), x)))
print("mm(g.t(), x2[256])    %.3f ms" % t(lambda: torch.mm(g.t(), x2)))
print("g.sum(0)              %.3f ms" % t(lambda: g.sum(0)))
print("g.t().contiguous()    %.3f ms" % t(lambda: g.t().contiguous()))
N = 64000
gn = torch.randn(N, 256, device=dev); xn = torch.randn(N, 128, device=dev)
print("node-level mm         %.3f ms" % t(lambda: torch.mm(gn.t(), xn)))
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcaonet_amd import ops
ref = torch.mm(g.double().t(), x.double())
got = ops.wgrad(g, x)
print("lcao_wgrad 128x128    %.3f ms   rel err vs fp64 %.2e (torch.mm fp32: %.2e)" % (
    t(lambda: ops.wgrad(g, x)), float((got - ref).abs().max() / ref.abs().max()),
    float((torch.mm(g.t(), x) - ref).abs().max() / ref.abs().max())))
g2 = torch.randn(E, 256, device=dev)
ref2 = torch.mm(g2.double().t(), x.double())
print("lcao_wgrad 256x128    %.3f ms   rel err %.2e" % (t(lambda: ops.wgrad(g2, x)), float((ops.wgrad(g2, x) - ref2).abs().max() / ref2.abs().max())))
gs, xs = torch.randn(1000, 64, device=dev), torch.randn(1000, 192, device=dev)
print("small 64x192 err %.2e" % float((ops.wgrad(gs, xs) - gs.double().t() @ xs.double()).abs().max()))
print("deterministic", bool((ops.wgrad(g, x) == got).all()))
