"""Time lcao_edge_node_message_agg (cfg2 graph) for diagnostic builds in tools/abl/
(LCAO_ABL & 1024: no segmented reduction, & 2048: identity activations, & 4096: no random gather of target rows)."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcaonet_amd import _lib
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model.lcaonet import EdgeContext
from bench import WORKLOADS
dev = torch.device("cuda:0")
g = jittered_cubic_cells(seed=1234, device=dev, **WORKLOADS["cfg2"][0])
N, E = g["z"].numel(), g["edge_index"].shape[1]
ctx = EdgeContext.build(g["edge_index"], N)
V = 128
gen = torch.Generator(device=dev).manual_seed(0)
bw = torch.randn(E, V, device=dev, generator=gen)
Ps, Pt = torch.randn(N, V, device=dev, generator=gen), torch.randn(N, V, device=dev, generator=gen)
Wn2 = torch.randn(V, V, device=dev, generator=gen) / V**0.5
bn2 = torch.randn(V, device=dev, generator=gen)
agg = torch.empty(N, V, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for path in sorted(glob.glob(sys.argv[1] + "/lib*.so")):
    lib = ctypes.CDLL(path)
    fn = lib.lcao_edge_node_message_agg
    fn.argtypes = _lib.SIGNATURES["lcao_edge_node_message_agg"]
    hint = torch.full((1,), int(os.environ.get("K4B_COLS", "0")), dtype=torch.int32, device=dev)   # 1: per-node-column form
    run = lambda: fn(p(bw), p(Ps), p(Pt), p(ctx.src_p), p(ctx.dst_p), p(ctx.out_ptr), p(Wn2), p(bn2), N, E, V, p(hint), p(agg), None)
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"{os.path.basename(path):14s} edge_node_message_agg {ms:7.3f} ms  {2*V*V*E/ms/1e9:7.1f} TFLOP/s")
