"""Time lcao_triplet_gemm and lcao_triplet_bwd_mfma (cfg2 graph) for diagnostic builds in tools/abl/
(LCAO_ABL & 32: no weight arithmetic, & 64: one MFMA per step; & 128 / 256 / 512: backward GEMM 1 / scalar chain / GEMM 2 cut)."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcaonet_amd import _lib, ops
from lcaonet_amd.data.synthetic import jittered_cubic_cells
from lcaonet_amd.model.lcaonet import EdgeContext
from bench import WORKLOADS
dev = torch.device("cuda:0")
g = jittered_cubic_cells(seed=1234, device=dev, **WORKLOADS["cfg2"][0])
N, E = g["z"].numel(), g["edge_index"].shape[1]
ctx = EdgeContext.build(g["edge_index"], N)
V, NL = 128, 3
gen = torch.Generator(device=dev).manual_seed(0)
Ag = torch.randn(E, NL, V, device=dev, generator=gen)
gram = torch.einsum("elv,emv->elm", Ag.double(), Ag.double())
iu = torch.triu_indices(4, 4)
G = torch.zeros(E, 4, 4, device=dev); G[:, :NL, :NL] = torch.linalg.cholesky(gram).transpose(1, 2).float()   # R, R^T R = A A^T
gram10 = G[:, iu[0], iu[1]].contiguous()            # the per-edge triangular factor the kernels read
u = torch.nn.functional.normalize(torch.randn(E, 3, device=dev, generator=gen), dim=1)
tbw = torch.empty(E, V, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for path in sorted(glob.glob(sys.argv[1] + "/lib*.so")):
    lib = ctypes.CDLL(path)
    fn = lib.lcao_triplet_gemm
    fn.argtypes = _lib.SIGNATURES["lcao_triplet_gemm"]
    in_pos = ctx.in_pos if not os.environ.get("K3_CONTIG") else torch.arange(E, device=dev, dtype=torch.int32)  # in-order Ag layout
    run = lambda: fn(p(Ag), p(gram10), p(u), p(ctx.in_ptr), p(in_pos), p(ctx.out_ptr), N, E, V, NL, p(tbw), None)
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"{os.path.basename(path):12s} triplet_gemm {ms:7.3f} ms  {(4*NL*V+52+4*V)*E/ms/1e6:7.1f} GB/s", end="   ")
    fb = lib.lcao_triplet_bwd_mfma
    fb.argtypes = _lib.SIGNATURES["lcao_triplet_bwd_mfma"]
    g_tbw = torch.randn(E, V, device=dev, generator=gen)
    g_Ag = torch.empty_like(Ag); g_gram = torch.empty(E, 10, device=dev); g_uq = torch.empty(E, 3, device=dev); g_up = torch.empty(E, 3, device=dev)
    runb = lambda: fb(p(g_tbw), p(Ag), p(gram10), p(u), p(ctx.in_ptr), p(ctx.in_pos), p(ctx.out_ptr), N, E, V, NL, p(g_Ag), p(g_gram), p(g_uq), p(g_up), None)
    for _ in range(2): assert runb() == 0
    torch.cuda.synchronize()
    a.record()
    for _ in range(10): runb()
    b.record(); torch.cuda.synchronize()
    print(f"triplet_bwd_mfma {a.elapsed_time(b) / 10:7.3f} ms")
