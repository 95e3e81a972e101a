"""Time lcao_coeff_mlp_contract (config-2 shapes) for ablated diagnostic builds of the library.
Build variants:  for a in 0 1 2 4 7; do hipcc ... -DLCAO_ABL=$a -o tools/abl/lib$a.so; done  (see tools/build_ablate.sh)"""
import ctypes, glob, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcaonet_amd import _lib

dev = torch.device("cuda:0")
E, D, C, V, NL, N = 1128478, 8, 128, 128, 3, 64000
g = torch.Generator(device=dev).manual_seed(0)
cst = torch.randn(E, D, C, device=dev, generator=g)
rb = torch.randn(E, D, device=dev, generator=g)
W1 = torch.randn(V, C, device=dev, generator=g) / C**0.5
W2 = torch.randn(V, V, device=dev, generator=g) / V**0.5
gate = torch.rand(N, V, device=dev, generator=g)
src = torch.sort(torch.randint(0, N, (E,), device=dev, generator=g)).values.to(torch.int32)
order = torch.arange(E, device=dev, dtype=torch.int32)
l_of_d = torch.tensor([0, 0, 1, 0, 1, 0, 2, 1], dtype=torch.int32, device=dev)
B = torch.empty(E, V, device=dev); Ag = torch.empty(E, NL, V, device=dev); gram = torch.empty(E, 10, device=dev)
ws = torch.zeros(4, dtype=torch.int32, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for path in sorted(glob.glob(sys.argv[1] + "/lib*.so")):
    lib = ctypes.CDLL(path)
    fn = lib.lcao_coeff_mlp_contract
    fn.argtypes = _lib.SIGNATURES["lcao_coeff_mlp_contract"]
    def run():
        rc = fn(p(cst), p(rb), p(order), None, p(l_of_d), None, p(W1), p(W2), p(gate), p(src), E, D, C, V, NL, 0, None, p(B), None,
                p(Ag), p(gram), p(ws), 16, None)
        assert rc == 0, rc
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{os.path.basename(path):12s} {ms:7.3f} ms  {2*D*(C*V+V*V)*E/ms/1e9:7.1f} TFLOP/s")
