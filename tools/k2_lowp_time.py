"""Time lcao_coeff_mlp_contract_f16 / _bf16x3 at config-2 shapes (coefficient rows gathered from a species-pair table,
as the product does) for every tools/abl/lowp_*.so variant, and check each against an fp32 torch evaluation.
Build variants with tools/build_lowp_variants.sh."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcaonet_amd import _lib

dev = torch.device("cuda:0")
E, D, C, V, NL, N, Z = 1128478, 8, 128, 128, 3, 64000, 37
g = torch.Generator(device=dev).manual_seed(0)
table = torch.randn(Z * Z, D, C, device=dev, generator=g)
rows = torch.randint(0, Z * Z, (E,), device=dev, generator=g, dtype=torch.int32)
rb = torch.randn(E, D, device=dev, generator=g)
W1 = torch.randn(V, C, device=dev, generator=g) / C**0.5
W2 = torch.randn(V, V, device=dev, generator=g) / V**0.5
gate = torch.rand(N, V, device=dev, generator=g)
src = torch.sort(torch.randint(0, N, (E,), device=dev, generator=g)).values.to(torch.int32)
l_list = [0, 0, 1, 0, 1, 0, 2, 1]
l_of_d = torch.tensor(l_list, dtype=torch.int32, device=dev)
B = torch.empty(E, V, device=dev); Ag = torch.empty(E, NL, V, device=dev); gram = torch.empty(E, 10, device=dev)
ws = torch.zeros(4, dtype=torch.int32, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())

# fp32 reference on a sample of edges
ns = 4096
xs = table[rows[:ns].long()]                                   # (ns, D, C)
f = torch.nn.functional.silu(torch.nn.functional.silu(xs @ W1.T) @ W2.T) * rb[:ns, :, None]
Aref = torch.stack([f[:, [d for d in range(D) if l_list[d] == l]].sum(1) for l in range(NL)], 1)   # (ns, NL, V)
Bref = Aref.sum(1)
Agref = Aref * gate[src[:ns].long()][:, None, :]

args = [a for a in sys.argv[1:] if not a.startswith("--")]
pat = args[0] if args else "tools/abl/lowp_*.so"
PH = ["staging", "loop top", "act1+prefetch", "layer 2", "act2+acc", "layer 1", "A stores", "gate req + B", "Gram", "-"]
for path in sorted(glob.glob(pat)):
    lib = ctypes.CDLL(path)
    for name in ("lcao_coeff_mlp_contract_f16", "lcao_coeff_mlp_contract_bf16x3"):
        fn = getattr(lib, name)
        fn.argtypes = _lib.SIGNATURES[name]
        if hasattr(lib, "lcao_lowp_pack_rows"):     # round-3 ABI: packed 16-bit rows
            code = 1 if name.endswith("f16") else 0
            lib.lcao_lowp_packed_bytes.restype = ctypes.c_size_t
            lib.lcao_lowp_packed_bytes.argtypes = _lib.SIGNATURES["lcao_lowp_packed_bytes"]
            lib.lcao_lowp_pack_rows.argtypes = _lib.SIGNATURES["lcao_lowp_pack_rows"]
            src_rows = torch.empty(lib.lcao_lowp_packed_bytes(table.numel(), code), dtype=torch.uint8, device=dev)
            assert lib.lcao_lowp_pack_rows(p(table), table.numel(), code, p(src_rows), None) == 0
        else:
            src_rows = table
        def run():
            rc = fn(p(src_rows), p(rb), None, p(rows), p(l_of_d), p(W1), p(W2), p(gate), p(src), E, D, C, V, NL, None, p(B),
                    p(Ag), p(gram), p(ws), 16, None)
            assert rc == 0, rc
        B.zero_(); Ag.zero_()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        eB = float((B[:ns] - Bref).abs().max() / Bref.abs().max())
        eA = float((Ag[:ns] - Agref).abs().max() / Agref.abs().max())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        if hasattr(lib, "lcao_debug_set_lowp_clock"):
            nw = 256 * 8
            clk = torch.zeros(nw * 13, dtype=torch.int64, device=dev)
            lib.lcao_debug_set_lowp_clock.argtypes = [ctypes.c_void_p]
            clk0 = torch.zeros_like(clk)
            lib.lcao_debug_set_lowp_clock(p(clk0))
            run()
            lib.lcao_debug_set_lowp_clock(p(clk))
            run(); torch.cuda.synchronize()
            lib.lcao_debug_set_lowp_clock(None)
            c0 = clk0.view(nw, 13).double(); c0 = c0[c0[:, 10] > 0]
            c1 = clk.view(nw, 13).double(); c1 = c1[c1[:, 10] > 0]
            print(f"   back-to-back launches: last wave end of launch 1 -> first wave start of launch 2: "
                  f"{(c1[:, 12].min() - (c0[:, 12] + c0[:, 11]).max()).item() / 100:.1f} us")
            c = clk.view(nw, 13).double()
            c = c[c[:, 10] > 0]
            tot, real = c[:, 10].mean().item(), c[:, 11].mean().item()
            st, en = c[:, 12], c[:, 12] + c[:, 11]
            print(f"   wave spans (us): min {c[:, 11].min().item() / 100:.0f} mean {real / 100:.0f} max {c[:, 11].max().item() / 100:.0f}; "
                  f"first start -> last start {(st.max() - st.min()).item() / 100:.0f}, first start -> last end {(en.max() - st.min()).item() / 100:.0f}, "
                  f"first end -> last end {(en.max() - en.min()).item() / 100:.0f}")
            print(f"   in-kernel clock {tot / real * 100:.0f} MHz; wave cycles {tot:.0f}; per phase (% of wave cycles): " +
                  ", ".join(f"{PH[i]} {100 * c[:, i].mean().item() / tot:.1f}" for i in range(9)))
        print(f"{os.path.basename(path):28s} {name[26:]:7s} {ms:7.3f} ms  {2*D*(C*V+V*V)*E/ms/1e9:7.1f} TFLOP/s  errB {eB:.2e} errAg {eA:.2e}", flush=True)
