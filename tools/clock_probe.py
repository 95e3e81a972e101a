"""In-kernel clock of lcao_coeff_mlp_contract on the config-2 workload: delta s_memtime / delta s_memrealtime * 100 MHz
per wave (MI355X_MICROARCH.md 'DVFS give-back' item 6), after ~2 s of back-to-back launches.  Diagnostic only."""
import ctypes, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcaonet_amd import _lib
from bench import WORKLOADS, build_model
from lcaonet_amd.data.synthetic import jittered_cubic_cells

dev = torch.device("cuda:0")
gen_kw, model_kw = WORKLOADS["cfg2"]
model = build_model(model_kw, 1234, dev)
batch = jittered_cubic_cells(seed=1234, device=dev, **gen_kw)
lib = _lib.load()
with torch.no_grad():
    t0 = time.time()
    while time.time() - t0 < 2.0:
        model(batch)
    torch.cuda.synchronize()
    buf = torch.zeros(4 * 8 * 512, dtype=torch.int64, device=dev)
    lib.lcao_debug_set_clock_probe(ctypes.c_void_p(buf.data_ptr()))
    model(batch)
    torch.cuda.synchronize()
    lib.lcao_debug_set_clock_probe(None)
raw = buf.view(-1, 4).cpu()
raw = raw[raw[:, 1] > 0]
span = (raw[:, 3].max() - raw[:, 2].min()).item()
print(f"loop start spread = {(raw[:,2].max()-raw[:,2].min()).item()/100:.1f} us, end spread = {(raw[:,3].max()-raw[:,3].min()).item()/100:.1f} us, "
      f"first-start..last-end = {span/100:.1f} us")
w = raw[:, 1].double() / 100
print(f"per-wave loop wall us: min={w.min():.0f} median={w.median():.0f} p90={w.quantile(0.9):.0f} max={w.max():.0f}")
# per workgroup (8 waves) max
wg = w.view(-1, 8).max(dim=1).values
print(f"per-WG loop wall us: min={wg.min():.0f} median={wg.median():.0f} max={wg.max():.0f}; slowest WGs: {torch.topk(wg, 8).indices.tolist()}")
v = raw.double()
ghz = v[:, 0] / v[:, 1] * 0.1
print(f"waves={len(v)} shader-cycles/wave median={v[:,0].median():.0f}  wall(100MHz ticks) median={v[:,1].median():.0f}")
print(f"in-kernel clock GHz: median={ghz.median():.3f} min={ghz.min():.3f} max={ghz.max():.3f}")
E = batch['edge_index'].shape[1]
mfma_cycles = (E / 16) * 8 * 512 * 32 / 1024   # per SIMD
print(f"MFMA-busy cycles per SIMD = {mfma_cycles:.0f}; pipe utilisation at the measured clock = {mfma_cycles / v[:,0].median():.3f}")
