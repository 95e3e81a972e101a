import sys, time
sys.path.insert(0, ".")
import torch
from bench import build_model
from lcaonet_amd import ops
from lcaonet_amd.data.synthetic import random_molecule_batch
dev = torch.device("cuda", 0)
batch = random_molecule_batch(8000, cutoff_graph=5.94, seed=1, device=dev)
E = batch["edge_index"].shape[1]
m = build_model(dict(emb_size=128, emb_size_coeff=128, emb_size_conv=128, n_interaction=3, cutoff=6.0, max_z=10), 1, dev)
with torch.no_grad():
    for prec in ("fp32", "f16"):
        kw = {} if prec == "fp32" else {"precision": prec}
        for _ in range(3): m(batch, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m(batch, **kw)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
        with ops.profile() as p:
            m(batch, **kw)
            ks = p.summary()
        top = sorted(ks.items(), key=lambda kv: -kv[1][1])[:6]
        print(prec, f"E={E} {ms:.2f} ms/step {E/ms/1e3:.1f} M edges/s", [(k, round(v[1], 3)) for k, v in top], flush=True)
