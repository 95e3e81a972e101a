"""BASELINE.md §4 step 1 (build container only; the reference cannot travel): time the REFERENCE ITSELF — imported from
/root/reference exactly as oracle/make_golden.py does (oracle/standins/ for torch_geometric / torch_scatter / torch_sparse) —
and the CPU restatement (oracle/restatement.py) on the same BASELINE config-1 inputs and weights, 8 threads, and compare
their outputs.  The restatement is what bench.py times on the GPU box as `cpu_baseline` (kind "port"); this script is the
record that it costs what the reference costs.   usage: python tools/time_reference_vs_restatement.py [> profiles/r02/...]"""
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "oracle", "standins"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

from lcaonet.model import LCAONet as RefLCAONet  # noqa: E402  (the reference)
from torch_geometric.data import Data  # noqa: E402  (stand-in container)

from lcaonet_amd.data.synthetic import random_periodic_cells  # noqa: E402
from oracle import restatement as O  # noqa: E402

torch.set_num_threads(8)
kw = dict(emb_size=64, n_interaction=2, rbf_type="hydrogen", cutoff=5.0)
torch.manual_seed(0)
ref = RefLCAONet(**kw).eval()
sd = {k: v.clone() for k, v in ref.state_dict().items()}
g = random_periodic_cells(32, 20, 7.4, 0.99 * 5.0, seed=1234).to_dict()
E = g["edge_index"].shape[1]
cfg = O.OracleConfig(**kw)


def run_ref():
    with torch.no_grad():
        return ref(Data(**{k: v.clone() for k, v in g.items()}))


def run_port():
    with torch.no_grad():
        return O.forward(sd, cfg, g)


def median_time(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


a, b = run_ref(), run_port()
err = float((a - b).abs().max() / a.abs().max())
t_ref, t_port = median_time(run_ref), median_time(run_port)
print(f"BASELINE config 1: 32 x 20-atom cells, E = {E}, LCAONet(emb_size=64, n_interaction=2, cutoff=5.0), fp32, eval, "
      f"torch {torch.__version__}, 8 threads, median of 5 after 1 warm-up")
print(f"reference (lcaonet.model.LCAONet via oracle/standins): {t_ref * 1e3:8.1f} ms = {E / t_ref:9.0f} edges/s")
print(f"restatement (oracle/restatement.py O.forward)        : {t_port * 1e3:8.1f} ms = {E / t_port:9.0f} edges/s")
print(f"time ratio restatement / reference = {t_port / t_ref:.3f} (BASELINE.md asks for 0.8 .. 1.2);  "
      f"max|out_ref - out_port| / max|out_ref| = {err:.2e}")
