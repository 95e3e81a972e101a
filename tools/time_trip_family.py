"""Time the three-body bilinear family (lcao_trip_spmm / spmm_t / sddmm, NL = n_l and the cosine form NL = 1) and the other
force-training primitives on the cfg2 batch shape.      python tools/time_trip_family.py [--n-cells 1000] [--iters 10]"""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lcaonet_amd import ops  # noqa: E402
from lcaonet_amd.data.synthetic import jittered_cubic_cells  # noqa: E402
from lcaonet_amd.model.double import _Geo  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-cells", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen_kw, model_kw = bench.WORKLOADS["cfg2"]
    batch = jittered_cubic_cells(seed=1234, device=dev, **dict(gen_kw, n_cells=args.n_cells))
    m = bench.build_model(model_kw, 1234, dev)
    geo = _Geo(m, batch)
    E, T, V, L = geo.E, geo.T, 128, geo.n_l
    print(f"E {E}  T {T}  n_l {L}  V {V}")
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    w, A, G = rnd(T, L), rnd(E, L, V), rnd(E, V)
    w1, u4 = rnd(T, 1), rnd(E, 1, 4)
    gb = lambda n_bytes, ms: n_bytes / ms / 1e6
    rows = [
        ("spmm   (w, A) -> (E,V)", lambda: ops.trip_spmm(w, A, geo), 4 * (T * L + E * L * V + E * V)),
        ("spmm_t (w, G) -> (E,L,V)", lambda: ops.trip_spmm_t(w, G, geo), 4 * (T * L + E * V + E * L * V)),
        ("sddmm  (G, A) -> (T,L)", lambda: ops.trip_sddmm(G, A, geo), 4 * (T * L + E * V + E * L * V)),
        ("spmm   cos", lambda: ops.trip_spmm(w1, u4, geo), 4 * (T + 8 * E)),
        ("spmm_t cos", lambda: ops.trip_spmm_t(w1, u4[:, 0], geo), 4 * (T + 8 * E)),
        ("sddmm  cos", lambda: ops.trip_sddmm(u4[:, 0], u4, geo), 4 * (T + 8 * E)),
    ]
    c = geo.ctx
    D = len(geo.l_of_d)
    rb, Tt, gate = rnd(E, D), rnd(geo.zdim ** 2, D, V), rnd(geo.N, V)
    gAg, gB, gG, h_rb, h_gate = rnd(E, L, V), rnd(E, V), rnd(E, L, L), rnd(E, D), rnd(geo.N, V)
    asl = lambda lv, **kw: ops.aside_levels(lv, Tt, rb, geo.l_of_d, geo.z, c.src_p, c.dst_p, gate, L, geo.zdim, **kw)
    asp = lambda lv, **kw: ops.aside_pairs(lv, Tt, rb, geo.l_of_d, geo.z, c.src_p, c.dst_p, gate, L, geo.zdim, geo.perm_pair,
                                           geo.chunk_ptr, gAg, gB, gG, **kw)
    g10, g_rb = rnd(E, 10), torch.zeros(E, D, device=dev)
    row = 4 * V
    rows += [
        ("aside level 0", lambda: asl(0), E * (row * (L + 1) + 4 * D + 4 * L * L)),
        ("aside level 1", lambda: asl(1, gAg=gAg, gB=gB, gG=gG), E * (row * (L + 3) + 8 * D)),
        ("aside level 1 + gA", lambda: asl(1, gAg=gAg, gB=gB, gG=gG, want_table_terms=True), E * (row * (2 * L + 3) + 8 * D)),
        ("aside level 2 + gA, dA", lambda: asl(2, gAg=gAg, gB=gB, gG=gG, h_rb=h_rb, h_gate=h_gate, want_table_terms=True),
         E * (row * (4 * L + 6) + 12 * D)),
        ("aside_pairs level 1 (+ table)", lambda: asp(1), E * (row * (L + 3) + 8 * D)),
        ("aside_pairs level 2 (+ table)", lambda: asp(2, h_rb=h_rb, h_gate=h_gate), E * (row * (2 * L + 6) + 12 * D)),
        ("pair_contract_bwd (forces)", lambda: ops.pair_contract_bwd(Tt, rb, None, geo.l_of_d, geo.z, c.src_p, c.dst_p, gate, gAg,
                                                                      gB, g10, E, L, geo.zdim, g_rb), E * (row * (L + 3) + 8 * D)),
        ("pair_contract_bwd + gA (training)", lambda: ops.pair_contract_bwd(Tt, rb, None, geo.l_of_d, geo.z, c.src_p, c.dst_p,
                                                                             gate, gAg, gB, g10, E, L, geo.zdim, g_rb, want_gA=True),
         E * (row * (2 * L + 3) + 8 * D)),
        ("pair_table_grad", lambda: ops.pair_table_grad(gAg, rb, None, geo.l_of_d, geo.perm_pair, geo.chunk_ptr), E * (row * L + 4 * D)),
    ]
    Xe, We, Ge = rnd(E, V), rnd(V, V), rnd(E, V)
    rows += [
        ("dense (E,V) x (V,V)", lambda: ops.dense(Xe, We), 8 * E * V),
        ("wgrad (E,V)^T (E,V)", lambda: ops.wgrad(Ge, Xe), 8 * E * V),
    ]
    for name, fn, nbytes in rows:
        ms = timed(fn, args.iters)
        print(f"{name:36s} {ms:8.3f} ms   {gb(nbytes, ms):8.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
