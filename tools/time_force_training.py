"""Time one force-training step (regress_forces=True, direct_forces=False, train mode: energy + exact autograd forces +
loss.backward through the force graph + Adam) on the cfg2 batch shape, and break it down per C-ABI entry point.

    python tools/time_force_training.py [--n-cells 1000] [--steps 3] [--torch-profile]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lcaonet_amd import ops  # noqa: E402
from lcaonet_amd.data.synthetic import jittered_cubic_cells  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-cells", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--torch-profile", action="store_true")
    ap.add_argument("--mode", default="forces", choices=["forces", "energy", "infer-forces"],
                    help="forces: train on energy + autograd forces; energy: train on energy only; infer-forces: eval, no grad")
    ap.add_argument("--chunks", type=int, default=1, help="model.force_training_chunks")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen_kw, model_kw = bench.WORKLOADS["cfg2"]
    gen_kw = dict(gen_kw, n_cells=args.n_cells)
    batch = jittered_cubic_cells(seed=1234, device=dev, **gen_kw)
    E = int(batch["edge_index"].shape[1])
    N = int(batch["z"].numel())
    forces = args.mode != "energy"
    m = bench.build_model(dict(model_kw, regress_forces=forces, direct_forces=False) if forces else model_kw, 1234, dev)
    m = m.eval() if args.mode == "infer-forces" else m.train()
    m.force_training_chunks = args.chunks
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    e_t = torch.zeros(args.n_cells, 1, device=dev)
    f_t = torch.zeros(N, 3, device=dev)

    def step():
        if args.mode == "infer-forces":
            with torch.no_grad():
                return m(batch)
        opt.zero_grad(set_to_none=True)
        if forces:
            e, f = m(batch)
            loss = torch.nn.functional.l1_loss(e, e_t) + torch.nn.functional.l1_loss(f, f_t)
        else:
            loss = torch.nn.functional.l1_loss(m(batch), e_t)
        loss.backward()
        opt.step()
        return loss

    step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    res = {"n_cells": args.n_cells, "N": N, "E": E, "ms_per_step": round(ms, 2), "edges_per_s": E / (ms * 1e-3),
           "peak_gib": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
    with ops.profile() as prof:
        step()
        ks = prof.summary()
    res["c_abi_ms"] = {k: [v[0], round(v[1], 3)] for k, v in sorted(ks.items(), key=lambda kv: -kv[1][1])}
    res["c_abi_total_ms"] = round(sum(v[1] for v in ks.values()), 2)
    if args.torch_profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as p:
            step()
            torch.cuda.synchronize()
        rows = []
        res["device_ms_total"] = round(sum(getattr(ev, "self_device_time_total", 0) for ev in p.key_averages()) / 1e3, 2)
        res["kernel_launches"] = sum(ev.count for ev in p.key_averages()
                                     if getattr(ev, "self_device_time_total", 0) and not ev.cpu_time_total)
        for ev in p.key_averages():
            t = getattr(ev, "device_time_total", None) or getattr(ev, "cuda_time_total", 0)
            if t:
                rows.append((ev.key[:90], ev.count, round(t / 1e3, 3)))
        rows.sort(key=lambda r: -r[2])
        res["torch_device_ms_top"] = rows[:40]
        by_shape = []
        for ev in p.key_averages(group_by_input_shape=True):
            t = getattr(ev, "self_device_time_total", 0)
            if t and ev.key.startswith("aten::"):
                by_shape.append((ev.key, str(ev.input_shapes)[:80], ev.count, round(t / 1e3, 3)))
        by_shape.sort(key=lambda r: -r[3])
        res["aten_by_shape_top"] = by_shape[:40]
        res["aten_self_device_ms"] = round(sum(r[3] for r in by_shape), 2)
    s = json.dumps(res, indent=1)
    print(s)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s)


if __name__ == "__main__":
    main()
