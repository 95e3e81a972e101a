#!/usr/bin/env python
"""bench.py — forward edges/s of the LCAOConv stack on the BASELINE config-2 workload, one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                 (N > 1, one rank per GPU, RCCL)

A step = one `model(batch)` forward (eval, fp32) over one resident batch: geometry, radial basis, embedding,
L interaction layers (coefficient MLP + three-body + two-body + node scatter), output head.  Weak scaling: every
rank owns its own batch of whole crystals (cells are independent graphs, so the data path has no collective);
`value` = edges processed by all ranks / max-over-ranks time.  Prints ONE JSON line on rank 0.

Other modes (never the driver's default):
  --scaling strong          the SAME 1 000-crystal batch cut into N slices of whole crystals; each rank replays its slice
                            as one HIP graph (lcaonet_amd.model.CapturedForward), so launch gaps do not grow with N
  --workload cfg4           BASELINE config 4: ONE periodic crystal cut into N slabs of x-layers, one per rank; every rank
                            builds only its slab (+ halo) and its plan (shard.plan_from_local), and runs
                            LCAONet.forward_sharded: L-1 all-to-alls of boundary rows over RCCL per forward, overlapped
                            with the owned-edge coefficient kernel, + one all-reduce of the per-graph sums.
                            126^3 atoms at N = 8 (the config), 250 k atoms per GPU otherwise (--n-side overrides)
  --emulate-world W         (N = 1) time rank --emulate-rank's slab of a W-way partition on this one GPU; its ghost rows
                            come from a device-local stand-in
LCAO_BENCH_REHEARSAL=1 runs every rank on cuda:0 over gloo (control-flow rehearsal on a one-GPU box; numbers meaningless).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# MI355X ceilings (/opt/skills/guides/MI355X_MICROARCH.md: "Chip-level parameters")
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_16BIT_MFMA_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA ("~2.5 PF dense"; the 5 PF figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (generator kwargs, model kwargs)
    "cfg2": (dict(n_cells=1000, n_side=4, a=3.2, sigma=0.05, cutoff_graph=4.95, max_z=36),
             dict(emb_size=128, emb_size_coeff=128, emb_size_conv=128, n_interaction=3, cutoff=5.0)),
    # one rank's share (1/8) of BASELINE config 4: a 63^3 = 250 047-atom periodic crystal, cutoff 6.0, 32 nearest neighbours
    "cfg4-1of8": (dict(crystal=True, n_side=63, a=3.2, sigma=0.05, cutoff_graph=5.94, max_neighbors=32, max_z=36),
                  dict(emb_size=128, emb_size_coeff=128, emb_size_conv=128, n_interaction=3, cutoff=6.0)),
    # one rank's share (1/8) of BASELINE config 5: 16 250 QM9-shaped molecules (3..29 atoms, z in {1,6,7,8,9}), cutoff 6.0,
    # hidden 256, 6 interactions, energy + forces via the reverse sweep (regress_forces=True, direct_forces=False)
    "cfg5-1of8": (dict(molecules=True, n_mol=16250, cutoff_graph=5.94),
                  dict(emb_size=256, emb_size_coeff=128, emb_size_conv=128, n_interaction=6, cutoff=6.0, max_z=10,
                       regress_forces=True, direct_forces=False)),
    # BASELINE config 4, one crystal over all ranks (run_sharded_crystal); n_side is chosen from the number of ranks
    "cfg4": (dict(sharded_crystal=True, a=3.2, sigma=0.05, cutoff_graph=5.94, max_neighbors=32, max_z=36),
             dict(emb_size=128, emb_size_coeff=128, emb_size_conv=128, n_interaction=3, cutoff=6.0)),
    "cfg2-small": (dict(n_cells=64, n_side=4, a=3.2, sigma=0.05, cutoff_graph=4.95, max_z=36),
                   dict(emb_size=128, emb_size_coeff=128, emb_size_conv=128, n_interaction=3, cutoff=5.0)),
}


def build_model(kw, seed, dev):
    from lcaonet_amd.model import LCAONet

    torch.manual_seed(seed)
    m = LCAONet(**kw).eval()
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():  # BASELINE/SURVEY §8d: BN running stats mean~N(0,0.1), var~U(0.5,1.5)
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    return m.to(dev)


def cpu_baseline(max_seconds: float = 30.0) -> dict:
    """The CPU oracle (a literal-dataflow restatement of the reference, oracle/restatement.py; timed beside the reference
    itself in the build container: profiles/r02/reference_vs_restatement.txt) on this box's host cores — BASELINE.md
    section 4 step 2: BASELINE config 1 (the reference's own CPU-runnable case) and a subset of config 2, median of 5
    runs after a warm-up, thread count stated."""
    from lcaonet_amd.data.synthetic import jittered_cubic_cells, random_periodic_cells
    from lcaonet_amd.model import LCAONet
    from oracle import restatement as O

    kw = dict(emb_size=64, n_interaction=2, rbf_type="hydrogen", cutoff=5.0)
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in LCAONet(**kw).eval().state_dict().items()}
    g = random_periodic_cells(32, 20, 7.4, 0.99 * 5.0, seed=1234).to_dict()
    E = g["edge_index"].shape[1]
    cfg = O.OracleConfig(**kw)

    def median_of(n, fn):
        fn()  # warm-up
        times = []
        for _ in range(n):
            t0 = time.time()
            fn()
            times.append(time.time() - t0)
        return sorted(times)[n // 2]

    # the literal dataflow is a chain of small / medium ATen ops: more threads is not faster.  One run per candidate thread
    # count picks the count; the reported number is the median of 5 at that count (cores = the threads actually used).
    ncpu = os.cpu_count() or 1
    candidates = sorted({min(ncpu, c) for c in (8, 16, 32, ncpu)})
    t_start = time.time()
    with torch.no_grad():
        probe = {}
        for cores in candidates:
            if probe and (time.time() - t_start) > max_seconds / 3:
                break
            torch.set_num_threads(cores)
            probe[cores] = median_of(1, lambda: O.forward(sd, cfg, g))
        cores = min(probe, key=probe.get)
        torch.set_num_threads(cores)
        med = median_of(5, lambda: O.forward(sd, cfg, g))
        # a subset of BASELINE config 2 (same generator, model and seed as the GPU workload, fewer crystals: the literal
        # dataflow holds ~3 T D V 4 B of triplet temporaries — 65-80 GB for the whole 1 000-crystal batch)
        gen_kw, model_kw = WORKLOADS["cfg2"]
        n_sub = 12
        sub = jittered_cubic_cells(seed=1234, device="cpu", **dict(gen_kw, n_cells=n_sub)).to_dict()
        torch.manual_seed(1234)
        sd2 = {k: v.clone() for k, v in LCAONet(**model_kw).eval().state_dict().items()}
        cfg2 = O.OracleConfig(**model_kw)
        E2 = sub["edge_index"].shape[1]
        med2 = median_of(5, lambda: O.forward(sd2, cfg2, sub)) if (time.time() - t_start) < max_seconds else None
    out = {"value": E / med, "unit": "edges/s", "cores": cores, "kind": "port",
           "sample": f"BASELINE config 1: LCAONet(emb_size=64,n_interaction=2,cutoff=5.0) forward on 32x20-atom cells, "
                     f"E={E}, median of 5 runs after a warm-up at {cores} torch CPU threads (the fastest of {sorted(probe)}, "
                     f"one run each; host has {ncpu})"}
    if med2 is not None:
        out["cfg2_subset"] = {"value": E2 / med2, "unit": "edges/s", "cores": cores,
                              "sample": f"BASELINE config 2 model on the first {n_sub} of its 1 000 crystals (E={E2}), median of 5"}
    return out


PMC_FILE = os.path.join("profiles", "r03", "pmc_traffic.json")


def pmc_traffic(kernel: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r03/pmc_traffic.json: separate FETCH_SIZE /
    WRITE_SIZE passes of this same workload, gfx950 x2 read correction), or None.  NOT measured in this run: the file
    carries the commit and date of its passes, repeated in the line as `traffic_source`."""
    try:
        return json.load(open(os.path.join(REPO, PMC_FILE)))[kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def pmc_source():
    try:
        d = json.load(open(os.path.join(REPO, PMC_FILE)))
        return f"{PMC_FILE} (PMC passes at commit {d.get('_commit')}, {d.get('_date')}; not re-measured in this run)"
    except Exception:
        return None


ROOFLINE_KERNELS = ("lcao_coeff_mlp_contract", "lcao_triplet_gemm", "lcao_triplet_sum")


def forward_work_per_edge(model, n_nodes: int, n_edges: int) -> tuple[float, float]:
    """ALGORITHMIC bytes and GEMM flops per edge of one whole forward (general per-edge cst, fp32, int32
    indices; one HBM round trip wherever a stage needs other edges' results)."""
    D, C, V, H, L = model.elec_info.n_orb, model.emb_size_coeff, model.emb_size_conv, model.emb_size, model.n_interaction
    ne = n_nodes / max(n_edges, 1)
    basis = 8 + 12 + ne * 12 + 4 + 12 + 4 * D
    per_layer = 4.0 * (D * C + 3 * D * V + 2 * D + 2 * V + 5) + ne * 4.0 * (H + 2 * V + V + H)
    flop_layer = 2.0 * D * (C * V + V * V) + 4.0 * V * V + 2.0 * (2 * V * V + V * V) + ne * (2.0 * H * 2 * V + 2.0 * V * H)
    return basis + L * per_layer, L * flop_layer


def roofline_forward(model, n_nodes: int, n_edges: int, seconds_per_forward: float) -> dict:
    byts, flop = forward_work_per_edge(model, n_nodes, n_edges)
    return {"note": "whole forward against both ceilings (SURVEY.md 8d): algorithmic bytes / flops per edge x E / t",
            "bytes_per_edge": round(byts, 1), "gemm_flop_per_edge": round(flop, 1),
            "hbm_frac": round(byts * n_edges / seconds_per_forward / (PEAK_HBM_GBS * 1e9), 4),
            "mfma_frac": round(flop * n_edges / seconds_per_forward / (PEAK_F32_MFMA_TFLOPS * 1e12), 4)}


def mode_roofline(fn, steps: int, work: dict, n_edges: int):
    """Roofline block of a secondary mode: `fn()` (one step of the mode) runs `steps` times with every C-ABI call
    bracketed by HIP events; the entry point with the largest total time is the mode's dominant kernel and is priced with
    its ALGORITHMIC work per launch from `work` = {entry point: (bound, per-edge flop or bytes, peak, unit, note)}."""
    from lcaonet_amd import ops

    with ops.profile() as prof:
        for _ in range(steps):
            fn()
        ks = prof.summary()
    if not ks:
        return None
    # the kernel that is priced: the largest single KERNEL of the mode among those with a stated algorithmic cost
    # (lcao_dense / lcao_wgrad are many launches of different shapes: their totals are listed, not priced)
    known = {k: v for k, v in ks.items() if k in work}
    name, (calls, ms) = max((known or ks).items(), key=lambda kv: kv[1][1])
    total = sum(v[1] for v in ks.values())
    out = {"kernel": name, "avg_launch_ms": round(ms / calls, 4), "launches_per_step": round(calls / steps, 2),
           "share_of_bracketed_time": round(ms / total, 3),
           "c_abi_ms_per_step": round(total / steps, 3), "c_abi_launches_per_step": round(sum(v[0] for v in ks.values()) / steps, 1),
           "top_entry_points_ms_per_step": {k: round(v[1] / steps, 3) for k, v in sorted(ks.items(), key=lambda kv: -kv[1][1])[:4]}}
    if name in work:
        bound, per_edge, peak, unit, note = work[name]
        rate = per_edge * n_edges / (ms / calls * 1e-3) / (1e12 if bound == "mfma" else 1e9)
        out.update({"bound": bound, "achieved": round(rate, 1), "peak": peak, "unit": unit, "frac": round(rate / peak, 4),
                    "traffic": None, "algorithmic_per_launch": per_edge * n_edges, "note": note})
    return out


def run_strong(args, model, gen_kw, rank, world, dev, barrier, reduce_times):
    """Strong scaling of config 2: the SAME batch (seed 1234) cut into `world` slices of whole crystals.  Each rank's
    forward is one HIP-graph replay (index structures of the fixed edge list built once, outside the capture), so the
    ~0.5 ms of launch gaps of the eager forward do not weigh more as the per-rank work shrinks."""
    from lcaonet_amd.data.synthetic import jittered_cubic_cells
    from lcaonet_amd.model import CapturedForward

    n_cells = gen_kw["n_cells"]
    c0, c1 = rank * n_cells // world, (rank + 1) * n_cells // world
    batch = jittered_cubic_cells(seed=1234, device=dev, cell_range=(c0, c1), **gen_kw)
    E, N = int(batch["edge_index"].shape[1]), int(batch["z"].numel())
    step = CapturedForward(model, batch)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    elapsed_max, e_total = reduce_times(elapsed, E)
    return {"metric": "edges/sec forward (LCAOConv stack), 1M-edge synthetic batch", "value": e_total * args.steps / elapsed_max,
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload} strong: the {n_cells}-crystal batch of seed 1234 in {world} slices of whole "
                                   f"crystals (rank 0: cells [{c0}, {c1}), N={N}, E={E}); whole forward replayed as one HIP graph "
                                   "per rank (CSR of the fixed edge list built once outside the capture)",
                       "edges_total": int(e_total), "sharding": "whole crystals per rank, no collective"},
            "roofline_forward": roofline_forward(model, N, E, elapsed / args.steps)}


def run_sharded_crystal(args, model, gen_kw, rank, world, dev, dist, barrier, reduce_times):
    """BASELINE config 4: one crystal, one slab of x-layers per rank, LCAONet.forward_sharded over RCCL."""
    from lcaonet_amd import shard
    from lcaonet_amd.data.synthetic import jittered_cubic_crystal_slab

    emu = args.emulate_world if world == 1 else 0
    parts = emu or world
    n_side = args.n_side or (126 if parts == 8 else int(round(63.0 * parts ** (1.0 / 3.0))))
    ck = {k: v for k, v in gen_kw.items() if k != "sharded_crystal"}
    t0 = time.perf_counter()
    g, gid, n_own, owner_of = jittered_cubic_crystal_slab(n_side, args.emulate_rank if emu else rank, parts, seed=1234,
                                                          device=dev, **ck)
    plan = shard.plan_from_local(args.emulate_rank if emu else rank, parts, g, gid, n_own, owner_of, negotiate=not emu)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    H = model.emb_size
    kw = {}
    if emu:     # no peers exist: ghost rows from a local stand-in, partial sums kept as they are
        ghost_rows = torch.randn(plan.n_ghost, H, device=dev)
        kw = dict(exchange_fn=lambda x_own: ghost_rows, reduce_fn=lambda part: part)
    with torch.no_grad():
        for _ in range(args.warmup):
            out = model.forward_sharded(plan, **kw)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model.forward_sharded(plan, **kw)
        barrier()
        elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    elapsed_max, e_total = reduce_times(elapsed, plan.e_own)
    e_loc = int(plan.graph["edge_index"].shape[1])
    stats = torch.tensor([plan.n_own, plan.n_ghost, plan.e_own, e_loc, plan.n_send, sum(1 for s in plan.send_splits if s)],
                         dtype=torch.float64)
    if dist is not None:
        st = stats.clone() if rehearsal_cpu(dist) else stats.to(dev)
        allst = [torch.zeros_like(st) for _ in range(world)]
        dist.all_gather(allst, st)
        allst = torch.stack([a.cpu() for a in allst])
    else:
        allst = stats.unsqueeze(0)
    L = model.n_interaction
    return {"metric": "edges/sec forward (LCAOConv stack), one crystal sharded by destination node", "value": e_total * args.steps / elapsed_max,
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg4: ONE periodic jittered {n_side}^3 = {n_side ** 3}-atom crystal, cutoff 6.0 (graph 5.94), "
                                   f"<= 32 nearest neighbours, in {parts} slabs of x-layers"
                                   + (f"; rank {args.emulate_rank} of {parts} EMULATED on one GPU (stand-in ghost rows)" if emu else "")
                                   + f"; LCAONet(emb_size={H}, coeff={model.emb_size_coeff}, conv={model.emb_size_conv}, "
                                   f"n_interaction={L}), eval, whole forward_sharded",
                       "owned_edges_total": int(e_total), "sharding": "centre nodes by x-layer slab; halo in-edges replicated",
                       "per_rank": {"n_own": allst[:, 0].tolist(), "n_ghost": allst[:, 1].tolist(), "e_own": allst[:, 2].tolist(),
                                    "e_local": allst[:, 3].tolist(), "rows_sent_per_exchange": allst[:, 4].tolist(),
                                    "peers": allst[:, 5].tolist()},
                       "exchange": {"collective": "all_to_all_single, exact split sizes (RCCL)" if not emu else "stand-in",
                                    "per_forward": L - 1, "bytes_sent_per_rank_per_exchange": (allst[:, 4] * H * 4).tolist(),
                                    "overlap": "launched asynchronously; the next layer's owned-edge coefficient kernel runs under it"},
                       "slab_and_plan_build_s_rank0": round(t_plan, 2),
                       "peak_memory_GiB_rank0": round(torch.cuda.max_memory_allocated() / 2**30, 1)},
            "roofline_forward": roofline_forward(model, plan.n_own, plan.e_own, elapsed / args.steps)}


def rehearsal_cpu(dist) -> bool:
    return dist.get_backend() == "gloo"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary modes (profiling runs)")
    ap.add_argument("--fused", type=int, default=-1, help="-1 auto, 0 generic kernels only, 1 force fused")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "f16"],
                    help="NOT for the headline: run the timed loop in a BASELINE config-3 precision mode (profiling)")
    ap.add_argument("--n-side", type=int, default=0, help="cfg4: atoms per crystal edge (0 = 126 at 8 ranks, 250 k atoms per rank otherwise)")
    ap.add_argument("--emulate-world", type=int, default=0, help="cfg4 on ONE GPU: time one rank's slab of a W-way partition")
    ap.add_argument("--emulate-rank", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # LCAO_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend — rehearses the N>1 control flow on a one-GPU
    # box (the number it prints is meaningless); the driver's runs use one GPU per rank over RCCL
    rehearsal = os.environ.get("LCAO_BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from lcaonet_amd import ops
    from lcaonet_amd.data.synthetic import jittered_cubic_cells

    gen_kw, model_kw = WORKLOADS[args.workload]
    model = build_model(model_kw, 1234, dev)                      # identical weights on every rank

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_times(elapsed, units):
        t = torch.tensor([elapsed, float(units)], dtype=torch.float64, device="cpu" if rehearsal else dev)
        if dist is None:
            return elapsed, float(units)
        tmax, tsum = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        return float(tmax[0]), float(tsum[1])

    if gen_kw.get("sharded_crystal"):
        line = run_sharded_crystal(args, model, gen_kw, rank, world, dev, dist, barrier, reduce_times)
    elif args.scaling == "strong":
        line = run_strong(args, model, gen_kw, rank, world, dev, barrier, reduce_times)
    else:
        line = None
    if gen_kw.get("sharded_crystal") or args.scaling == "strong":
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if gen_kw.get("crystal"):
        from lcaonet_amd.data.synthetic import jittered_cubic_crystal
        ck = {k: v for k, v in gen_kw.items() if k != "crystal"}
        batch = jittered_cubic_crystal(seed=1234 + rank, device=dev, **ck)
        gen_kw = dict(gen_kw, n_cells=1)
    elif gen_kw.get("molecules"):
        from lcaonet_amd.data.synthetic import random_molecule_batch
        batch = random_molecule_batch(gen_kw["n_mol"], cutoff_graph=gen_kw["cutoff_graph"], seed=1234 + rank, device=dev)
    else:
        batch = jittered_cubic_cells(seed=1234 + rank, device=dev, **gen_kw)  # this rank's own crystals
    E = int(batch["edge_index"].shape[1])
    N = int(batch["z"].numel())
    fused = None if args.fused < 0 else bool(args.fused)

    with torch.no_grad():
        mkw = {} if args.precision == "fp32" else {"precision": args.precision}
        for _ in range(args.warmup):
            out = model(batch, fused=fused, **mkw)
        barrier()
        t0 = time.perf_counter()
        # HIP events (torch's current stream = the stream the library launches on) around the two roofline kernels
        # only: bracketing all ~60 calls of a forward costs ~1 ms per step
        with ops.profile(only=ROOFLINE_KERNELS) as prof:
            for _ in range(args.steps):
                out = model(batch, fused=fused, **mkw)
            barrier()
            elapsed = time.perf_counter() - t0
            ksum = prof.summary()
    with_forces = isinstance(out, tuple)
    if with_forces:
        assert bool(torch.isfinite(out[1]).all())
        out = out[0]
    assert bool(torch.isfinite(out).all())
    elapsed_max, e_total = reduce_times(elapsed, E)

    # secondary measurements (rank 0 only, after the headline region and its reduction; a failure here is reported in
    # the line, it cannot take the headline down)
    def timed(**kw):
        with torch.no_grad():
            o = model(batch, **kw)
            err = float((o - out).abs().max() / out.abs().max())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                model(batch, **kw)
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t1) / args.steps, err

    secondary_error = None
    pair_ms = bf_ms = bp_ms = pair_err = bf_err = bp_err = forces_ms = train_ms = ftrain_ms = mat_ms = mat_err = h_ms = h_err = None
    roof_bf = roof_h = roof_forces = roof_train = roof_ftrain = replay = None
    D_, C_, V_, NL_ = model.elec_info.n_orb, model.emb_size_coeff, model.emb_size_conv, model.n_l
    mlp_flop = 2.0 * D_ * (C_ * V_ + V_ * V_)
    # algorithmic work per edge of the kernels that dominate the secondary modes
    mode_work = {
        "lcao_coeff_mlp_contract_f16": ("mfma", mlp_flop, PEAK_16BIT_MFMA_TFLOPS, "TFLOP/s",
                                        "one v_mfma_f32_16x16x32_f16 per product; the kernel is VALU-issue bound: two "
                                        "transcendentals per activation on a SIMD shared with the MFMA issue (DESIGN.md section 4)"),
        "lcao_coeff_mlp_contract_bf16x3": ("mfma", mlp_flop, PEAK_16BIT_MFMA_TFLOPS, "TFLOP/s",
                                           "algorithmic flop; three bf16 MFMAs are issued per product (3x this rate on the pipe)"),
        # reverse of the three-body term: Ag, g_tbw, factor, unit vector read; g_Ag, g_gram, g_u written
        "lcao_triplet_bwd_mfma": ("hbm", 8.0 * NL_ * V_ + 4.0 * V_ + 116.0, PEAK_HBM_GBS, "GB/s",
                                  "in- and out-edge lists are cut into 16-wide MFMA tiles (a degree-18 list fills them to 32 %): "
                                  "the kernel sits between its HBM and matrix ceilings (DESIGN.md section 4)"),
        # force training: levels 1 and 2 of the contraction + pair-table gradient, launched equally often (mean of the two)
        "lcao_aside_pairs": ("hbm", 0.5 * ((4.0 * V_ * (NL_ + 3) + 8.0 * D_ + 4.0 * NL_ * NL_) * 2 + 4.0 * V_ * 1 + 4.0 * D_
                                           + 4.0 * V_ * (NL_ + 2) + 4.0 * D_ + 4.0 * NL_ * NL_ + 4.0 * V_), PEAK_HBM_GBS, "GB/s",
                             "mean over its two derivative levels (gAg, gB, gate rows read; level 2 also writes d/dgAg)"),
        "lcao_pair_contract_bwd": ("hbm", 4.0 * NL_ * V_ + 8.0 * V_ + 40.0 + 8.0 * D_, PEAK_HBM_GBS, "GB/s", ""),
        "lcao_pair_contract": ("hbm", 4.0 * (NL_ + 1) * V_ + 40.0 + 4.0 * D_, PEAK_HBM_GBS, "GB/s", ""),
        "lcao_triplet_gemm": ("hbm", 4.0 * NL_ * V_ + 52.0 + 4.0 * V_, PEAK_HBM_GBS, "GB/s", ""),
    }
    kall = {}
    if rank == 0:
        try:
            with torch.no_grad(), ops.profile() as prof_all:                 # per-entry-point breakdown, separate pass
                for _ in range(args.steps):
                    model(batch, fused=fused, **mkw)
                kall = prof_all.summary()
        except Exception as e:  # noqa: BLE001
            secondary_error = f"{type(e).__name__}: {e}"
    # secondary modes are one-GPU measurements: reported by the N = 1 run only (at N > 1 the other ranks would sit in the final
    # barrier while rank 0 runs them)
    if rank == 0 and world == 1 and not args.no_secondary and args.precision == "fp32":
        try:
            if not with_forces:   # the force workloads already run on the pair table; no secondary modes there
                mat_ms, mat_err = timed(cst_mode="edge_tensor")              # cst materialised per edge as the reference holds it
                pair_ms, pair_err = timed(cst_mode="pair")                   # species-pair table
                bf_ms, bf_err = timed(precision="bf16x3")                    # BASELINE config 3
                h_ms, h_err = timed(precision="f16")                         # ... at its stated tolerance (1e-3)
                with torch.no_grad():
                    roof_bf = mode_roofline(lambda: model(batch, precision="bf16x3"), 3, mode_work, E)
                    roof_h = mode_roofline(lambda: model(batch, precision="f16"), 3, mode_work, E)
                bp_ms, bp_err = timed(precision="bf16x3", cst_mode="pair")
            if not with_forces:
                # the same forward replayed as ONE HIP graph (model/capture.py: the MD inner loop between neighbour-list
                # rebuilds — positions change, the edge list and its CSR do not)
                from lcaonet_amd.model import CapturedForward
                cap = CapturedForward(model, batch)
                cap_err = float((cap(batch["pos"]) - out).abs().max() / out.abs().max())
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    cap(batch["pos"])
                torch.cuda.synchronize()
                replay = (1e3 * (time.perf_counter() - t1) / args.steps, cap_err)
                del cap
            if not with_forces and args.workload == "cfg2":
                # rows f3 of SURVEY.md 8: energy + autograd forces (reverse sweep) and one full training step, same batch
                fm = build_model(dict(model_kw, regress_forces=True, direct_forces=False), 1234, dev)
                with torch.no_grad():
                    fm(batch)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(max(args.steps // 2, 2)):
                        fm(batch)
                    torch.cuda.synchronize()
                    forces_ms = 1e3 * (time.perf_counter() - t1) / max(args.steps // 2, 2)
                    roof_forces = mode_roofline(lambda: fm(batch), 3, mode_work, E)
                del fm
                tm = build_model(model_kw, 1234, dev).train()
                opt = torch.optim.Adam(tm.parameters(), lr=1e-4)
                target = torch.zeros(gen_kw["n_cells"], 1, device=dev)

                def train_step():
                    opt.zero_grad(set_to_none=True)
                    torch.nn.functional.l1_loss(tm(batch), target).backward()
                    opt.step()

                train_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(max(args.steps // 4, 2)):
                    train_step()
                torch.cuda.synchronize()
                train_ms = 1e3 * (time.perf_counter() - t1) / max(args.steps // 4, 2)
                roof_train = mode_roofline(train_step, 2, mode_work, E)
                del tm, opt
                # training ON FORCES: energy + exact autograd forces (create_graph) + backward through the force graph
                torch.cuda.reset_peak_memory_stats()
                ftm = build_model(dict(model_kw, regress_forces=True, direct_forces=False), 1234, dev).train()
                fopt = torch.optim.Adam(ftm.parameters(), lr=1e-4)
                f_target = torch.zeros(N, 3, device=dev)

                def force_train_step():
                    fopt.zero_grad(set_to_none=True)
                    e_pred, f_pred = ftm(batch)
                    (torch.nn.functional.l1_loss(e_pred, target) + torch.nn.functional.l1_loss(f_pred, f_target)).backward()
                    fopt.step()

                force_train_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    force_train_step()
                torch.cuda.synchronize()
                ftrain_ms = 1e3 * (time.perf_counter() - t1) / 2
                roof_ftrain = mode_roofline(force_train_step, 1, mode_work, E)
                ftrain_peak_gib = torch.cuda.max_memory_allocated() / 2**30
                # the same step with the graph rebuilt in the backward pass for two groups of crystals in turn
                fopt.zero_grad(set_to_none=True)
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats()
                ftm.force_training_chunks = 2
                force_train_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    force_train_step()
                torch.cuda.synchronize()
                ftrain2 = (1e3 * (time.perf_counter() - t1) / 2, torch.cuda.max_memory_allocated() / 2**30)
                del ftm, fopt
                torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            secondary_error = f"{type(e).__name__}: {e}"

    if rank == 0:
        D = model.elec_info.n_orb
        C, V, L, n_l = model.emb_size_coeff, model.emb_size_conv, model.n_interaction, model.n_l
        # dominant kernel: lcao_coeff_mlp_contract (f_coeffs on the matrix cores).  Algorithmic work per launch
        # (SURVEY.md §8d row "f_coeffs"): 2*D*(C*V + V*V) flop per edge; HBM 4*D*C read + 4*(n_l+1)*V write per edge.
        name = "lcao_coeff_mlp_contract"
        roof = None
        if with_forces:   # force workloads (cfg5): the dominant kernel of the energy + forces step, priced like the secondary modes
            with torch.no_grad():
                roof = mode_roofline(lambda: model(batch, fused=fused, **mkw), 2, mode_work, E)
        if name in ksum:
            calls, ms = ksum[name]
            per_launch_ms = ms / calls
            flop = 2.0 * D * (C * V + V * V) * E
            achieved = flop / (per_launch_ms * 1e-3) / 1e12
            roof = {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    "traffic": pmc_traffic(name) if args.workload == "cfg2" else None,
                    "avg_launch_ms": round(per_launch_ms, 4), "launches": calls,
                    "algorithmic_bytes_per_launch": (4 * D * C + 4 * D + 4 * (n_l + 1) * V) * E}
        # the HBM-bound group of the north star ("edge-basis + scatter path"): three-body kernel
        roof_hbm = None
        tk = "lcao_triplet_gemm" if "lcao_triplet_gemm" in ksum else "lcao_triplet_sum"
        if tk in ksum:
            calls, ms = ksum[tk]
            per_launch_ms = ms / calls
            byts = (4 * n_l * V + 40 + 12 + 4 * V) * E  # Ag rows + gram + unit vector read, tbw written
            gbs = byts / (per_launch_ms * 1e-3) / 1e9
            roof_hbm = {"kernel": tk, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                        "traffic": pmc_traffic("lcao_triplet_gemm") if args.workload == "cfg2" else None,
                        "avg_launch_ms": round(per_launch_ms, 4), "launches": calls,
                        "algorithmic_bytes_per_launch": byts}
        # every edge-level kernel against its own bound (per-launch averages of the separate fully bracketed pass, which
        # run ~3 % longer than in the timed region because of the event pairs): algorithmic work as in DESIGN.md section 4
        per_kernel = {}
        H = model.emb_size
        work = {  # entry point: (bound, algorithmic flop or bytes per edge)
            "lcao_coeff_mlp_contract": ("mfma", 2.0 * D * (C * V + V * V)),
            "lcao_edge_lcao_weight": ("mfma", 4.0 * V * V),
            "lcao_edge_node_message": ("mfma", 2.0 * V * V),
            "lcao_edge_node_message_agg": ("mfma", 2.0 * V * V),          # + the E -> N scatter, in the same kernel
            "lcao_triplet_gemm": ("hbm", 4 * n_l * V + 52 + 4 * V),
            "lcao_embed_coeffs": ("hbm", 4 * D * C),
            # pos -> dist, unit vectors (caller's order + p-space), radial basis: 3 int32 + shift read, 4 + 12 + 12 + 4 D written
            "lcao_edge_geometry_basis": ("hbm", 12 + 12 + 4 + 12 + 12 + 4 * D),
            "lcao_node_tables": ("mfma", 2.0 * H * 3 * V * N / E),
        }
        for kname, (bound, per_edge) in work.items():
            if kname in kall and kall[kname][0] > 0:
                calls, ms = kall[kname]
                if kname == "lcao_embed_coeffs":
                    calls, ms = args.steps, ms          # the pair-table launches of other modes are tiny; one real launch per step
                rate = per_edge * E / (ms / calls * 1e-3)
                peak = PEAK_F32_MFMA_TFLOPS * 1e12 if bound == "mfma" else PEAK_HBM_GBS * 1e9
                per_kernel[kname] = {"bound": bound, "avg_launch_ms": round(ms / calls, 4),
                                     "achieved": round(rate / (1e12 if bound == "mfma" else 1e9), 1),
                                     "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": round(rate / peak, 4)}
        shape = (f"{gen_kw['n_mol']} random molecules of 3..29 atoms" if gen_kw.get("molecules") else
                 f"{gen_kw['n_cells']} jittered {gen_kw['n_side']}^3 simple-cubic crystals")
        line = {
            "metric": "edges/sec forward (LCAOConv stack), 1M-edge synthetic batch",
            "value": e_total * args.steps / elapsed_max, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32"}.get(args.precision, args.precision), "data": "synthetic",
            "config": {"workload": f"{args.workload}: {shape} per GPU (N={N}, E={E} per GPU), graph cutoff "
                                   f"{gen_kw['cutoff_graph']}, "
                                   f"LCAONet(emb_size={model.emb_size}, coeff={C}, conv={V}, n_interaction={L}, D={D}, "
                                   f"cutoff={model.cutoff}), random-init weights, eval, whole forward; general per-edge "
                                   "coefficient MLP (every (edge, orbital) row through both f_coeffs layers), its input rows "
                                   "gathered from the (z_s, z_t) table inside the kernel instead of a materialised (E,D,C) tensor"
                                   + (" + forces = -dE/dpos (reverse sweep)" if with_forces else ""),
                       "edges_per_gpu": E, "nodes_per_gpu": N, "sharding": "whole crystals per rank, no collective"},
            "roofline": roof, "roofline_hbm": roof_hbm, "traffic_source": pmc_source(),
            "roofline_forward": roofline_forward(model, N, E, elapsed_max / args.steps),
            "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(kall.items(), key=lambda kv: -kv[1][1])},
            "kernel_ms_per_step_note": "separate pass after the timed region with every C-ABI call bracketed by HIP events",
            "kernel_rooflines": per_kernel,
        }
        if mat_ms is not None:
            line["materialized_cst_mode"] = {
                "note": "NOT the headline: cst_mode='edge_tensor' — the (E, D, C) coefficient tensor written by lcao_embed_coeffs "
                        "and re-read by every layer, as the reference holds it (the headline gathers the same rows from the "
                        "species-pair table inside lcao_coeff_mlp_contract; same flops, same outputs)",
                "ms_per_step_rank0": round(mat_ms, 3), "edges_per_s_rank0": E / (mat_ms * 1e-3),
                "max_rel_diff_vs_headline_output": mat_err}
        if pair_ms is not None:
            line["pair_table_mode"] = {
                "note": "NOT the headline: LCAONet.forward(cst_mode='pair') — f_coeffs evaluated on the (z_s,z_t) species-pair "
                        "table (SURVEY.md 8f-2) instead of per edge; same outputs", "ms_per_step_rank0": round(pair_ms, 3),
                "edges_per_s_rank0": E / (pair_ms * 1e-3), "max_rel_diff_vs_headline_output": pair_err}
        if replay is not None:
            line["hip_graph_replay_mode"] = {
                "note": "NOT the headline: the whole forward captured once (CapturedForward) and replayed with the positions "
                        "copied in — fixed edge list, so lcao_build_csr (0.19 ms) is outside the replay; no launch gaps",
                "ms_per_step_rank0": round(replay[0], 3), "edges_per_s_rank0": E / (replay[0] * 1e-3),
                "max_rel_diff_vs_headline_output": replay[1]}
        if secondary_error is not None:
            line["secondary_error"] = secondary_error
        if bf_ms is not None:
            line["bf16x3_mode"] = {
            "note": "NOT the headline: BASELINE config 3 — the per-edge GEMMs (coefficient MLP, f_three, basis_weight, f_node.2) on "
                    "bf16 MFMA, each fp32 operand split into two bf16 terms (3 MFMAs per product, fp32 accumulate); "
                    "general per-edge cst", "dtype": "bf16x3",
            "ms_per_step_rank0": round(bf_ms, 3), "edges_per_s_rank0": E / (bf_ms * 1e-3),
            "max_rel_diff_vs_fp32_output": bf_err, "roofline": roof_bf,
            "with_pair_table": None if bp_ms is None else {
                "ms_per_step_rank0": round(bp_ms, 3), "edges_per_s_rank0": E / (bp_ms * 1e-3),
                "max_rel_diff_vs_fp32_output": bp_err}}
        if h_ms is not None:
            line["f16_mode"] = {
                "note": "NOT the headline: BASELINE config 3 at its stated tolerance (1e-3) — the per-edge GEMMs with operands rounded "
                        "once to fp16, ONE v_mfma_f32_16x16x32_f16 per product, fp32 accumulation; general per-edge cst",
                "dtype": "f16", "ms_per_step_rank0": round(h_ms, 3), "edges_per_s_rank0": E / (h_ms * 1e-3),
                "max_rel_diff_vs_fp32_output": h_err, "roofline": roof_h}
        if forces_ms is not None:
            line["autograd_forces_mode"] = {
                "note": "NOT the headline: regress_forces=True, direct_forces=False — energy and F = -dE/dpos by the HIP reverse "
                        "sweep (species-pair forward)", "ms_per_step_rank0": round(forces_ms, 3),
                "edges_per_s_rank0": E / (forces_ms * 1e-3), "roofline": roof_forces}
        if train_ms is not None:
            line["training_step_mode"] = {
                "note": "NOT the headline: train mode, forward + HIP backward (all parameter gradients) + Adam step",
                "ms_per_step_rank0": round(train_ms, 3), "edges_per_s_rank0": E / (train_ms * 1e-3), "roofline": roof_train}
        if ftrain_ms is not None:
            line["force_training_step_mode"] = {
                "note": "NOT the headline: train mode, regress_forces=True, direct_forces=False — energy, exact autograd forces "
                        "(create_graph), L1 loss on both, backward through the force graph (twice-differentiable HIP "
                        "primitives, model/double.py) + Adam step",
                "ms_per_step_rank0": round(ftrain_ms, 3), "edges_per_s_rank0": E / (ftrain_ms * 1e-3),
                "peak_memory_GiB": round(ftrain_peak_gib, 1), "roofline": roof_ftrain,
                "in_two_groups_of_graphs": {
                    "note": "model.force_training_chunks = 2: forward by the fused graph-less sweep, the twice-differentiable "
                            "graph rebuilt in the backward pass for one half of the crystals at a time (same gradients)",
                    "ms_per_step_rank0": round(ftrain2[0], 3), "peak_memory_GiB": round(ftrain2[1], 1)}}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
