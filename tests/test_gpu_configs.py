"""GPU parity of BASELINE configs 4 and 5 against the CPU oracle ("configs_untested"), through the C ABI.

cfg4: one periodic crystal, cutoff 6.0 (graph cutoff 0.99 x), 32 nearest neighbours, H = C = V = 128, L = 3 — unsharded
      and cut into 2 / 4 slabs with the real halo exchange between the emulated ranks (the ranks run in lock step on the
      one GPU; what crosses ranks is exactly what shard.start_exchange moves over RCCL, computed by shard.local_exchange).
cfg5: QM9-shaped molecules, max_z = 10, H = 256, L = 6, energy + forces = -dE/dpos (regress_forces, direct_forces=False)
      against torch.autograd through the fp64 oracle.
The oracle's literal dataflow holds a (T, D, V) tensor, so the sizes are what it finishes in seconds."""
import pytest
import torch

from conftest import relerr
from oracle import restatement as O
from test_gpu_parity import _ocfg, _oracle_run, _random_model

pytestmark = pytest.mark.gpu
TOL = 1e-5

CFG4 = dict(emb_size=128, emb_size_coeff=128, emb_size_conv=128, n_interaction=3, cutoff=6.0)


def _lockstep_sharded_forward(model, plans, with_forces=False):
    """LCAONet.forward_sharded for every rank of `plans` at once: layer by layer, with the halo rows of layer i's output
    moved between the ranks' buffers exactly as the all-to-all does."""
    from lcaonet_amd import shard
    from lcaonet_amd.model.sharded import ShardedForward
    sfs = [ShardedForward(model, p) for p in plans]
    xs = [sf.layer(0, None) for sf in sfs]
    n_exchanges = 0
    for i in range(1, model.n_interaction):
        ghosts = shard.local_exchange(plans, xs)
        n_exchanges += 1
        xs = [sf.layer(i, lambda g=g: g) for sf, g in zip(sfs, ghosts)]
    tot = sum(sf.head_partial() for sf in sfs)                         # the all-reduce
    outs = [sf.finish(tot) for sf in sfs]
    forces = None
    if with_forces:
        ghosts = shard.local_exchange(plans, xs)
        forces = [sf.direct_forces(g) for sf, g in zip(sfs, ghosts)]
    return xs, outs, forces, n_exchanges


def test_cfg4_crystal_vs_oracle_unsharded_and_slabs(dev):
    """BASELINE config 4 at oracle size: an 8^3 = 512-atom jittered crystal, cutoff 6.0 / graph 5.94, 32 nearest
    neighbours, the full-width model.  Unsharded HIP vs the oracle on every layer's state and the energy; then 2 and 4
    slabs, each compared with the ORACLE (not with the unsharded HIP run)."""
    from lcaonet_amd import shard
    from lcaonet_amd.data.synthetic import jittered_cubic_crystal
    m, sd = _random_model(dev, 404, **CFG4)
    g = jittered_cubic_crystal(8, cutoff_graph=0.99 * 6.0, max_neighbors=32, device=dev)
    N, E = g["z"].numel(), g["edge_index"].shape[1]
    assert N == 512 and 24 * N < E <= 32 * N
    gc = {k: v.cpu() for k, v in g.to_dict().items() if k in ("z", "pos", "edge_index", "edge_shift", "lattice", "batch")}
    want, wtr = _oracle_run(sd, CFG4, gc)
    tr = {}
    got = m(g, trace=tr)
    for i in (1, 2, 3):
        assert relerr(tr[f"x{i}"], wtr[f"x{i}"]) <= TOL, i
    assert relerr(got, want) <= TOL
    for world in (2, 4):
        plans = shard.make_plans(g, world)
        assert all(p.n_ghost > 0 for p in plans) and sum(p.e_own for p in plans) == E
        xs, outs, _, n_ex = _lockstep_sharded_forward(m, plans)
        assert n_ex == CFG4["n_interaction"] - 1
        for p, x_own, out in zip(plans, xs, outs):
            assert relerr(x_own, wtr["x3"][p.owned_global.cpu()]) <= TOL, (world, p.rank)
            assert relerr(out, want) <= TOL, (world, p.rank)


def test_cfg4_per_rank_slabs_equal_the_unsharded_crystal(dev):
    """The route of `bench.py --workload cfg4 --gpus N`: every rank builds ONLY its slab (window neighbour search, no
    global graph) and its plan; 2 slabs of a 24^3 crystal in lock step equal the unsharded forward of the same crystal."""
    from lcaonet_amd import shard
    from lcaonet_amd.data.synthetic import jittered_cubic_crystal, jittered_cubic_crystal_slab
    kw = dict(emb_size=64, emb_size_coeff=64, emb_size_conv=64, n_interaction=3, cutoff=6.0)
    m, _ = _random_model(dev, 405, **kw)
    world, n_side = 2, 24                       # 12 owned layers + 2 x 5 halo layers < 24: the window path
    g = jittered_cubic_crystal(n_side, cutoff_graph=0.99 * 6.0, max_neighbors=32, device=dev)
    tr = {}
    full = m(g, trace=tr)
    parts = [jittered_cubic_crystal_slab(n_side, r, world, cutoff_graph=0.99 * 6.0, max_neighbors=32, device=dev)
             for r in range(world)]
    assert sum(int((lg["edge_index"][0] < n_own).sum()) for lg, _, n_own, _ in parts) == g["edge_index"].shape[1]
    # plan_from_local is collective; with one process the negotiation is done by hand: rank q's requests to rank r
    plans = _plans_from_parts(parts, world)
    for p in plans:
        assert sum(1 for s in p.send_splits if s) == 1 and p.n_ghost < p.n_own      # two slabs: one neighbour each
    xs, outs, _, _ = _lockstep_sharded_forward(m, plans)
    for p, x_own, out in zip(plans, xs, outs):
        assert relerr(x_own, tr["x3"][p.owned_global]) <= 5e-6, p.rank
        assert relerr(out, full) <= 5e-6


def _plans_from_parts(parts, world):
    """shard.plan_from_local for every rank in ONE process: the two plan-time all-to-alls are served from the other
    ranks' inputs by monkeypatching shard._a2a (counts, then ids)."""
    from lcaonet_amd import shard
    ghost_lists = []
    for r, (lg, gid, n_own, owner_of) in enumerate(parts):
        gg = gid[n_own:]
        ow = owner_of(gg)
        o = torch.argsort(ow * (int(gid.max()) + 1) + gg)
        ghost_lists.append((gg[o], ow[o]))
    plans = []
    real = shard._a2a
    try:
        for r, (lg, gid, n_own, owner_of) in enumerate(parts):
            def fake(out, inp, out_splits, in_splits, group=None, async_op=False, r=r):
                if out_splits is None:      # counts: what every rank q wants from r
                    out.copy_(torch.stack([(ghost_lists[q][1] == r).sum() for q in range(world)]))
                else:                       # ids: rank q's ghosts owned by r, in q's order
                    out.copy_(torch.cat([ghost_lists[q][0][ghost_lists[q][1] == r] for q in range(world)]))
                return None
            shard._a2a = fake
            plans.append(shard.plan_from_local(r, world, lg, gid, n_own, owner_of))
    finally:
        shard._a2a = real
    return plans


def test_cfg5_molecules_autograd_forces_vs_oracle(dev):
    """BASELINE config 5 at oracle size: QM9-shaped molecules (3..29 atoms, z in {1,6,7,8,9}), max_z = 10 (D = 3),
    hidden 256, 6 interactions, forces via autograd (regress_forces=True, direct_forces=False; lcaonet.py:310-317):
    energy <= 1e-5 and forces <= 5e-5 of the largest component against torch.autograd through the fp64 oracle."""
    from lcaonet_amd.data.synthetic import random_molecule_batch
    kw = dict(emb_size=256, emb_size_coeff=128, emb_size_conv=128, n_interaction=6, cutoff=6.0, max_z=10,
              regress_forces=True, direct_forces=False)
    m, sd = _random_model(dev, 56, **kw)
    g = random_molecule_batch(24, 3, 29, cutoff_graph=0.99 * 6.0, seed=4, device=dev)
    assert m.elec_info.n_orb == 3 and m.n_l == 2
    gc = {k: v.cpu() for k, v in g.to_dict().items() if k in ("z", "pos", "edge_index", "edge_shift", "lattice", "batch")}
    want, wforce = O.forward(sd, _ocfg({**kw, "rbf_type": "hydrogen"}), gc, dtype=torch.float64)
    out, force = m(g)
    assert force.shape == wforce.shape
    assert relerr(out, want.detach()) <= TOL
    assert relerr(force, wforce.detach()) <= 5 * TOL
    assert float(force.sum(0).abs().max()) <= 1e-4 * float(force.abs().max())      # momentum conservation


@pytest.mark.parametrize("V", [64, 32])
def test_dense_cluster_degree_89_forces_vs_oracle(dev, V):
    """A degree no other case reaches: 90 atoms inside a 2.8 A cube, every pair within the cutoff (89 neighbours each,
    8 010 edges, 705 k triplets).  Exercises the second 64-wide in-edge block of lcao_triplet_gemm, the chunked out-lists of the
    three-phase triplet backward (> 80 out-edges at n_l = 3: later chunks add into rows already written) and six-tile nodes
    in every per-node kernel.  Energy and autograd forces against the fp64 oracle."""
    from lcaonet_amd.data.synthetic import random_molecules
    kw = dict(emb_size=32, emb_size_coeff=V, emb_size_conv=V, n_interaction=2, cutoff=5.0, regress_forces=True,
              direct_forces=False)
    m, sd = _random_model(dev, 91, **kw)
    g = random_molecules(1, 90, 2.8, 0.99 * 5.0, seed=12, device=dev)
    assert g["edge_index"].shape[1] == 90 * 89 and m.n_l == 3
    gc = {k: v.cpu() for k, v in g.to_dict().items() if k in ("z", "pos", "edge_index", "edge_shift", "lattice", "batch")}
    want, wforce = O.forward(sd, _ocfg({**kw, "rbf_type": "hydrogen"}), gc, dtype=torch.float64)
    out, force = m(g)
    assert relerr(out, want.detach()) <= TOL
    assert relerr(force, wforce.detach()) <= 5 * TOL
    fm, _ = _random_model(dev, 91, **{**kw, "regress_forces": False})          # the forward-only pipeline on the same graph
    assert relerr(fm(g), want.detach()) <= TOL


def test_cfg5_molecules_direct_forces_vs_oracle(dev):
    """The direct-force head on the same shape (the reference's default for regress_forces)."""
    from lcaonet_amd.data.synthetic import random_molecule_batch
    kw = dict(emb_size=256, emb_size_coeff=128, emb_size_conv=128, n_interaction=6, cutoff=6.0, max_z=10,
              regress_forces=True, direct_forces=True)
    m, sd = _random_model(dev, 55, **kw)
    g = random_molecule_batch(40, 3, 29, cutoff_graph=0.99 * 6.0, seed=3, device=dev)
    gc = {k: v.cpu() for k, v in g.to_dict().items() if k in ("z", "pos", "edge_index", "edge_shift", "lattice", "batch")}
    want, wforce = _oracle_run(sd, kw, gc)[0]
    got, force = m(g)
    assert relerr(got, want) <= TOL and relerr(force, wforce) <= TOL
    got_p, force_p = m(g, cst_mode="pair")
    assert relerr(got_p, want) <= TOL and relerr(force_p, wforce) <= TOL


@pytest.mark.parametrize("rc", [5.0, 6.0])
@pytest.mark.parametrize("cut", ["envelope", "polynomial", "cosine"])
def test_radial_basis_flat_tolerance_inside_095_rc(golden_loader, dev, rc, cut):
    """a5 on r <= 0.95 rc at a FLAT 1e-5 per orbital against the reference's fp32 tables: the relaxed bound of
    test_basis_kernels_vs_reference_tables (needed only inside the envelope-cancellation shell next to rc) cannot hide
    a regression in any of the 18 orbitals (4f / 5f / 6d included)."""
    from lcaonet_amd.atomistic.info import ElecInfo
    from lcaonet_amd.nn.rbf import HydrogenRadialBasis
    from lcaonet_amd.utils.resolve import cutoffnet_resolver
    z = golden_loader("basis_tables").z
    r = torch.from_numpy(z[f"r_rc{rc:g}"]).float()
    keep = r <= 0.95 * rc
    ei = ElecInfo(96, None, None, 1)
    rb = HydrogenRadialBasis(rc, ei, cutoffnet_resolver(cut, cutoff=rc)).to(dev)(r.to(dev)).cpu()
    ref = torch.from_numpy(z[f"hydrogen_{cut}_rc{rc:g}_f32"])
    assert ei.n_orb == 18 and int(keep.sum()) > 300          # 331 of the 384 grid points
    for d in range(ei.n_orb):
        assert relerr(rb[keep, d], ref[keep, d]) <= TOL, (d, relerr(rb[keep, d], ref[keep, d]))


def test_basis_options_vs_reference_tables(golden_loader, dev):
    """Operator-level options LCAONet itself never sets: HydrogenRadialBasis(integral_norm=True) (rbf.py:107-127: each
    orbital normalised over the cutoff sphere) and EnvelopeCutoff(p != 5) (cutoff.py:46-67), against tables dumped from the
    reference; flat 1e-5 per orbital on the whole grid."""
    from lcaonet_amd.atomistic.info import ElecInfo
    from lcaonet_amd.nn.cutoff import EnvelopeCutoff
    from lcaonet_amd.nn.rbf import HydrogenRadialBasis
    z = golden_loader("basis_options").z
    r = torch.from_numpy(z["r"]).float().to(dev)
    ei = ElecInfo(96, None, None, 1)
    rb = HydrogenRadialBasis(5.0, ei, EnvelopeCutoff(5.0), integral_norm=True).to(dev)(r)
    ref = torch.from_numpy(z["hydrogen_integral_norm_f32"])
    for d in range(ei.n_orb):
        assert relerr(rb[:, d], ref[:, d]) <= TOL, ("integral_norm", d, relerr(rb[:, d], ref[:, d]))
    for p in (3, 7):
        cn = EnvelopeCutoff(5.0, p)
        assert relerr(cn(r), z[f"envelope_p{p}_f32"]) <= TOL
        rb = HydrogenRadialBasis(5.0, ei, cn).to(dev)(r)
        ref = torch.from_numpy(z[f"hydrogen_envelope_p{p}_f32"])
        ref64 = torch.from_numpy(z[f"hydrogen_envelope_p{p}_f64"])
        for d in range(ei.n_orb):
            # the kernel's factored envelope is exact to 3e-7, so the distance to the reference's fp32 table IS the reference's
            # fp32-vs-fp64 noise — 6.2e-6 at p = 3, 1.16e-5 at p = 7 (terms of 63 in the expanded form), orbital 4f
            noise = relerr(ref[:, d], ref64[:, d])
            assert relerr(rb[:, d], ref[:, d]) <= max(TOL, 1.02 * noise), (p, d, relerr(rb[:, d], ref[:, d]), noise)
            assert relerr(rb[:, d], ref64[:, d]) <= max(0.5 * noise, 1e-6), (p, d)


def _sweep_cases():
    import random
    rnd = random.Random(20261004)
    cases = []
    for i in range(24):
        V = rnd.choice([16, 24, 32, 48, 64, 96, 128])
        C = rnd.choice([V, V, 32, 64, 128])
        H = rnd.choice([16, 40, 64, 128, 256])
        kw = dict(emb_size=H, emb_size_coeff=C, emb_size_conv=V, n_interaction=rnd.choice([1, 2, 3]),
                  cutoff=rnd.choice([4.0, 5.0]), max_z=rnd.choice([10, 36, 84]), add_valence=rnd.random() < 0.3,
                  n_per_orb=rnd.choice([1, 1, 2]), cutoff_net=rnd.choice(["envelope", "polynomial", "cosine"]),
                  rbf_type=rnd.choice(["hydrogen", "hydrogen", "sphericalbessel"]), elec_to_node=rnd.random() < 0.8,
                  is_extensive=rnd.random() < 0.7, out_size=rnd.choice([1, 2]), extend_orb=rnd.random() < 0.2)
        cases.append((i, kw, rnd.choice(["cells", "cells", "molecules"])))
    return cases


@pytest.mark.parametrize("idx,kw,kind", _sweep_cases())
def test_random_configuration_sweep_vs_oracle(dev, idx, kw, kind):
    """24 seeded random constructor / width / graph combinations (fused and unfused kernel paths, node tables with and
    without the folded launch, in-kernel coefficient-row gather, fused scatter, valence, n_per_orb, all cutoffs and bases),
    default forward and cst_mode='pair' against the oracle at 1e-5."""
    from lcaonet_amd.data.synthetic import random_molecule_batch, random_periodic_cells
    m, sd = _random_model(dev, 9000 + idx, **kw)
    if kind == "cells":
        g = random_periodic_cells(3, 10 + idx % 5, 6.0, 0.99 * kw["cutoff"], seed=idx, max_z=kw["max_z"]).to(dev)
    else:
        species = (1, 6, 7, 8, 9) if kw["max_z"] >= 9 else (1, 2)
        g = random_molecule_batch(5, 3, 9, cutoff_graph=0.99 * kw["cutoff"], seed=idx, species=species, device=dev)
    gc = {k: v.cpu() for k, v in g.to_dict().items() if k in ("z", "pos", "edge_index", "edge_shift", "lattice", "batch")}
    want, wtr = _oracle_run(sd, kw, gc)
    tr = {}
    got = m(g, trace=tr)
    L = kw["n_interaction"]
    assert relerr(tr[f"x{L}"], wtr[f"x{L}"]) <= TOL, (kw, relerr(tr[f"x{L}"], wtr[f"x{L}"]))
    assert relerr(got, want) <= TOL, (kw, relerr(got, want))
    with torch.no_grad():
        assert relerr(m(g), want) <= TOL                       # no trace: the fully fused route
        assert relerr(m(g, cst_mode="pair"), want) <= TOL
