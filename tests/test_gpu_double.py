"""Exact double backward (lcaonet_amd/model/double.py): training on autograd forces as the reference does it
(lcaonet/model/lcaonet.py:310-317, torch.autograd.grad(create_graph=True); tests/model/test_lcaonet.py:219-243) — against
torch.autograd through the fp64 oracle.  No step size anywhere: the HIP primitives are linear / bilinear and their
backward passes are built from the same family, so autograd differentiates the force computation itself."""
import pytest
import torch

from conftest import relerr
from oracle import restatement as O
from test_gpu_parity import _ocfg, _random_model

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _oracle_force_loss_grads(sd, kw, g_cpu, training, w_e=0.1):
    cfg = _ocfg({**kw, "rbf_type": kw.get("rbf_type", "hydrogen"), "training": training})
    sdg = {k: (v.clone().double().requires_grad_("running_" not in k) if v.is_floating_point() else v.clone())
           for k, v in sd.items()}
    e_ref, f_ref = O.forward(sdg, cfg, g_cpu, dtype=torch.float64)
    t_e, t_f = torch.ones_like(e_ref), torch.full_like(f_ref, 0.05)
    (w_e * ((e_ref - t_e) ** 2).mean() + ((f_ref - t_f) ** 2).mean()).backward()
    return e_ref.detach(), f_ref.detach(), t_e, t_f, {k: v.grad for k, v in sdg.items() if torch.is_tensor(v) and v.grad is not None}


def _check_param_grads(m, want, tol):
    checked, worst = 0, 0.0
    for name, p in m.named_parameters():
        ref = want.get(name)
        if ref is None or float(ref.abs().max()) < 1e-12:
            continue
        assert p.grad is not None, name
        got, ref = (p.grad[1:], ref[1:]) if "e_embeds" in name else (p.grad, ref)     # padding row: see test_gpu_parity
        if float(ref.abs().max()) < 1e-12:
            continue
        err = relerr(got, ref)
        worst = max(worst, err)
        assert err <= tol, (name, err)
        checked += 1
    return checked, worst


@pytest.mark.parametrize("case", ["v64_eval", "v64_train", "v128_out2_mean", "v48_valence_poly"])
def test_exact_force_training_vs_oracle_double_backward(dev, case):
    """Parameter gradients of an energy + force loss, forces = -dE/dpos with create_graph=True: every tensor within 2e-5
    of the fp64 oracle's double backward (the central-difference cross-check needed 2e-4)."""
    from lcaonet_amd.data.synthetic import random_periodic_cells
    kw = dict(emb_size=32, emb_size_coeff=64, emb_size_conv=64, n_interaction=2, cutoff=5.0, regress_forces=True,
              direct_forces=False)
    training = case == "v64_train"
    if case == "v128_out2_mean":
        kw.update(emb_size_coeff=128, emb_size_conv=128, out_size=2, is_extensive=False)
    elif case == "v48_valence_poly":
        kw.update(emb_size_coeff=24, emb_size_conv=48, add_valence=True, cutoff_net="polynomial", n_interaction=3)
    m, sd = _random_model(dev, 777, grad=True, **kw)
    m.train(training)
    g = random_periodic_cells(2, 10, 6.0, 0.99 * 5.0, seed=31)
    e_ref, f_ref, t_e, t_f, want = _oracle_force_loss_grads(sd, kw, g.to_dict(), training)
    assert getattr(m, "force_training", "exact") == "exact"
    e, f = m(g.to(dev))                                   # autograd on + parameters require grad: the differentiable path
    assert e.requires_grad and f.requires_grad and f.grad_fn is not None
    assert relerr(e.detach(), e_ref) <= 2e-5 and relerr(f.detach(), f_ref) <= 5 * TOL and f.shape == f_ref.shape
    loss = 0.1 * ((e - t_e.to(dev).float()) ** 2).mean() + ((f - t_f.to(dev).float()) ** 2).mean()
    loss.backward()
    checked, worst = _check_param_grads(m, want, 2e-5)
    assert checked > 25, checked
    print(f"exact force training [{case}]: worst parameter-gradient deviation {worst:.2e} over {checked} tensors")
    if training:
        assert int(m.emb_layer.coeff_embed.bn.num_batches_tracked) == 1


@pytest.mark.parametrize("V,n_int,training", [(32, 2, False), (64, 3, True), (128, 2, False)])
def test_fused_levels_equal_their_composition_from_primitives(dev, V, n_int, training):
    """The fused maps with hand-written derivative levels (double._ASideFn ...) against the same function assembled from
    the smaller primitives (double.FUSED_LEVELS = False): energy, forces and every parameter gradient of an energy +
    force loss (second derivatives), and the plain first-order gradients of an energy-only loss."""
    from lcaonet_amd.data.synthetic import random_periodic_cells
    from lcaonet_amd.model import double
    kw = dict(emb_size=32, emb_size_coeff=32, emb_size_conv=V, n_interaction=n_int, cutoff=5.0, regress_forces=True,
              direct_forces=False)
    m, _ = _random_model(dev, 4242, grad=True, **kw)
    m.train(training)
    g = random_periodic_cells(3, 9, 6.0, 0.99 * 5.0, seed=77).to(dev)

    def run(fused, energy_only):
        double.FUSED_LEVELS = fused
        try:
            m.zero_grad(set_to_none=True)
            e, f = m(g)
            loss = (e ** 2).mean() if energy_only else 0.1 * (e ** 2).mean() + ((f - 0.05) ** 2).mean()
            loss.backward()
            return e.detach(), f.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            double.FUSED_LEVELS = True

    for energy_only in (False, True):
        e1, f1, g1 = run(True, energy_only)
        e0, f0, g0 = run(False, energy_only)
        assert relerr(e1, e0) <= 2e-6 and relerr(f1, f0) <= 1e-5
        assert g1.keys() == g0.keys() and len(g1) > 25
        worst = max(relerr(g1[n], g0[n]) for n in g0 if float(g0[n].abs().max()) > 1e-12)
        assert worst <= 2e-5, worst


@pytest.mark.parametrize("training,chunks", [(True, 2), (False, 3), (True, 7)])
def test_force_training_in_groups_of_graphs_gives_the_same_gradients(dev, training, chunks):
    """model.force_training_chunks = k: forward by the fused graph-less sweep, the twice-differentiable graph rebuilt in the
    backward pass for k groups of graphs in turn (double._RecomputedForcesFn) — energy, forces, every parameter gradient
    and the BatchNorm statistics equal those of the one-graph path; edges of the batch in shuffled order."""
    from lcaonet_amd.data.synthetic import random_periodic_cells
    kw = dict(emb_size=32, emb_size_coeff=32, emb_size_conv=64, n_interaction=2, cutoff=5.0, regress_forces=True,
              direct_forces=False)
    m, _ = _random_model(dev, 99, grad=True, **kw)
    m.train(training)
    g = random_periodic_cells(5, 8, 6.0, 0.99 * 5.0, seed=5).to(dev)
    perm = torch.randperm(g["edge_index"].shape[1], generator=torch.Generator().manual_seed(3)).to(dev)
    g["edge_index"], g["edge_shift"] = g["edge_index"][:, perm].contiguous(), g["edge_shift"][perm].contiguous()
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def run(k):
        m.load_state_dict(state)
        m.force_training_chunks = k
        m.zero_grad(set_to_none=True)
        e, f = m(g)
        assert e.requires_grad and f.requires_grad
        (0.1 * (e ** 2).mean() + ((f - 0.05) ** 2).mean()).backward()
        stats = {n: b.clone() for n, b in m.named_buffers() if "running" in n}
        return e.detach(), f.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, stats

    try:
        e0, f0, g0, s0 = run(1)
        e1, f1, g1, s1 = run(chunks)
    finally:
        m.force_training_chunks = 1
    assert relerr(e1, e0) <= 5e-6 and relerr(f1, f0) <= 2e-5
    assert g1.keys() == g0.keys() and len(g1) > 25
    worst = max(relerr(g1[n], g0[n]) for n in g0 if float(g0[n].abs().max()) > 1e-12)
    assert worst <= 2e-5, worst
    assert all(torch.equal(s0[n], s1[n]) for n in s0)


def test_eval_mode_forces_are_differentiable_and_match_the_inference_sweep(dev):
    """Eval mode with autograd on returns forces WITH a graph (the reference's create_graph=True), equal to the fast
    graph-less inference sweep (frozen parameters / torch.no_grad()); the force graph reaches pos-independent parameters
    only through the second derivative, and a second call is bitwise reproducible."""
    from lcaonet_amd.data.synthetic import random_molecule_batch
    kw = dict(emb_size=64, emb_size_coeff=32, emb_size_conv=32, n_interaction=2, cutoff=6.0, max_z=10,
              regress_forces=True, direct_forces=False)
    m, _ = _random_model(dev, 31, grad=True, **kw)
    g = random_molecule_batch(5, 3, 9, cutoff_graph=0.99 * 6.0, seed=8, device=dev)
    e, f = m(g)
    assert f.requires_grad and e.requires_grad and not m.training
    with torch.no_grad():
        e0, f0 = m(g)
    assert not f0.requires_grad
    assert relerr(e.detach(), e0) <= 5e-6 and relerr(f.detach(), f0) <= 2e-5
    gW = torch.autograd.grad((f ** 2).sum(), m.out_layer.out_lin[4].weight, retain_graph=True)[0]
    assert float(gW.abs().max()) > 0                     # d|F|^2 / d(last head weight): pure second-derivative path
    e2, f2 = m(g)
    assert torch.equal(f2.detach(), f.detach())


@pytest.mark.parametrize("activation", ["ShiftedSoftplus", "Tanh", "ELU"])
def test_other_activations_in_the_differentiable_paths(dev, activation):
    """lcaonet/nn/activation.py / utils/resolve.py:66: activations other than SiLU in training and in autograd forces
    (the generic twice-differentiable graph serves them; the fused sweeps are SiLU-only)."""
    from lcaonet_amd.data.synthetic import random_periodic_cells
    kw = dict(emb_size=32, emb_size_coeff=32, emb_size_conv=32, n_interaction=2, cutoff=4.5, activation=activation)
    g = random_periodic_cells(2, 9, 6.0, 0.99 * 4.5, seed=13)
    # energies + parameter gradients
    m, sd = _random_model(dev, 801, grad=True, **kw)
    okw = {k: v for k, v in kw.items() if k != "activation"}
    sdg = {k: (v.clone().double().requires_grad_("running_" not in k) if v.is_floating_point() else v.clone())
           for k, v in sd.items()}
    want = O.forward(sdg, _ocfg({**okw, "rbf_type": "hydrogen", "extra": {"activation": activation}}), g.to_dict(),
                     dtype=torch.float64)
    want.sum().backward()
    out = m(g.to(dev))
    assert out.requires_grad and relerr(out.detach(), want.detach()) <= 2e-5
    out.sum().backward()
    checked, worst = _check_param_grads(m, {k: v.grad for k, v in sdg.items() if torch.is_tensor(v) and v.grad is not None}, 2e-4)
    assert checked > 20
    # autograd forces at inference (no graph wanted): served by the same graph, detached
    kwf = dict(kw, regress_forces=True, direct_forces=False)
    mf, sdf = _random_model(dev, 802, **kwf)
    okwf = {k: v for k, v in kwf.items() if k != "activation"}
    wout, wforce = O.forward(sdf, _ocfg({**okwf, "rbf_type": "hydrogen", "extra": {"activation": activation}}), g.to_dict(),
                             dtype=torch.float64)
    o, f = mf(g.to(dev))
    assert not f.requires_grad
    assert relerr(o, wout.detach()) <= 2e-5 and relerr(f, wforce.detach()) <= 5 * TOL


def test_trip_bilinear_family_is_consistent(dev):
    """csrc/lcao_double.hip: spmm / spmm_t / sddmm against dense per-node einsums, and <spmm(w, A), G> = <w, sddmm(G, A)>
    = <A, spmm_t(w, G)> (each op is the adjoint of the others in one argument)."""
    from lcaonet_amd import ops
    from lcaonet_amd.data.synthetic import random_periodic_cells
    from lcaonet_amd.model import LCAONet
    from lcaonet_amd.model.double import _Geo
    from lcaonet_amd.data.synthetic import random_molecules
    g_small = random_periodic_cells(2, 12, 6.0, 0.99 * 4.5, seed=3).to(dev)
    g_dense = random_molecules(1, 70, 2.8, 0.99 * 5.0, seed=2, device=dev)        # 69 neighbours each: five 16-wide tiles
    for V, n_l, max_z, g in ((48, 3, 36, g_small), (128, 2, 10, g_small), (200, 4, 96, g_small), (128, 4, 36, g_small),
                             (50, 3, 36, g_small), (32, 1, 36, g_small), (128, 3, 36, g_dense), (64, 4, 36, g_dense)):
        m = LCAONet(emb_size=16, emb_size_coeff=16, emb_size_conv=V, n_interaction=1, cutoff=5.0 if g is g_dense else 4.5,
                    max_z=max(max_z, 36)).to(dev)
        geo = _Geo(m, g)
        geo.n_l = n_l
        E, T = geo.E, geo.T
        gen = torch.Generator(device="cpu").manual_seed(V)
        w = torch.randn(T, n_l, generator=gen).to(dev)
        A = torch.randn(E, n_l, V, generator=gen).to(dev)
        G = torch.randn(E, V, generator=gen).to(dev)
        tp, tq = geo.tp.idx, geo.tq.idx
        want = torch.zeros(E, V, device=dev, dtype=torch.float64).index_add_(
            0, tp, torch.einsum("tl,tlv->tv", w.double(), A.double()[tq]))
        assert relerr(ops.trip_spmm(w, A, geo), want) <= 2e-6
        want_t = torch.zeros(E, n_l, V, device=dev, dtype=torch.float64).index_add_(
            0, tq, w.double().unsqueeze(2) * G.double()[tp].unsqueeze(1))
        assert relerr(ops.trip_spmm_t(w, G, geo), want_t) <= 2e-6
        want_d = torch.einsum("tv,tlv->tl", G.double()[tp], A.double()[tq])
        assert relerr(ops.trip_sddmm(G, A, geo), want_d) <= 2e-6


def test_triplet_weight_kernels_first_and_second_derivatives(dev):
    """csrc/lcao_double.hip k_trip_weight (w = Y_l(c) / |.|_G with the q == p slot dropped, lcaonet.py:175-184,470): values,
    gradient and gradient-of-gradient against the same function from fp64 tensor ops under autograd."""
    from lcaonet_amd.data.synthetic import random_periodic_cells
    from lcaonet_amd.model import LCAONet
    from lcaonet_amd.model.double import _Geo, _TripletWeightFn
    g = random_periodic_cells(2, 12, 6.0, 0.99 * 4.5, seed=5).to(dev)
    m = LCAONet(emb_size=16, emb_size_coeff=16, emb_size_conv=32, n_interaction=1, cutoff=4.5).to(dev)
    ys = (lambda c: torch.full_like(c, 0.28209479177387814), lambda c: 0.48860251190292 * c,
          lambda c: 0.94617469575756 * c * c - 0.31539156525252, lambda c: 0.373176332590115 * c * (5.0 * c * c - 3.0))
    for n_l in (1, 2, 3, 4):
        geo = _Geo(m, g)
        geo.n_l = n_l
        E, T = geo.E, geo.T
        tq, tmask = geo.tq.idx, geo.tmask.double()
        gen = torch.Generator(device="cpu").manual_seed(n_l)
        c0 = (2.0 * torch.rand(T, generator=gen) - 1.0).to(dev)
        B = torch.randn(E, n_l, 6, generator=gen).to(dev)
        G0 = torch.einsum("elv,emv->elm", B, B) + 0.05 * torch.randn(E, n_l, n_l, generator=gen).to(dev)  # not symmetric
        G0[0] = 0.0                                          # a clamped row: |.| = 1e-12, no gradient through the norm
        R, Hc, HG = (torch.randn(T, n_l, generator=gen).to(dev), torch.randn(T, generator=gen).to(dev),
                     torch.randn(E, n_l, n_l, generator=gen).to(dev))

        def ref(c, G):
            y = torch.stack([f(c) for f in ys[:n_l]], dim=1)
            n2 = torch.einsum("tl,tlm,tm->t", y, G[tq], y)
            return y * torch.rsqrt(torch.clamp_min(n2, 1e-24)).unsqueeze(1) * tmask.to(c.dtype)

        def chain(fn, c, G, R_, Hc_, HG_):
            c, G, R_ = c.clone().requires_grad_(True), G.clone().requires_grad_(True), R_.clone().requires_grad_(True)
            w = fn(c, G)
            gc, gG = torch.autograd.grad((w * R_).sum(), (c, G), create_graph=True, allow_unused=True)
            gc = torch.zeros_like(c) if gc is None else gc            # n_l = 1: no dependence on the angle
            second = torch.autograd.grad((gc * Hc_).sum() + (gG * HG_).sum(), (c, G, R_), allow_unused=True)
            second = [torch.zeros_like(x) if s_ is None else s_ for s_, x in zip(second, (c, G, R_))]
            return (w.detach(), gc.detach(), gG.detach()) + tuple(second)
        got = chain(lambda c, G: _TripletWeightFn.apply(c, G, geo), c0, G0, R, Hc, HG)
        want = chain(ref, c0.double(), G0.double(), R.double(), Hc.double(), HG.double())
        live = torch.ones(E, dtype=torch.bool, device=dev)
        live[0] = False                                      # the clamped row's w is ~1e12 y: compare the others in relerr
        tl = live[tq]
        names = ("w", "gc", "gG", "d2/dc", "d2/dG", "d2/dgw")
        for name, a, b in zip(names, got, want):
            mask = tl if a.shape[0] == T else live
            assert relerr(a[mask], b[mask]) <= 2e-5, (n_l, name)
            assert torch.allclose(a[~mask].double(), b[~mask], rtol=1e-4, atol=0.0) or name != "w", (n_l, name)


@pytest.mark.parametrize("which", ["silu", "rownorm"])
def test_pointwise_levels_first_and_second_derivatives(dev, which):
    """csrc/lcao_double.hip k_silu_levels / k_rownorm_levels (the f_node activations and F.normalize of lcaonet.py:142-150,
    196): value, gradient and gradient-of-gradient against torch autograd of the fp64 function."""
    from lcaonet_amd.model.double import _RowNormFn, _SiluFn
    fn, ref = {"silu": (_SiluFn.apply, torch.nn.functional.silu),
               "rownorm": (_RowNormFn.apply, lambda t: torch.nn.functional.normalize(t, dim=-1))}[which]
    for shape in ((1, 1), (37, 48), (501, 128), (5, 200), (3, 512), (1030, 7)):
        if which == "rownorm" and shape[1] == 1:
            continue                                         # y = sign(x): every derivative is exactly zero
        gen = torch.Generator(device="cpu").manual_seed(shape[0] + shape[1])
        x0 = (3.0 * torch.randn(*shape, generator=gen)).to(dev)
        if which == "rownorm" and shape[0] > 2:
            x0[1] = 0.0                                      # a clamped row: y = x / eps, g / eps, no curvature
        R, H = torch.randn(*shape, generator=gen).to(dev), torch.randn(*shape, generator=gen).to(dev)

        def chain(f, x, R_, H_):
            x, R_ = x.clone().requires_grad_(True), R_.clone().requires_grad_(True)
            y = f(x)
            (gx,) = torch.autograd.grad((y * R_).sum(), x, create_graph=True)
            g2x, g2r = torch.autograd.grad((gx * H_).sum(), (x, R_))
            return y.detach(), gx.detach(), g2x, g2r
        got = chain(fn, x0, R, H)
        want = chain(ref, x0.double(), R.double(), H.double())
        clamped = which == "rownorm" and shape[0] > 2
        for name, a, b in zip(("y", "gx", "d2/dx", "d2/dg"), got, want):
            if clamped and name == "d2/dx":                  # torch's own second derivative is NaN on the zero row; ours is 0
                assert bool((a[1] == 0).all())
                a, b = torch.cat([a[:1], a[2:]]), torch.cat([b[:1], b[2:]])
            assert relerr(a, b) <= 5e-6, (which, shape, name)


@pytest.mark.parametrize("rbf_type,cutoff_type", [("hydrogen", "envelope"), ("hydrogen", "polynomial"),
                                                  ("hydrogen", "cosine"), ("sphericalbessel", "envelope"),
                                                  ("hydrogen", "envelope_p7")])
def test_radial_levels_first_and_second_derivatives(dev, rbf_type, cutoff_type):
    """csrc/lcao_double.hip k_radial_levels (R, R', R'' of lcaonet/nn/rbf.py:17-182 with the envelopes of cutoff.py:10-67):
    value, gradient and gradient-of-gradient against fp64 tensor ops under autograd — the closed forms written out here
    from the orbital table."""
    import math
    from lcaonet_amd.atomistic.info import ElecInfo
    from lcaonet_amd.nn.cutoff import CosineCutoff, EnvelopeCutoff, PolynomialCutoff
    from lcaonet_amd.nn.rbf import HydrogenRadialBasis, SphericalBesselRadialBasis
    rc = 5.0
    cn = {"envelope": EnvelopeCutoff(rc), "polynomial": PolynomialCutoff(rc), "cosine": CosineCutoff(rc),
          "envelope_p7": EnvelopeCutoff(rc, 7)}[cutoff_type]
    ei = ElecInfo(96, None)
    basis = (HydrogenRadialBasis if rbf_type == "hydrogen" else SphericalBesselRadialBasis)(rc, ei, cn).to(dev)
    tab = basis.orb_table.double().cpu()
    D = tab.shape[0]
    gen = torch.Generator(device="cpu").manual_seed(7)
    r0 = (0.4 + 4.9 * torch.rand(4001, generator=gen)).to(dev)              # includes r > rc (identically zero there)
    R, H = torch.randn(4001, D, generator=gen).to(dev), torch.randn(4001, generator=gen).to(dev)

    def ref(r):
        cols = []
        for d in range(D):
            row = [float(v) for v in tab[d]]
            if rbf_type == "hydrogen":
                zeta = row[0] * r
                l, k = int(row[2]), int(row[3])
                poly = sum(row[4 + i] * zeta**i for i in range(k + 1))
                cols.append(row[1] * poly * zeta**l * torch.exp(-0.5 * zeta))
            else:
                cols.append(torch.sin(row[11] / rc * r) / r)
        return torch.stack(cols, dim=1) * cn.differentiable(r).unsqueeze(1)

    def chain(f, r, R_, H_):
        r, R_ = r.clone().requires_grad_(True), R_.clone().requires_grad_(True)
        rb = f(r)
        (gr,) = torch.autograd.grad((rb * R_).sum(), r, create_graph=True)
        g2r, g2g = torch.autograd.grad((gr * H_).sum(), (r, R_))
        return rb.detach(), gr.detach(), g2r, g2g
    got = chain(basis.differentiable, r0, R, H)
    want = chain(ref, r0.double(), R.double(), H.double())
    # the reverse sums mix orbitals of very different magnitude: compare per orbital for the (E, D) results
    for name, a, b in zip(("rb", "gr", "d2/dr", "d2/dg"), got, want):
        if a.dim() == 2:
            for d in range(D):
                assert relerr(a[:, d], b[:, d]) <= 2e-5, (name, d, relerr(a[:, d], b[:, d]))
        else:
            assert relerr(a, b) <= 2e-5, (name, relerr(a, b))
    assert math.isfinite(float(got[2].abs().max()))


def test_product_primitives_first_and_second_derivatives(dev):
    """model/double.py _MulPFn (a (c0 + b): the message product and the injection B (1 + ft)), _GateSumFn (gated copy +
    sum over l) and _PairPreFn (Ps[src] + Pt[dst]): value, gradient and gradient-of-gradient against the same expressions
    from fp64 tensor ops under autograd."""
    from lcaonet_amd.data.synthetic import random_periodic_cells
    from lcaonet_amd.model import LCAONet
    from lcaonet_amd.model.double import _Geo, _GateSumFn, _PairPreFn, mulp
    gen = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *shape: torch.randn(*shape, generator=gen).to(dev)

    def chain(f, inputs, R, Hs):
        xs = [x.clone().requires_grad_(True) for x in inputs]
        Rr = [r.clone().requires_grad_(True) for r in R]
        outs = f(*xs)
        outs = outs if isinstance(outs, tuple) else (outs,)
        g1 = torch.autograd.grad(sum((o * r).sum() for o, r in zip(outs, Rr)), xs, create_graph=True)
        g2 = torch.autograd.grad(sum((g * h).sum() for g, h in zip(g1, Hs)), xs + Rr, allow_unused=True)
        g2 = [torch.zeros_like(t) if g is None else g for g, t in zip(g2, xs + Rr)]
        return [o.detach() for o in outs] + [g.detach() for g in g1] + list(g2)
    dbl = lambda ts: [t.double() for t in ts]
    for c0 in (0.0, 1.0):
        for n in (5, 1030 * 48):
            a, b, R, Ha, Hb = rnd(n), rnd(n), rnd(n), rnd(n), rnd(n)
            got = chain(lambda x, y: mulp(x, y, c0), [a, b], [R], [Ha, Hb])
            want = chain(lambda x, y: x * (c0 + y), dbl([a, b]), dbl([R]), dbl([Ha, Hb]))
            assert all(relerr(g, w) <= 2e-6 for g, w in zip(got, want)), (c0, n)
    for E, L, V in ((7, 1, 16), (301, 3, 128), (64, 4, 48)):
        A, ge = rnd(E, L, V), rnd(E, V)
        R, Hs = [rnd(E, L, V), rnd(E, V)], [rnd(E, L, V), rnd(E, V)]
        got = chain(lambda x, y: _GateSumFn.apply(x, y), [A, ge], R, Hs)
        want = chain(lambda x, y: (x * y.unsqueeze(1), x.sum(1)), dbl([A, ge]), dbl(R), dbl(Hs))
        assert all(relerr(g, w) <= 2e-6 for g, w in zip(got, want)), (E, L, V)
    g = random_periodic_cells(2, 12, 6.0, 0.99 * 4.5, seed=3).to(dev)
    m = LCAONet(emb_size=16, emb_size_coeff=16, emb_size_conv=32, n_interaction=1, cutoff=4.5).to(dev)
    geo = _Geo(m, g)
    Ps, Pt = rnd(geo.N, 32), rnd(geo.N, 32)
    R, Hs = [rnd(geo.E, 32)], [rnd(geo.N, 32), rnd(geo.N, 32)]
    got = chain(lambda x, y: _PairPreFn.apply(x, y, geo), [Ps, Pt], R, Hs)
    want = chain(lambda x, y: x[geo.src.idx] + y[geo.dst.idx], dbl([Ps, Pt]), dbl(R), dbl(Hs))
    assert all(relerr(g_, w) <= 2e-6 for g_, w in zip(got, want))


def test_fused_level_maps_first_and_second_derivatives(dev):
    """model/double.py _InjNormFn (F.normalize(a (1 + b))), _SiluMulFn (a silu(x)), _PairSiluFn (silu(Ps[src] + Pt[dst])) and
    _ASideFn (contraction + gated copy +
    sum over l + Gram matrix; first level restricted to rb and gate as in the force pass): value, gradient and
    gradient-of-gradient against the same expressions from fp64 tensor ops under autograd."""
    import torch.nn.functional as F
    from lcaonet_amd.data.synthetic import random_periodic_cells
    from lcaonet_amd.model import LCAONet, double
    from lcaonet_amd.model.double import _ASideFn, _Geo, _InjNormFn, _PairSiluFn, _SiluMulFn
    gen = torch.Generator(device="cpu").manual_seed(11)
    rnd = lambda *shape: torch.randn(*shape, generator=gen).to(dev)

    def chain(f, inputs, R, Hs, first=None):
        xs = [x.clone().requires_grad_(True) for x in inputs]
        Rr = [r.clone().requires_grad_(True) for r in R]
        outs = f(*xs)
        outs = outs if isinstance(outs, tuple) else (outs,)
        wrt = xs if first is None else [xs[i] for i in first]
        g1 = torch.autograd.grad(sum((o * r).sum() for o, r in zip(outs, Rr)), wrt, create_graph=True)
        g2 = torch.autograd.grad(sum((g * h).sum() for g, h in zip(g1, Hs)), xs + Rr, allow_unused=True)
        g2 = [torch.zeros_like(t) if g is None else g for g, t in zip(g2, xs + Rr)]
        return [o.detach() for o in outs] + [g.detach() for g in g1] + list(g2)
    dbl = lambda ts: [t.double() for t in ts]
    for E, V in ((5, 16), (700, 128), (33, 48), (9, 200)):
        a, b = rnd(E, V), 0.5 * rnd(E, V)
        a[1] = 0.0                                                       # a row at the clamp of F.normalize
        R, Hs = [rnd(E, V)], [rnd(E, V), rnd(E, V)]
        got = chain(lambda x, y: _InjNormFn.apply(x, y), [a, b], R, Hs)
        want = chain(lambda x, y: F.normalize(x * (1 + y), dim=-1), dbl([a, b]), dbl(R), dbl(Hs))
        keep = torch.ones(E, dtype=torch.bool, device=dev)
        keep[1] = False                                                  # the clamped row: value and first level only
        for i, (g_, w_) in enumerate(zip(got, want)):
            assert relerr(g_[keep], w_[keep]) <= 5e-6, ("injnorm", E, V, i, relerr(g_[keep], w_[keep]))
        assert float(got[0][1].abs().max()) == 0.0
        got = chain(lambda x, y: _SiluMulFn.apply(x, y), [a, b], R, Hs)
        want = chain(lambda x, y: x * F.silu(y), dbl([a, b]), dbl(R), dbl(Hs))
        assert all(relerr(g_, w_) <= 5e-6 for g_, w_ in zip(got, want)), ("silumul", E, V)
    for V, max_z in ((32, 10), (64, 36), (128, 36), (64, 2), (64, 56), (32, 96)):   # n_l = 2, 3, 3, 1, 3, 4; D = 3 .. 18
        g = random_periodic_cells(2, 11, 6.0, 0.99 * 4.5, seed=3, max_z=max_z).to(dev)
        m = LCAONet(emb_size=16, emb_size_coeff=16, emb_size_conv=V, n_interaction=1, cutoff=4.5, max_z=max_z).to(dev)
        geo = _Geo(m, g)
        D, L = len(geo.l_of_d), geo.n_l
        rb, T, gate = rnd(geo.E, D), 0.3 * rnd(geo.zdim ** 2, D, V), rnd(geo.N, V)
        R = [rnd(geo.E, L, V), rnd(geo.E, V), rnd(geo.E, L, L)]
        Hs = [rnd(geo.E, D), rnd(geo.N, V)]
        lod = geo.l_of_d.long()
        pair = (geo.z[geo.src.idx] * geo.zdim + geo.z[geo.dst.idx])

        def literal(rb_, T_, gate_):
            onehot = torch.zeros(D, L, dtype=rb_.dtype, device=dev)
            onehot[torch.arange(D, device=dev), lod] = 1
            A = torch.einsum("pd,pdv,dl->plv", rb_, T_[pair], onehot)
            return A * gate_[geo.src.idx].unsqueeze(1), A.sum(1), torch.einsum("plv,pmv->plm", A, A)
        # first level restricted to (rb, gate) as in the force pass; second level w.r.t. everything
        xs = [t.clone().requires_grad_(True) for t in (rb, T, gate)]
        Rr = [r.clone().requires_grad_(True) for r in R]
        with double.positions_only():
            g1 = torch.autograd.grad(sum((o * r).sum() for o, r in zip(_ASideFn.apply(*xs, geo), Rr)), [xs[0], xs[2]],
                                     create_graph=True)
        g2 = torch.autograd.grad(sum((a_ * h).sum() for a_, h in zip(g1, Hs)), xs + Rr)
        got = [t.detach() for t in _ASideFn.apply(rb, T, gate, geo)] + [t.detach() for t in g1] + list(g2)
        want = chain(literal, dbl([rb, T, gate]), dbl(R), dbl(Hs), first=(0, 2))
        for i, (g_, w_) in enumerate(zip(got, want)):
            assert relerr(g_, w_) <= 1e-5, ("aside", V, i, relerr(g_, w_))
        Ps, Pt = rnd(geo.N, V), rnd(geo.N, V)
        R, Hs = [rnd(geo.E, V)], [rnd(geo.N, V), rnd(geo.N, V)]
        got = chain(lambda x, y: _PairSiluFn.apply(x, y, geo), [Ps, Pt], R, Hs)
        want = chain(lambda x, y: F.silu(x[geo.src.idx] + y[geo.dst.idx]), dbl([Ps, Pt]), dbl(R), dbl(Hs))
        assert all(relerr(g_, w_) <= 5e-6 for g_, w_ in zip(got, want)), ("pairsilu", V)


def test_rowgram_family_first_and_second_derivatives(dev):
    """csrc/lcao_double.hip rowgram / rowmix (the per-edge Gram matrix of lcaonet.py:179-184 and its derivatives): values
    against fp64 einsums, and gradient + gradient-of-gradient through model/double.py's Functions against autograd of
    the same einsum."""
    from lcaonet_amd import ops
    from lcaonet_amd.model.double import rowgram
    for E, n_l, V in ((1, 1, 16), (37, 3, 48), (501, 4, 128), (64, 2, 200), (9, 4, 512)):
        gen = torch.Generator(device="cpu").manual_seed(E + V)
        X = torch.randn(E, n_l, V, generator=gen).to(dev)
        Y = torch.randn(E, n_l, V, generator=gen).to(dev)
        S = torch.randn(E, n_l, n_l, generator=gen).to(dev)
        assert relerr(ops.rowgram(X, Y), torch.einsum("elv,emv->elm", X.double(), Y.double())) <= 2e-6
        assert relerr(ops.rowmix(S, X), torch.einsum("elm,emv->elv", S.double(), X.double())) <= 2e-6
        assert relerr(ops.rowmix(S, X, True), torch.einsum("eml,emv->elv", S.double(), X.double())) <= 2e-6
        # scalar f(A) = sum(c * tanh(G(A))) with G = A A^T;  compare df/dA and d/dA <df/dA, R> with the fp64 einsum graph
        A = (0.3 * X).requires_grad_(True)
        A64 = (0.3 * X).double().requires_grad_(True)
        c, R = S, Y

        def chain(A_, gram, c_, R_):
            f = (c_ * torch.tanh(gram(A_))).sum()
            (g1,) = torch.autograd.grad(f, A_, create_graph=True)
            (g2,) = torch.autograd.grad((g1 * R_).sum(), A_)
            return g1.detach(), g2
        g1, g2 = chain(A, lambda a: rowgram(a, a), c, R)
        w1, w2 = chain(A64, lambda a: torch.einsum("elv,emv->elm", a, a), c.double(), R.double())
        assert relerr(g1, w1) <= 5e-6 and relerr(g2, w2) <= 5e-6
    assert ops.rowgram(torch.zeros(0, 4, 32, device=dev), torch.zeros(0, 4, 32, device=dev)).shape == (0, 4, 4)
    with pytest.raises(ValueError):
        ops.rowmix(torch.zeros(3, 2, 2, device=dev), torch.zeros(3, 4, 32, device=dev))


@pytest.mark.parametrize("mode", ["inference-forces", "energy-gradients"])
def test_reverse_sweeps_are_additive_over_graphs_at_mid_size(dev, mode):
    """The same property for the hand-written reverse sweeps (model/forces.py, model/train.py) at ~108 k edges: forces of the
    whole batch equal the forces of its halves, and eval-mode parameter gradients of sum_g E_g^2 add up."""
    from lcaonet_amd.data.synthetic import jittered_cubic_cells
    forces = mode == "inference-forces"
    kw = dict(emb_size=64, emb_size_coeff=64, emb_size_conv=64, n_interaction=2, cutoff=5.0)
    if forces:
        kw.update(regress_forces=True, direct_forces=False)
    m, _ = _random_model(dev, 6, grad=not forces, **kw)
    gk = dict(n_side=4, a=3.2, sigma=0.05, cutoff_graph=4.95, device=dev)
    whole = jittered_cubic_cells(96, seed=22, cell_range=(0, 96), **gk)
    halves = [jittered_cubic_cells(96, seed=22, cell_range=r, **gk) for r in ((0, 48), (48, 96))]
    if forces:
        with torch.no_grad():
            ew, fw = m(whole)
            parts = [m(h) for h in halves]
        assert relerr(ew, torch.cat([p[0] for p in parts])) <= 1e-6
        assert relerr(fw, torch.cat([p[1] for p in parts])) <= 1e-6 and float(fw.abs().max()) > 0
        return

    def grads(g):
        m.zero_grad(set_to_none=True)
        e = m(g)
        (e * e).sum().backward()
        return [p.grad.clone() for p in m.parameters() if p.grad is not None]
    gw, ga, gb = grads(whole), grads(halves[0]), grads(halves[1])
    worst = max(relerr(a, b + c) for a, b, c in zip(gw, ga, gb))
    assert worst <= 2e-5, worst


def test_force_training_gradients_are_additive_over_graphs(dev):
    """A size-independent property at a size the oracle cannot hold (96 crystals, ~108 k edges, ~1.9 M triplets — past the
    single-tile / single-chunk regime of every kernel of the twice-differentiable path): in eval mode the graphs of a
    batch are independent, so d/dtheta sum_g (E_g^2 + |F_g|^2) over the whole batch equals the sum over its two halves."""
    from lcaonet_amd.data.synthetic import jittered_cubic_cells
    kw = dict(emb_size=64, emb_size_coeff=64, emb_size_conv=64, n_interaction=2, cutoff=5.0, regress_forces=True,
              direct_forces=False)
    m, _ = _random_model(dev, 5, grad=True, **kw)
    gk = dict(n_side=4, a=3.2, sigma=0.05, cutoff_graph=4.95, device=dev)
    whole = jittered_cubic_cells(96, seed=21, cell_range=(0, 96), **gk)
    halves = [jittered_cubic_cells(96, seed=21, cell_range=r, **gk) for r in ((0, 48), (48, 96))]

    def grads(g):
        m.zero_grad(set_to_none=True)
        e, f = m(g)
        ((e * e).sum() + (f * f).sum()).backward()
        return [p.grad.clone() for p in m.parameters() if p.grad is not None], e.detach(), f.detach()
    gw, ew, fw = grads(whole)
    parts = [grads(h) for h in halves]
    assert torch.equal(ew, torch.cat([p[1] for p in parts])) or relerr(ew, torch.cat([p[1] for p in parts])) <= 1e-6
    assert relerr(fw, torch.cat([p[2] for p in parts])) <= 1e-6
    assert float(fw.abs().max()) > 0
    worst = max(relerr(a, b + c) for a, b, c in zip(gw, parts[0][0], parts[1][0]))
    assert worst <= 2e-5, worst


# ---------------------------------------------------------------------------------------------------------------------
# single-graph TRAINING over ranks (BatchNorm statistics of the whole graph, autograd-aware halo exchange)
# ---------------------------------------------------------------------------------------------------------------------
SH_KW = dict(emb_size=32, emb_size_coeff=32, emb_size_conv=32, n_interaction=3, cutoff=4.0, out_size=2)


def _sharded_train_worker(rank, world, port, training, ret):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lcaonet_amd import shard
        from lcaonet_amd.data.synthetic import random_periodic_cells
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        m, _ = _random_model(dev, 4242, grad=True, **SH_KW)
        m.train(training)
        g = random_periodic_cells(1, 48, 9.0, 0.99 * 4.0, seed=23).to(dev)
        plan = shard.make_plans(g, world)[rank]
        out = m.forward_sharded_train(plan)
        loss = ((out - 0.3) ** 2).sum()
        loss.backward()
        shard.all_reduce_gradients(m)
        ret[rank] = (out.detach().cpu(), {n: p.grad.detach().cpu() for n, p in m.named_parameters()},
                     m.emb_layer.node_embed.bn.running_mean.detach().cpu(), plan.n_ghost)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("training", [False, True])
def test_sharded_training_matches_single_process(dev, training):
    """Two ranks (sharing cuda:0, talking over gloo) each hold half of ONE periodic cell: the global output, every
    parameter gradient after shard.all_reduce_gradients and the BatchNorm running statistics equal the single-process
    differentiable forward on the whole graph (train mode: batch statistics of the WHOLE graph from all-reduced counts)."""
    import socket
    import torch.multiprocessing as mp
    from lcaonet_amd.data.synthetic import random_periodic_cells
    m, _ = _random_model(dev, 4242, grad=True, **SH_KW)
    m.train(training)
    g = random_periodic_cells(1, 48, 9.0, 0.99 * 4.0, seed=23).to(dev)
    want = m(g, differentiable=True)
    ((want - 0.3) ** 2).sum().backward()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_sharded_train_worker, args=(2, port, training, ret), nprocs=2, join=True)
    assert len(ret) == 2
    for r in range(2):
        out, grads, run_mean, n_ghost = ret[r]
        assert n_ghost > 0
        assert relerr(out, want.detach()) <= 1e-5
        assert relerr(run_mean, m.emb_layer.node_embed.bn.running_mean) <= 1e-6
        checked = 0
        for n, p in m.named_parameters():
            if p.grad is None or float(p.grad.abs().max()) < 1e-10:
                continue
            assert relerr(grads[n], p.grad) <= 2e-4, (n, relerr(grads[n], p.grad))
            checked += 1
        assert checked > 25


def _grad_sweep_cases():
    import random
    rnd = random.Random(777)
    cases = []
    for i in range(10):
        V = rnd.choice([16, 24, 32, 48, 64, 128])
        kw = dict(emb_size=rnd.choice([16, 32, 64]), emb_size_coeff=rnd.choice([V, 32]), emb_size_conv=V,
                  n_interaction=rnd.choice([1, 2]), cutoff=4.5, max_z=rnd.choice([10, 36]), add_valence=rnd.random() < 0.3,
                  n_per_orb=rnd.choice([1, 2]), cutoff_net=rnd.choice(["envelope", "polynomial", "cosine"]),
                  rbf_type=rnd.choice(["hydrogen", "sphericalbessel"]), is_extensive=rnd.random() < 0.7,
                  activation=rnd.choice(["SiLU", "SiLU", "Tanh", "ShiftedSoftplus"]))
        cases.append((i, kw, rnd.random() < 0.5, rnd.random() < 0.5))
    return cases


@pytest.mark.parametrize("idx,kw,training,forces", _grad_sweep_cases())
def test_random_configuration_gradient_sweep(dev, idx, kw, training, forces):
    """Seeded random configurations through the differentiable paths — the fused training sweep (SiLU) or the generic
    twice-differentiable graph (other activations), with and without a force term in the loss (exact double backward) —
    every parameter gradient against torch.autograd through the fp64 oracle."""
    from lcaonet_amd.data.synthetic import random_periodic_cells
    kw = dict(kw)
    if forces:
        kw.update(regress_forces=True, direct_forces=False)
    m, sd = _random_model(dev, 6000 + idx, grad=True, **kw)
    m.train(training)
    g = random_periodic_cells(2, 9, 6.0, 0.99 * 4.5, seed=50 + idx, max_z=kw["max_z"])
    okw = {k: v for k, v in kw.items() if k != "activation"}
    cfg = _ocfg({**okw, "rbf_type": kw["rbf_type"], "training": training, "extra": {"activation": kw["activation"]}})
    sdg = {k: (v.clone().double().requires_grad_("running_" not in k) if v.is_floating_point() else v.clone())
           for k, v in sd.items()}
    ref = O.forward(sdg, cfg, g.to_dict(), dtype=torch.float64)
    out = m(g.to(dev))
    if forces:
        (e_ref, f_ref), (e, f) = ref, out
        assert relerr(e.detach(), e_ref.detach()) <= 2e-5 and relerr(f.detach(), f_ref.detach()) <= 5 * TOL
        ((e_ref ** 2).mean() + (f_ref ** 2).mean()).backward()
        ((e ** 2).mean() + (f ** 2).mean()).backward()
    else:
        assert relerr(out.detach(), ref.detach()) <= 2e-5
        (ref ** 2).mean().backward()
        (out ** 2).mean().backward()
    want = {k: v.grad for k, v in sdg.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
    checked, worst = _check_param_grads(m, want, 2e-4)
    assert checked > 15, checked
