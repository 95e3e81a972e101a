"""Build container only: the REFERENCE's own fp32 autograd forces vs its fp64 forces on random periodic cells (graph cutoff
0.99 rc), and the same with its radial basis swapped for the restatement's (one envelope evaluation shared by the D orbitals).
Finding (profiles/r02/reference_fp32_force_noise.txt): the reference's fp32 backward is 0.3-0.8 % off its own fp64 forces; with
the restated basis 2e-6.  Parity tests therefore compare autograd forces with the fp64 run.
usage: python tools/diag_reference_force_noise.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo/oracle/standins"); sys.path.insert(0, "/root/reference"); sys.path.insert(0, "/root/repo")
from lcaonet.model import LCAONet as Ref
from torch_geometric.data import Data
from lcaonet_amd.data.synthetic import random_periodic_cells
from oracle import restatement as O
g = random_periodic_cells(2, 12, 6.0, 0.99 * 5.0, max_z=36, seed=7).to_dict()
def rel(a,b): return float((a-b).abs().max()/b.abs().max())
kw = dict(emb_size=16, emb_size_coeff=16, emb_size_conv=16, n_interaction=2, cutoff=5.0, regress_forces=True, direct_forces=False)
torch.manual_seed(41)
m = Ref(**kw).eval()
sd = {k: v.clone() for k, v in m.state_dict().items()}
def run(dt, patch=None):
    mm = Ref(**kw).eval(); mm.load_state_dict(sd); mm = mm.to(dt)
    if patch: patch(mm)
    gd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in g.items()}
    out, f = mm(Data(**gd)); return f.detach().double()
truth = run(torch.float64)
print("plain ref32", rel(run(torch.float32), truth))
nl = O.nl_list(36)
def p_rbf(mm): mm.rbf.forward = lambda r: O.hydrogen_radial_basis(r, nl, 5.0)
print("oracle rbf ", rel(run(torch.float32, p_rbf), truth))
def p_shb(mm): mm.shbf.forward = lambda c: O.spherical_harmonics_basis(c, nl)
print("oracle shbf", rel(run(torch.float32, p_shb), truth))
