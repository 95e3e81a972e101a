"""The MD snippet of INTEGRATION.md, runnable: velocity Verlet on a random-init LCAONet with a captured force graph."""
import sys, time
sys.path.insert(0, ".")
import torch
from lcaonet_amd.model import LCAONet, CapturedForward
from lcaonet_amd.data.neighbors import make_graph

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n_side, a = 4, 3.2
r = torch.arange(n_side, dtype=torch.float32)
pos = (torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3) * a + 0.05 * torch.randn(n_side**3, 3)).to(dev)
z = torch.randint(1, 37, (n_side**3,), device=dev)
lattice = (torch.eye(3) * n_side * a).unsqueeze(0).to(dev)
model = LCAONet(cutoff=5.0, regress_forces=True, direct_forces=False).eval().to(dev)
graph = make_graph(z, pos, lattice, cutoff=4.9, max_neighbors=None)
step = CapturedForward(model, graph)
vel = 0.02 * torch.randn_like(pos)
dt, n_steps = 0.02, 200
energy, forces = step(pos)
e0 = float(energy.sum()) + 0.5 * float((vel * vel).sum())
f = forces.clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_steps):
    vel += 0.5 * dt * f
    pos = pos + dt * vel
    energy, forces = step(pos)
    f = forces.clone()
    vel += 0.5 * dt * f
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n_steps * 1e3
e1 = float(energy.sum()) + 0.5 * float((vel * vel).sum())
print(f"{n_side**3} atoms, {graph['edge_index'].shape[1]} edges: {ms:.2f} ms per MD step; total energy {e0:.6f} -> {e1:.6f}")
