import numpy as np, torch, sys
sys.path.insert(0, '.')
from lcaonet_amd.atomistic.info import ElecInfo
from lcaonet_amd.nn.rbf import HydrogenRadialBasis
from lcaonet_amd.utils.resolve import cutoffnet_resolver
z = np.load('tests/golden/basis_tables.npz')
dev = torch.device('cuda')
ei = ElecInfo(96, None, None, 1)
for rc in (5.0, 6.0):
    for cut in ('envelope', 'polynomial', 'cosine'):
        r = torch.from_numpy(z[f'r_rc{rc:g}']).float().to(dev)
        cn = cutoffnet_resolver(cut, cutoff=rc)
        ref = torch.from_numpy(z[f'cutoff_{cut}_rc{rc:g}_f32'])
        ref64 = torch.from_numpy(z[f'cutoff_{cut}_rc{rc:g}_f64'])
        got = cn(r).cpu()
        print(rc, cut, 'cutoff max|d| vs f32ref', float((got-ref).abs().max()), 'nbits-equal', int((got==ref).sum()), '/', got.numel(), ' ref32 vs 64', float((ref.double()-ref64).abs().max()), ' mine vs 64', float((got.double()-ref64).abs().max()))
        rb = HydrogenRadialBasis(rc, ei, cn).to(dev)(r).cpu()
        R = torch.from_numpy(z[f'hydrogen_{cut}_rc{rc:g}_f32']); R64 = torch.from_numpy(z[f'hydrogen_{cut}_rc{rc:g}_f64'])
        worst = []
        for d in range(18):
            e = (rb[:, d]-R[:, d]).abs(); i = int(e.argmax())
            worst.append((float(e.max()/R[:, d].abs().max()), d, float(r[i])/rc, float(((R[:, d].double()-R64[:, d]).abs().max()/R64[:, d].abs().max())), float(((rb[:, d].double()-R64[:, d]).abs().max()/R64[:, d].abs().max()))))
        worst.sort(reverse=True)
        print('   worst cols (err, d, r/rc, ref32-vs-64, mine-vs-64):', [tuple(round(v, 8) if isinstance(v, float) else v for v in w) for w in worst[:3]])
