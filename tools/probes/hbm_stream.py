"""Achievable HBM rates on this box for calibration: a streaming read (sum), a copy, and a random 1.5 KB row gather."""
import torch, time
dev = torch.device("cuda:0")
n = 1128478 * 3 * 128
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: y.copy_(x)); print(f"copy   {ms:.3f} ms  {2*4*n/ms/1e6:.0f} GB/s")
ms = t(lambda: x.sum()); print(f"sum    {ms:.3f} ms  {4*n/ms/1e6:.0f} GB/s")
rows = x.view(-1, 384)
idx = torch.randperm(rows.shape[0], device=dev)
out = torch.empty_like(rows)
ms = t(lambda: torch.index_select(rows, 0, idx, out=out)); print(f"gather {ms:.3f} ms  {2*4*n/ms/1e6:.0f} GB/s (random 1.5 KB rows, read+write)")
