"""Idle gaps between consecutive kernels in a rocprofv3 --kernel-trace csv (which kernel pairs leave the GPU idle).

usage: python tools/trace_gaps.py <dir containing *_kernel_trace.csv> [skip_fraction]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = sorted(glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[int(len(rows) * skip):]           # steady state only
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = max(0, s1 - e0)
    k = f"{n0}  ->  {n1}"
    gaps[k][0] += g
    gaps[k][1] += 1
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms ({100*(span-busy)/span:.1f} %)")
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"{g/1e3:9.1f} us total  {g/n/1e3:7.1f} us avg  x{n:4d}  {k}")
