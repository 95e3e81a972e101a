"""Summarise rocprofv3 --pmc CSVs: per kernel, mean of each counter per dispatch and mean duration.
Usage: pmc_summary.py <dir with *_counter_collection.csv and *_kernel_trace.csv> [name filter]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    vals[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals, key=lambda k: -sum(dur.get(k, [0]))):
    if flt and flt not in k:
        continue
    n = len(dur.get(k, []))
    if n == 0:
        continue
    md = sum(dur[k]) / n
    print(f"{k[:70]:70s} calls={n:4d} avg_us={md/1e3:10.1f}")
    for c, v in sorted(vals[k].items()):
        print(f"      {c:28s} {sum(v)/len(v):18.1f}")
