"""Build profiles/rNN/pmc_traffic.json from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE in SEPARATE
passes: they do not fit one TCC pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots').  HBM bytes per launch =
2 * FETCH_SIZE[KiB] * 1024 + WRITE_SIZE[KiB] * 1024 (gfx950: FETCH_SIZE tallies the 128-B requests of wide coalesced reads
at 64 B, same guide, 'HBM').  usage: pmc_traffic.py <fetch dir> <write dir> <out.json> <commit> <date>"""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter):
    cc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == counter:
            vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in vals.items()}


# C-ABI entry point -> substring of the kernel's (mangled-free) name
ENTRY = {"lcao_coeff_mlp_contract": "k_coeff_mlp<", "lcao_triplet_gemm": "k_triplet_gemm<", "lcao_edge_lcao_weight": "k_edge_lcao_weight<",
         "lcao_edge_node_message_agg": "k_edge_node_message_agg<", "lcao_node_tables": "k_node_tables<", "lcao_dense": "k_dense_mfma<",
         "lcao_edge_geometry_basis": "k_edge_geometry_basis", "lcao_embed_coeffs": "k_embed_coeffs"}
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 --no-secondary "
                  "--no-cpu-baseline, cfg2; per-launch means over the dispatches of each kernel",
       "_correction": "bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (MI355X_MICROARCH.md, HBM: FETCH_SIZE counts 1/2 of wide "
                      "coalesced 16-B/lane reads on gfx950; cache-resident table reads never reach these memory-side counters)",
       "_commit": sys.argv[4], "_date": sys.argv[5]}
# entries served by two kernels per call of which one returns at once (device-side choice): per-call traffic = the SUM of the
# two kernels' per-launch means
PAIRED = {"lcao_edge_node_message_agg": "k_edge_node_message_cols<"}
for entry, pat in ENTRY.items():
    f = [(k, v) for k, v in fetch.items() if pat in k]
    w = [(k, v) for k, v in write.items() if pat in k]
    if not f or not w:
        continue
    fk = sum(v[0] * v[1] for _, v in f) / sum(v[1] for _, v in f)
    wk = sum(v[0] * v[1] for _, v in w) / sum(v[1] for _, v in w)
    if entry in PAIRED:
        f2 = [v for k, v in fetch.items() if PAIRED[entry] in k]
        w2 = [v for k, v in write.items() if PAIRED[entry] in k]
        if f2 and w2:
            fk += sum(v[0] * v[1] for v in f2) / sum(v[1] for v in f2)
            wk += sum(v[0] * v[1] for v in w2) / sum(v[1] for v in w2)
    out[entry] = {"fetch_kib": round(fk, 1), "write_kib": round(wk, 1), "launches_seen": int(sum(v[1] for _, v in f)),
                  "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
