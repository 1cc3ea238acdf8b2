#!/usr/bin/env python3
"""Per-kernel-family HBM traffic of one bench.py run from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled here;
WRITE_SIZE is taken as is.  Both counters are in KiB.  Usage: pmc_traffic.py FETCH_DIR WRITE_DIR  (prints JSON)"""
import csv, glob, json, sys, collections

FAM = [("gemm_conv", ("k_gemm_pipe", "k_gemm8", "k_gemm144", "k_splitk_reduce")), ("attention", ("k_attention",)),
       ("groupnorm", ("k_gn_",)), ("layernorm", ("k_layernorm",)), ("other", ("",))]


def fam(name):
    for f, keys in FAM:
        if any(k in name for k in keys):
            return f
    return "other"


def load(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for t in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(t)):
            if r['Counter_Name'] != counter:
                continue
            n = r['Kernel_Name']
            if not any(k in n for k in ('k_gemm', 'k_splitk', 'k_attention', 'k_gn_', 'k_layernorm', 'k_transpose', 'k_concat', 'k_conv_', 'k_gemv', 'k_cfg', 'k_time', 'k_f32')):
                continue
            a = agg[fam(n)]
            a[0] += 1; a[1] += float(r['Counter_Value'])
    return agg


f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 --cpu-baseline 0 --vae 0` (4 UNet evaluations)",
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B); WRITE_SIZE as reported; KiB -> bytes", "families": {}}
for k in f:
    n = f[k][0]
    rd = f[k][1] * 1024 * 2
    wr = w.get(k, [0, 0.0])[1] * 1024
    out["families"][k] = {"launches": n, "read_bytes": rd, "write_bytes": wr, "bytes_per_launch": (rd + wr) / max(n, 1)}
print(json.dumps(out, indent=1))
