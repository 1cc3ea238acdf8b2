#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) totals per forward. Usage: prof_summary.py DIR NFWD [TOP]"""
import csv, glob, collections, sys
d, nf = sys.argv[1], float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
t = glob.glob(d + '/*/*kernel_trace.csv')[0]
agg = collections.defaultdict(lambda: [0, 0.0]); byname = collections.defaultdict(float)
for r in csv.DictReader(open(t)):
    n = r['Kernel_Name']
    if not any(k in n for k in ('gemm', 'splitk', 'attention', 'gn_', 'layernorm', 'transpose', 'concat', 'conv_', 'gemv', 'cfg_plms', 'time_embed', 'f32_to')):
        continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    key = (n[:36], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])
    agg[key][0] += 1; agg[key][1] += dur; byname[n[:28]] += dur
print("total us/forward %.1f" % (sum(v[1] for v in agg.values()) / nf))
for k, v in sorted(byname.items(), key=lambda kv: -kv[1]):
    print(f"  {k:30s} {v / nf:9.1f} us/fwd")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:36s} grid {k[1]:>8s}x{k[2]:>2s} wg {k[3]:>4s} n/fwd {v[0] / nf:5.1f} {v[1] / nf:8.1f} us/fwd avg {v[1] / v[0]:7.1f} us")
