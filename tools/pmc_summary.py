#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel (first 40 chars of the name + grid). Usage: pmc_summary.py DIR [filter]"""
import csv, glob, collections, sys
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for t in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(t)):
        n = r['Kernel_Name']
        if flt not in n: continue
        key = (n[:44], r['Grid_Size'], r['Workgroup_Size'], r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''), r.get('LDS_Block_Size', ''))
        agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[key].add(r['Dispatch_Id'])
for k, v in agg.items():
    nd = len(cnt[k])
    print(f"{k[0]} grid={k[1]} wg={k[2]} vgpr={k[3]} agpr={k[4]} lds={k[5]} dispatches={nd}")
    for c, x in sorted(v.items()):
        print(f"    {c:36s} {x / nd:16.0f} /dispatch")
