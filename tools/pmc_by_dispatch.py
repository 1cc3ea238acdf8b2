#!/usr/bin/env python3
"""Per-dispatch memory-side traffic of ONE UNet evaluation from the two rocprofv3 --pmc passes of tools/pmc_traffic.sh
(FETCH_SIZE doubled as on gfx950, WRITE_SIZE as is): dispatches of the last evaluation in launch order, grouped by
(kernel, grid).  Usage: pmc_by_dispatch.py FETCH_DIR WRITE_DIR [--seq]"""
import csv, glob, sys, collections

KEYS = ('k_gemm', 'k_splitk', 'k_attention', 'k_gn_', 'k_layernorm', 'k_concat', 'k_conv_', 'k_gemv', 'k_cfg', 'k_time', 'k_f32', 'k_f16', 'k_geglu', 'k_transpose')


def load(d, counter):
    rows = []
    for t in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(t)):
            if r['Counter_Name'] == counter and any(k in r['Kernel_Name'] for k in KEYS):
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], int(r['Grid_Size']), float(r['Counter_Value']),
                             int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    rows.sort()
    return rows


def last_eval(rows):
    # an evaluation starts at the timestep-embedding kernel
    starts = [i for i, r in enumerate(rows) if 'k_time' in r[1]]
    if len(starts) < 2:
        return rows
    return rows[starts[-2]:starts[-1]]


f = last_eval(load(sys.argv[1], 'FETCH_SIZE'))
w = last_eval(load(sys.argv[2], 'WRITE_SIZE'))
assert len(f) == len(w), (len(f), len(w))
seq = '--seq' in sys.argv
agg = collections.OrderedDict()
tot_r = tot_w = 0.0
for i, (a, b) in enumerate(zip(f, w)):
    assert a[1] == b[1] and a[2] == b[2]
    rd, wr = a[3] * 2048.0, b[3] * 1024.0
    tot_r += rd; tot_w += wr
    name = a[1].replace('void ', '').replace('(GemmArgs)', '')[:46]
    if seq:
        print(f"{i:4d} {name:48s} grid {a[2]:>8d}  read {rd / 1e6:8.2f} MB  write {wr / 1e6:8.2f} MB  {a[4] / 1e3:7.1f} us")
    k = (name, a[2])
    e = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += rd; e[2] += wr; e[3] += a[4] / 1e3
print(f"# one evaluation: {len(f)} launches, read {tot_r / 1e9:.2f} GB, written {tot_w / 1e9:.2f} GB")
for k, e in sorted(agg.items(), key=lambda x: -(x[1][1] + x[1][2])):
    print(f"{k[0]:48s} grid {k[1]:>8d} x{e[0]:3d}  read {e[1] / e[0] / 1e6:8.2f} MB  write {e[2] / e[0] / 1e6:7.2f} MB each  {e[3] / e[0]:7.1f} us (under pmc)  total {(e[1] + e[2]) / 1e6:8.1f} MB")
