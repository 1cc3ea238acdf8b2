#!/usr/bin/env python3
"""Idle time between consecutive kernels of the UNet step in a rocprofv3 --kernel-trace CSV: for the timed steps of bench.py
(one stream), sort dispatches by start time and sum max(0, start[i+1] - end[i]).  Usage: prof_gaps.py DIR NSTEPS"""
import csv, glob, sys
d, nsteps = sys.argv[1], int(sys.argv[2])
t = glob.glob(d + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(t))]
rows.sort()
# the steady region: the last nsteps k_cfg_plms-delimited steps
marks = [i for i, r in enumerate(rows) if 'k_cfg_plms' in r[2]]
marks = marks[-(nsteps + 1):]
seg = rows[marks[0] + 1: marks[-1] + 1]
busy = sum(e - s for s, e, _ in seg)
gaps = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
over = sum(max(0, seg[i][1] - seg[i + 1][0]) for i in range(len(seg) - 1))
wall = seg[-1][1] - seg[0][0]
n = len(seg)
print(f"steps {len(marks) - 1}: {n / (len(marks) - 1):.0f} kernels/step, wall {wall / (len(marks) - 1) / 1e3:.1f} us/step, kernel time {busy / (len(marks) - 1) / 1e3:.1f} us/step, "
      f"gaps {sum(gaps) / (len(marks) - 1) / 1e3:.1f} us/step (mean {sum(gaps) / max(1, len(gaps)) / 1e3:.2f} us, overlap {over / (len(marks) - 1) / 1e3:.1f} us/step)")
big = sorted(((g, seg[i][2][:40], seg[i + 1][2][:40]) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, a, b in big:
    print(f"  gap {g / 1e3:7.2f} us after {a} before {b}")
hist = {}
for g in gaps:
    k = min(int(g / 500), 10)
    hist[k] = hist.get(k, 0) + 1
print("gap histogram (0.5 us bins):", {f"{k * 0.5:.1f}": v for k, v in sorted(hist.items())})
