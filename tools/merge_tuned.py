#!/usr/bin/env python3
"""Merge plan tables written by tools/tune_gemm.py --emit: rows of the later files replace rows of the earlier ones with the
same (conv, M, N, K, flags, epi) key; new keys are appended.  Usage: merge_tuned.py BASE.h NEW1.h [NEW2.h ...] > OUT.h"""
import re, sys

ROW = re.compile(r"\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (-?\d+), (-?\d+), (\d+)\},(.*)")
head, rows, order = [], {}, []
for n, path in enumerate(sys.argv[1:]):
    for line in open(path):
        m = ROW.match(line)
        if m:
            key = tuple(int(x) for x in m.groups()[:6])
            if key not in rows:
                order.append(key)
            rows[key] = (tuple(int(x) for x in m.groups()[6:9]), m.group(10).strip())
        elif n == 0 and not rows and not line.startswith("};"):
            head.append(line)
sys.stdout.write("".join(head))
for key in order:
    (tile, use8, S), note = rows[key]
    sys.stdout.write("    {%d, %d, %d, %d, %d, %d, %d, %d, %d},   %s\n" % (*key, tile, use8, S, note))
sys.stdout.write("};\n")
