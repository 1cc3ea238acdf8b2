#!/bin/bash
# Median device time of the GroupNorm kernels per (kernel, grid) at batch $1 (default 2): tools/prof_gn.sh 2 tag
B=${1:-2}; TAG=${2:-gnprof}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -- python3 $GRAFT_REPO_ROOT/tools/bench_gn_batch.py $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$TAG/*/*kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gn_" in n:
        agg[(n.split("(")[0][-22:], int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v = sorted(v); print("%-24s grid %7d x %2d wg %4d  n %3d  median %6.1f us" % (k[0], k[1], k[2], k[3], len(v), v[len(v) // 2]))
PY
