#!/bin/bash
# Median device time of the LayerNorm kernels per (kernel, grid): tools/prof_ln.sh tag   (CTX_LN_G8=0 for the one-wave-per-row kernels)
TAG=${1:-lnprof}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -- python3 $GRAFT_REPO_ROOT/tools/bench_elementwise.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$TAG/*/*kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "layernorm" in n:
        agg[(n.split("(")[0][:40], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v = sorted(v); print("%-42s grid %8d wg %4d  n %3d  median %6.1f us" % (k[0], k[1], k[2], len(v), v[len(v) // 2]))
PY
