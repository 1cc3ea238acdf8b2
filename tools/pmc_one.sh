#!/bin/bash
# PMC passes over ONE GEMM / conv kernel (tools/one_gemm.py arguments after the tag).  Usage: tools/pmc_one.sh TAG USE8 TILE B H W Cin Cout FLAGS S ITERS
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL --output-format csv -d gpurun_out/$TAG/a -- python3 tools/one_gemm.py "$@" > gpurun_out/$TAG/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM --output-format csv -d gpurun_out/$TAG/b -- python3 tools/one_gemm.py "$@" > gpurun_out/$TAG/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in "ab":
    agg = collections.defaultdict(lambda: [0, 0.0])
    for t in glob.glob("gpurun_out/$TAG/%s/*/*counter_collection.csv" % p):
        for r in csv.DictReader(open(t)):
            if "k_gemm" in r["Kernel_Name"] or "k_conv" in r["Kernel_Name"]:
                a = agg[(r["Kernel_Name"][:40], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, a in sorted(agg.items()):
        print("%-42s %-36s per launch %.4g" % (k[0], k[1], a[1] / a[0]))
PY
