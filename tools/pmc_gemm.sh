#!/bin/bash
# PMC passes over a few GEMM/conv shapes (indices into tools/bench_gemm.py's list). Usage: tools/pmc_gemm.sh "0,9,2" tag
set -e
IDX=${1:-0,9}
TAG=${2:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d gpurun_out/${TAG}_a -- python3 tools/bench_gemm.py 1 $IDX > gpurun_out/${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_b -- python3 tools/bench_gemm.py 1 $IDX > gpurun_out/${TAG}_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/${TAG}_c -- python3 tools/bench_gemm.py 1 $IDX > gpurun_out/${TAG}_c.log 2>&1
ls gpurun_out/${TAG}_a/*/ gpurun_out/${TAG}_b/*/ gpurun_out/${TAG}_c/*/ 
