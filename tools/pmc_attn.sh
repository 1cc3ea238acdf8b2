#!/bin/bash
# PMC passes over the attention kernel at the UNet's largest self-attention shape. Usage: tools/pmc_attn.sh tag
TAG=${1:-pmcattn}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/${TAG}_a -- python3 tools/bench_attn.py 3 0 > gpurun_out/${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_b -- python3 tools/bench_attn.py 3 0 > gpurun_out/${TAG}_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVE32_INSTS SQ_INSTS_SALU SQ_ACTIVE_INST_FLAT --output-format csv -d gpurun_out/${TAG}_c -- python3 tools/bench_attn.py 3 0 > gpurun_out/${TAG}_c.log 2>&1
for x in a b c; do python3 tools/pmc_summary.py gpurun_out/${TAG}_$x attention; done
