#!/bin/bash
# Matrix-pipe utilisation per kernel family from SQ counters (one --pmc pass each, --kernel-trace only):
# the UNet step (bench.py) and the texture field (tools/bench_uvmlp.py).  Usage: tools/pmc_mfma.sh TAG
TAG=${1:-mfma}
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d gpurun_out/${TAG}_unet -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --vae 0 --two-views 0 --mesh 0 > gpurun_out/${TAG}_unet.log 2>&1
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d gpurun_out/${TAG}_field -- python3 tools/bench_uvmlp.py 1024 1 > gpurun_out/${TAG}_field.log 2>&1
python3 tools/pmc_mfma.py gpurun_out/${TAG}_unet gpurun_out/${TAG}_field > gpurun_out/${TAG}.txt
cat gpurun_out/${TAG}.txt
