#!/bin/bash
# HBM traffic of the UNet step's kernels from the L2's memory-side counters (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC
# slots"): FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one), --kernel-trace only.  Usage: tools/pmc_traffic.sh TAG
TAG=${1:-traffic}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --vae 0 --two-views 0 --mesh 0 > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --vae 0 --two-views 0 --mesh 0 > gpurun_out/${TAG}_write.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write > gpurun_out/${TAG}.json
cat gpurun_out/${TAG}.json
