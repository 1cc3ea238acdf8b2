#!/bin/bash
# Regenerates everything under profiles/ on a GPU box (run through gpurun; outputs land in gpurun_out/prof_refresh/ and are
# copied into profiles/ afterwards by tools/install_profiles.py).  Two gpurun calls (each under the 20-minute limit):
# Usage: ROUND=r03 PART=A tools/refresh_profiles.sh ; ROUND=r03 PART=B tools/refresh_profiles.sh
set -x
R=${ROUND:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_refresh
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "${PART:-A}" = "A" ]; then
bash tools/pmc_traffic.sh prof_refresh/traffic > /dev/null 2>&1
cp gpurun_out/prof_refresh/traffic.json profiles/${R}_pmc_traffic.json      # so that the default bench line below can quote it
bash tools/pmc_mfma.sh prof_refresh/mfma > /dev/null 2>&1
timeout -k 10 400 python3 bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log > $OUT/${R}_bench_default.json
timeout -k 10 300 python3 bench.py --mode mesh --steps 2 --warmup 1 > $OUT/bench_mesh.log 2>&1
tail -1 $OUT/bench_mesh.log > $OUT/${R}_bench_mesh_mode.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/unet -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-baseline 0 --vae 0 --two-views 0 --mesh 0 > $OUT/unet.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/geom -- python3 $GRAFT_REPO_ROOT/tools/bench_geometry.py > $OUT/geom.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py $OUT/unet 8 70 > $OUT/${R}_bench_by_kernel_and_grid.txt 2>&1
fi
if [ "${PART:-A}" = "B" ]; then
timeout -k 10 200 python3 tools/bench_geometry.py --cpu 1 > $OUT/${R}_geometry_bench.jsonl 2>&1
timeout -k 10 120 python3 tools/bench_gemm.py 3 all > $OUT/${R}_gemm_layers.txt 2>&1
timeout -k 10 120 python3 tools/bench_gemm_square.py > $OUT/${R}_gemm_square.txt 2>&1
timeout -k 10 300 python3 tools/bench_mesh.py 2>/dev/null | tail -1 > $OUT/${R}_mesh_bench.json
timeout -k 10 200 python3 tools/bench_concurrent.py 96 10 2>/dev/null | grep view > $OUT/${R}_views_in_flight.txt
timeout -k 10 200 python3 tools/bench_uvmlp.py 1024 5 2>/dev/null | tail -1 > $OUT/${R}_uvmlp_bench.json
timeout -k 10 200 python3 tools/bench_volume.py 512 128 3 2>/dev/null | tail -1 > $OUT/${R}_volume_bench.json
timeout -k 10 200 python3 tools/bench_zero123.py 10 2>/dev/null | tail -1 > $OUT/${R}_zero123_bench.json
timeout -k 10 300 python3 tools/bench_sds_loop.py 8 2>/dev/null | tail -1 > $OUT/${R}_sds_loop_bench.json
timeout -k 10 300 python3 tools/bench_batch.py 2>/dev/null | tail -1 > $OUT/${R}_mesh_batch_bench.json
timeout -k 10 200 python3 tools/bench_batched.py 96 10 2>/dev/null | grep views_batched > $OUT/${R}_views_batched.txt
CTX_UVMLP_EXACT_F32=1 timeout -k 10 200 python3 tools/bench_uvmlp.py 1024 5 2>/dev/null | tail -1 > $OUT/${R}_uvmlp_bench_exact_f32.json
fi
ls $OUT
